/*
 * comprox_amd/csrc/crgpu_rop5.h — comprop lzdecode for the batched API: the ppm_decode step in GCN assembly.
 *
 * Reference: /root/reference/src/ropmain/cr-coder.c:231-292 (lzdecode), src/cr-ppm.c:169-235
 * (ppm_decode), src/cr-rangecoder.c:81-104 (range decoder), src/cr-o2model.c:54-71,93-113 (node update,
 * symbol search), src/cr-ppm.c:66-98 (order-3 / order-1 updates).
 *
 * Why assembly: a datablock is a chain of ~43 000 dependent ppm_decode steps run by one wavefront, and a
 * lone wave issues one instruction per ~4.1 clocks whatever its kind (tools/issue_probe.hip), ~25 clocks
 * for a taken branch. In-kernel stamps (tools/dec_profile.py) put ~2 300 clocks of every step of the C++
 * versions (round 1, since removed) into plain instruction issue: the compiler turns every uniform
 * condition into a 64-bit lane mask and a branch, ~350 instructions per step however the source is
 * phrased. The step below is ~190 instructions on the common path (byte symbol found in the order-2 node)
 * with one taken branch; what is rare is out of line.
 *
 * Division of labour: this file's asm statement runs coding steps until something RARE happens and
 * returns an event to the C++ around it: a match token (LZP lookup + copy, crgpu_lzp.h), 64 pending
 * literal positions to learn, the 256-byte input window used up, end of block. Every event leaves the
 * model tables complete in memory (all stores done), so re-entry simply loads the model of the current
 * context again.
 *
 * Tables (arena offsets are compile-time constants, crgpu_device.h): an order-2 node is ONE 128-byte line at context * 128 —
 * up to 62 {symbol, count} pairs in symbol order, one per lane, and the generation-tagged flag word in its last four bytes
 * (lanes 62 / 63) — so the node and its flag word cost one line to fetch and one to write back (round 3; before: 256 count
 * bytes = two lines + a flag word in an array of its own = a third). 88 % of the steps of a bench block meet such a node;
 * a node that gets a 63rd symbol moves to a slot of 256 count bytes in a small dense area (its line then names the slot),
 * and the step has a second variant for those. Order-3 predictor direct-indexed by the reference's 22-bit key, u16 {byte,
 * 4-bit generation, confidence}; order-1 rows dense. A step's three loads go out as soon as the symbol is known, its
 * three or four stores after the register updates, and the wait before the next step is vmcnt(k) — the node and order-3
 * loads only (the order-1 row is waited for by the escape path). Loads issued before the previous step's stores are patched
 * from registers (same node: keep PP / W / SX; same order-3 key: O3LV; same order-1 row after an escape: ROWU).
 *
 * Hazards are padded by hand as the compiler pads them for gfx950 (VALU result -> DPP 2 wait states,
 * VALU/DPP result -> v_readlane / v_readfirstlane 1, v_rcp result 1, VALU-written VCC -> VALU 2).
 */
#ifndef CRGPU_ROP5_H
#define CRGPU_ROP5_H

#include "crgpu_dec.h"

/* The statement exists in two modes (an assembly-time switch in front of the register map):
 *   0  comprop: escape byte / length symbol tokens, LZP tables, pending positions (everything below);
 *   1  plain PPM symbol stream (comprox's main stream, crgpu_rox5.h): a literal is stored and pushed into the context,
 *      the escape byte ends the statement (CR_V5_EV_ESC) after its model update, without a push;
 *   2  the same with mode 0's pending positions (comprolz, crgpu_rolz5.h): lane j of the pending registers holds the
 *      8 bytes in front of position learned + j, 64 of them end the statement (CR_V5_EV_LEARN); nothing is pending
 *      below position 16 (cr-matcher.c:68). */
#define CR_V5_ASM_MODE(m_) ".set c5_mode, " #m_ "\n .set c5_hw, 0\n .set c5_dl, 0\n"
#define CR_V5_ASM_MODE_DL(m_) ".set c5_mode, " #m_ "\n .set c5_hw, 0\n .set c5_dl, 1\n"       /* with the first dense nodes in LDS (CR_V5_DLDS below) */
/* mode 0 with a HELPER wave (round 5, CRGPU_OPT_DECODER_HELPER; VERDICT r4 task 1): the workgroup has a second wave that prepares the
 * escape's order-1 sums. At the head of every step on a line node the coder posts the line's pairs, the order-1 row and the
 * predicted byte in an LDS mailbox (cr_rop_decode_helper below computes the masked weights, their 64-lane prefix and the total
 * from them, cr-ppm.c:209-211); on symbol 257 it waits for the helper's word and reads its lane's four values back instead of
 * computing them. Halvings, first-use nodes and dense nodes keep the coder's own path. The mailbox follows the wave's 272
 * scratch bytes: +0 post word (seq << 9 | predicted byte; 0xffffffff = stop), +4 the helper's word (seq << 9), +8 total,
 * +12 the coder's seq between statements, +16 pairs u16[64], +144 row u32[64], +400 {below, even weights, odd weights,
 * prefix} u32[4] per lane, +1424 the helper's own 272 scratch bytes. Only what the CURRENT post describes is ever read by
 * the coder, and a post is complete before its word is written: no lock, a helper that falls behind just answers late. */
#define CR_V5_ASM_MODE_HW ".set c5_mode, 0\n .set c5_hw, 1\n .set c5_dl, 0\n"
#define CR_V5_HW_LDS_BYTES (272u + 1696u)
#define CR_V5_EV_ESC    8u
#define CR_V5_EV_MATCH  1u
#define CR_V5_EV_LEARN  2u
#define CR_V5_EV_WINDOW 3u
#define CR_V5_EV_DONE   4u
#define CR_V5_EV_FAIL   5u

/* diagnostic build (-DCR_V5_PROF=k): shader clocks spent (stats slot 9) and visits (slot 12) of one place:
 * 1 the wait at the end of every step, 2 a whole match token, 3 / 4 / 5 the match token's waits for the table
 * operations / the context checks and source bytes / the next context's model */
#ifndef CR_V5_ALIGN                                /* experiment: 64-byte alignment of the step's branch targets (1 head, 2 not-in-node, 4 hit) */
#define CR_V5_ALIGN 0
#endif
#ifdef CR_V5_SWAP                                  /* experiment: the node's line ahead of the order-3 entry */
#define CR_V5_SWAP_SET ".set c5_swap, 1\n .set c5_align, " CR_V5_STR(CR_V5_ALIGN) "\n"
#else
#define CR_V5_SWAP_SET ".set c5_swap, 0\n .set c5_align, " CR_V5_STR(CR_V5_ALIGN) "\n"
#endif
#define CR_V5_STR2(x) #x
#define CR_V5_STR(x) CR_V5_STR2(x)
#ifdef CR_V5_PROF
#define CR_V5_PROF_SET ".set c5_prof, " CR_V5_STR(CR_V5_PROF) "\n" CR_V5_SWAP_SET
#else
#define CR_V5_PROF_SET ".set c5_prof, 0\n" CR_V5_SWAP_SET
#endif
/* cache policy of the step's model stores (an experiment switch, -DCR_V5_STPOL=k): 0 plain, 1 nt, 2 sc1, 3 sc0 sc1, 4 sc1 nt */
#ifndef CR_V5_STPOL
#define CR_V5_STPOL 0
#endif
#define CR_V5_STPOL_SET ".set c5_stpol, " CR_V5_STR(CR_V5_STPOL) "\n"
/* software prefetch of the NEXT step's model (-DCR_V5_PF=bits): a step only learns its successor context when its symbol is
 * decoded, but the successor is (previous byte, symbol) and the symbol is — on the steps that do not escape — one of the
 * node's own pairs or the order-3 prediction, so the candidates' node lines (context << 7) can be asked for as soon as the
 * line is in: the real load behind the search then meets a line that is on its way (or in the L2) instead of starting an
 * HBM miss of its own. bit 0: the lines of the pairs with count >= CR_V5_PFTHR and of the predicted byte, from the scan's
 * wait states; bit 1: the 512-byte group of order-3 entries of every successor (it only depends on the context: issued a
 * step ahead, beside the context's own loads); bit 2: the predicted byte's line on a first-use (empty) node. The loads
 * land in a register nothing reads; every vmcnt wait of the statement stays at least as strict as before. */
#ifndef CR_V5_PF
#define CR_V5_PF 0
#endif
#ifndef CR_V5_PFTHR
#define CR_V5_PFTHR 1
#endif
/* -DCR_V5_FAIR=k: paired waves take turns at the higher issue priority every 2^k output bytes (c5_prio below; 0 = off) */
#ifndef CR_V5_FAIR
#define CR_V5_FAIR 12
#endif
/* the first CR_V5_DLDS dense nodes also live in the wave's LDS (round 5): a dense node is named by its line, so its 256 count
 * bytes used to cost a second, dependent round trip through global memory on 12 % of the bench block's steps (~740 clocks each at
 * 1 526 blocks, profiles/r06j). The dense area in the arena stays what it was (every store still goes there, so the count of
 * vector-memory operations a step issues is unchanged); the LDS copy is written beside it and is what a step reads. Slots are
 * handed out in the order nodes outgrow their lines, i.e. the busiest contexts come first (16 slots serve every dense step of the
 * bench corpus). A kernel chooses by its mode string (CR_V5_ASM_MODE_DL) and sizes its LDS with CR_V5_LDS_BYTES. */
#ifndef CR_V5_DLDS
#define CR_V5_DLDS 32
#endif
/* mode 0's learn event (64 pending literals into the LZP tables, cr-matcher.c:66-72) without its waits (round 5): the event used to cost
 * 6 400 clocks at 1 526 blocks, 280 times a block (3.8 % of the kernel, profiles/r06j) — a compare-and-swap round trip and three to four
 * collision rounds, each waited for. Nothing reads the tables before the next match token, so the event now only ISSUES its first
 * round; every following step's end (which has waited for everything older than its own loads) looks at the results and issues the
 * next round, until no lane is left. Until then LIMIT stays 0, which sends each step's end down the rare path (no instruction on the
 * common one); any event, the next learn event and every exit finish what is in flight first (c5_learn_drain). The walk's registers
 * (v78 - v95) belong to the match token, which a step does not touch; what is still pending lives in v96 / v98 (the match token's too: it
 * only runs once the walks are finished), "in flight" in v144, which nothing else writes. */
#ifndef CR_V5_ALEARN
#define CR_V5_ALEARN 1
#endif
#define CR_V5_LDS_BYTES (272u + 256u * (CR_V5_DLDS + 1u))           /* scratch (256 + 16), CR_V5_DLDS slots, one slot nothing reads */
#ifndef CR_V5_SIDE_DL                                               /* comprox's / comprolz's decoders (crgpu_rox5.h, crgpu_rolz5.h) with them? */
#define CR_V5_SIDE_DL 1
#endif
#if CR_V5_SIDE_DL
#define CR_V5_SIDE_MODE(m_) CR_V5_ASM_MODE_DL(m_)
#define CR_V5_SIDE_LDS_WORDS (CR_V5_LDS_BYTES / 4u)
#else
#define CR_V5_SIDE_MODE(m_) CR_V5_ASM_MODE(m_)
#define CR_V5_SIDE_LDS_WORDS 68u
#endif
#define CR_V5_PF_SET ".set c5_fair, " CR_V5_STR(CR_V5_FAIR) "\n .set c5_pf, " CR_V5_STR(CR_V5_PF) "\n .set c5_pfthr, " CR_V5_STR(CR_V5_PFTHR) "\n" \
    ".set c5_dlds, c5_dl * " CR_V5_STR(CR_V5_DLDS) "\n .set c5_VDL, 61\n .set c5_alearn, (c5_mode == 0) * (1 - c5_hw) * " CR_V5_STR(CR_V5_ALEARN) "\n" \
    ".set c5_PMV8, 96\n .set c5_FLV, 144\n .set c5_PMV4, 98\n"


/* register map of the asm statement (all clobbered): SGPR 34..99, VGPR 32..71. Some names share a register
 * with one whose value is dead by then: OL = T4, WW = TB (after the in-node decision), SL = FHIT (after the token), NOW = TB and XOFF = FESC (from the
 * model update to the stores). */
#define CR_V5_ASM_DEFS \
    ".set c5_MW, 94\n .set c5_ARENA, 34\n .set c5_DST, 36\n .set c5_LIMIT, 38\n .set c5_LOFF, 39\n" \
    ".set c5_CTX, 40\n " \
    ".set c5_WIDX, 41\n .set c5_HAVE, 48\n .set c5_LEARNED, 49\n .set c5_AESC, 50\n .set c5_NCTX, 51\n" \
    ".set c5_X8LO, 52\n .set c5_X8HI, 53\n .set c5_NDNO, 54\n .set c5_SX, 55\n .set c5_O3LK, 56\n .set c5_O3LV, 57\n" \
    ".set c5_LRIDX, 58\n .set c5_TOTAL, 59\n .set c5_GEN, 60\n .set c5_G3S, 61\n .set c5_K3N, 62\n .set c5_EV, 63\n" \
    ".set c5_KEY, 64\n .set c5_K3, 65\n .set c5_PRED, 66\n .set c5_CONF, 67\n .set c5_ROWI, 68\n .set c5_BYTES, 69\n" \
    ".set c5_TB, 72\n .set c5_SS, 73\n .set c5_LOWER, 74\n .set c5_FRQ, 75\n" \
    ".set c5_SYM, 76\n .set c5_FHIT, 77\n .set c5_FESC, 78\n .set c5_LIT, 79\n" \
    ".set c5_T0, 80\n .set c5_T1, 81\n .set c5_T2, 82\n .set c5_T3, 83\n .set c5_T4, 84\n .set c5_T5, 85\n .set c5_T6, 86\n .set c5_T7, 87\n" \
    ".set c5_LB, 88\n .set c5_LUTM, 90\n .set c5_LUTH, 92\n .set c5_OL, 84\n .set c5_NO, 96\n" \
    ".set c5_HALV, 97\n .set c5_NON, 98\n .set c5_PM, 99\n .set c5_SL, 77\n .set c5_WW, 72\n .set c5_NOW, 72\n .set c5_XOFF, 78\n" \
    ".set c5_LANE, 32\n .set c5_VLANE4, 33\n .set c5_VFB, 34\n .set c5_W, 36\n .set c5_NW, 37\n .set c5_FX, 38\n" \
    ".set c5_FE, 39\n .set c5_FROW, 40\n .set c5_WX, 41\n .set c5_SUM, 42\n .set c5_INCL, 43\n .set c5_P, 44\n .set c5_ROWU, 45\n" \
    ".set c5_PENDLO, 46\n .set c5_PENDHI, 47\n .set c5_WIN, 48\n .set c5_VPM, 49\n .set c5_VT0, 50\n .set c5_VT1, 51\n" \
    ".set c5_KEEP, 54\n .set c5_MINE, 55\n .set c5_INCL1, 56\n .set c5_ROW, 57\n" \
    ".set c5_AW, 60\n .set c5_AX, 61\n .set c5_AE, 62\n .set c5_AR, 63\n .set c5_SA, 64\n .set c5_SA2, 65\n .set c5_SD2, 66\n" \
    ".set c5_SA3, 67\n .set c5_SD3, 68\n .set c5_SA4, 69\n .set c5_SD4, 70\n .set c5_SA5, 71\n" \
    ".set c5_PACC, 72\n .set c5_PCNT, 73\n" \
    /* the coder in vector registers (uniform values, see "Why the coder is vector code" above): state in 118..123, the \
     * rest are temporaries of one step; 64..68 are the store registers (dead until the stores), 103..116 belong to the \
     * match token's tail (dead during a step) */ \
    ".set c5_BN, 42\n .set c5_B3, 44\n .set c5_B1, 46\n .set c5_VCLO, 118\n .set c5_VCACHE, 119\n .set c5_VIBLO, 120\n .set c5_VIBHI, 121\n .set c5_VRANGE, 122\n .set c5_VIBITS, 123\n" \
    ".set c5_VTP, 124\n .set c5_VUNIT, 126\n .set c5_VTOT, 127\n" \
    ".set c5_DM, 64\n .set c5_DNEG, 65\n .set c5_DQ1, 66\n .set c5_DR, 67\n .set c5_DR1, 68\n" \
    /* the division's doubles (even-aligned pairs; v71 is C2): total, its reciprocal, the Newton residual over the store registers, range + 0.5 in a pair of its own */ \
    ".set c5_O3E, 87\n .set c5_HWSEQ, 100\n .set c5_HWPOST, 101\n .set c5_HWB, 34\n .set c5_HWA2, 35\n .set c5_HWA4, 38\n .set c5_HWA16, 39\n .set c5_HWR, 64\n .set c5_DD, 64\n .set c5_DRC, 66\n .set c5_DE, 68\n .set c5_DN, 58\n" \
    ".set c5_EXCL, 103\n .set c5_C1, 104\n .set c5_C2, 105\n .set c5_C3, 106\n .set c5_VFHIT, 107\n .set c5_VFESC, 108\n" \
    ".set c5_VHE, 109\n .set c5_VTB, 110\n .set c5_VLOWU, 111\n .set c5_VFRQ, 112\n .set c5_VWW, 113\n .set c5_VUNIT1, 114\n" \
    ".set c5_ROWK, 115\n .set c5_FE, 116\n .set c5_FO, 52\n .set c5_P0, 53\n" \
    /* match token (live from the end of the length symbol's step to the next step's head only) */ \
    ".set c5_MK8, 64\n .set c5_MK4, 65\n .set c5_MK2, 66\n .set c5_MH8, 67\n .set c5_MH4, 68\n .set c5_C8, 69\n .set c5_C4, 70\n" \
    ".set c5_C2, 71\n .set c5_LZM, 72\n .set c5_E8K, 73\n .set c5_ACT, 74\n .set c5_E8P, 77\n .set c5_E4K, 78\n .set c5_E4P, 79\n" \
    ".set c5_LM, 88\n .set c5_U0, 94\n .set c5_U1, 95\n .set c5_U2, 96\n .set c5_U3, 97\n .set c5_PF, 98\n .set c5_PM8, 94\n .set c5_PM4, 96\n" \
    ".set c5_VQ, 74\n .set c5_VK8, 75\n .set c5_VK4, 76\n .set c5_VK2, 77\n .set c5_VH8, 78\n .set c5_VH4, 79\n .set c5_A8, 80\n" \
    ".set c5_A4, 81\n .set c5_A2, 82\n .set c5_E2, 83\n .set c5_D8, 84\n .set c5_D4, 88\n .set c5_R8, 92\n .set c5_R4, 94\n" \
    ".set c5_E8, 96\n .set c5_E4, 98\n .set c5_LA8, 100\n .set c5_LA4, 101\n .set c5_LA2, 102\n .set c5_V4, 103\n .set c5_V8, 104\n" \
    ".set c5_S8, 106\n .set c5_S4, 107\n .set c5_S2, 108\n .set c5_CPY, 109\n .set c5_XALO, 110\n .set c5_XAHI, 111\n .set c5_XT, 112\n .set c5_XU, 113\n .set c5_B8, 114\n .set c5_B4, 115\n .set c5_B2, 116\n" \
    /* the node as a line of pairs (lane i = pair i, lanes 62 / 63 = the flag word's halves) and what goes with it; per-lane \
     * constants: VLANE2 = 2 x lane, VMCNT = 0xff below lane 62 (else 0), VHI = 0x100 from lane 62 on (else 0), VLDZ = the \
     * lane's dword of the wave's 256-byte LDS scratch, VLDB = the scratch's first byte, VDOFF4 = the dense area + 4 x lane; \
     * VDA = this dense node's dword of the lane, VDSLOT = the next free dense slot (uniform) */ \
    ".set c5_PP, 128\n .set c5_VSYM, 129\n .set c5_CX, 130\n .set c5_VLANE2, 131\n .set c5_VMCNT, 132\n .set c5_VHI, 133\n .set c5_PRES, 134\n" \
    ".set c5_VLDZ, 135\n .set c5_VLDB, 136\n .set c5_VDA, 137\n .set c5_VDSLOT, 138\n .set c5_VDOFF4, 139\n .set c5_VT2, 140\n .set c5_VT3, 141\n" \
    ".set c5_VZERO, 142\n .set c5_VONE, 143\n .set c5_VLDX, 117\n" \
    CR_V5_PROF_SET CR_V5_STPOL_SET CR_V5_PF_SET ".set c5_PFD, 144\n .set c5_PFA, 145\n .set c5_OFF_NODES, 262144\n .set c5_OFF_O1, 8650752\n .set c5_OFF_O3D, 8716288\n .set c5_OFF_SCR, 4096\n"
static_assert(CRGPU_OFF_NODES == 262144u && CRGPU_LINE_BYTES == 128u && CRGPU_LINE_PAIRS == 62u && CRGPU_NODE_BYTES == 256u && CRGPU_DEC_OFF_O1 == 8650752u &&
              CRGPU_DEC_OFF_O3D == 8716288u && CRGPU_OFF_SCRATCH == 4096u, "the assembly's table offsets follow crgpu_device.h");

/* macros: inclusive 64-lane scan, 32-bit division (the compiler's reciprocal sequence), the four model
 * loads of a context, range_decoder_decode (cr-rangecoder.c:91-99), o2_model_update's halving pass */
#define CR_V5_ASM_MACROS R"ASM(
.ifndef c5_macros
.set c5_macros, 1
.macro c5_scan dst, src
  s_nop 1
  v_add_u32_dpp v[\dst], v[\src], v[\src] row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1
  s_nop 1
  v_add_u32_dpp v[\dst], v[\dst], v[\dst] row_shr:2 row_mask:0xf bank_mask:0xf bound_ctrl:1
  s_nop 1
  v_add_u32_dpp v[\dst], v[\dst], v[\dst] row_shr:4 row_mask:0xf bank_mask:0xf bound_ctrl:1
  s_nop 1
  v_add_u32_dpp v[\dst], v[\dst], v[\dst] row_shr:8 row_mask:0xf bank_mask:0xf bound_ctrl:1
  s_nop 1
  v_add_u32_dpp v[\dst], v[\dst], v[\dst] row_bcast:15 row_mask:0xa bank_mask:0xf
  s_nop 1
  v_add_u32_dpp v[\dst], v[\dst], v[\dst] row_bcast:31 row_mask:0xc bank_mask:0xf
  s_nop 0
.endm
.macro c5_vdiv_pre
  ; the dividend's half of c5_vdiv: DN = range + 0.5 as a double (it only depends on the coder: placed in a scan's wait states)
  v_cvt_f64_u32 v[c5_DN:c5_DN+1], v[c5_VRANGE]
  v_add_f64 v[c5_DN:c5_DN+1], v[c5_DN:c5_DN+1], 0.5
.endm
.macro c5_vdiv q, fill:vararg
  ; \q = VRANGE / VTOT (cr-rangecoder.c:101-104), all uniform vector registers, in double precision (round 5): r = 1 / VTOT by
  ; v_rcp_f64 + one Newton step (relative error ~2^-46), q = trunc((range + 0.5) x r). (range + 0.5) / total lies at least
  ; 0.5 / total from an integer and the product is off by less than (range / total) x 2^-45: exact for every u32 range and every
  ; total below 2^20 (tools/div_probe.hip: 5e8 divisions at the quotient boundaries of every total, 0 wrong; 100 clocks against
  ; the 160 of the u32 reciprocal sequence it replaces). DN = range + 0.5 comes from c5_vdiv_pre. \fill = an instruction of
  ; the caller's that has to run somewhere around here. A total of 0 (damaged stream) gives NaN -> 0, no trap.
  v_cvt_f64_u32 v[c5_DD:c5_DD+1], v[c5_VTOT]
  v_rcp_f64 v[c5_DRC:c5_DRC+1], v[c5_DD:c5_DD+1]
  \fill
  v_fma_f64 v[c5_DE:c5_DE+1], -v[c5_DD:c5_DD+1], v[c5_DRC:c5_DRC+1], 1.0
  v_fma_f64 v[c5_DRC:c5_DRC+1], v[c5_DE:c5_DE+1], v[c5_DRC:c5_DRC+1], v[c5_DRC:c5_DRC+1]
  v_mul_f64 v[c5_DN:c5_DN+1], v[c5_DN:c5_DN+1], v[c5_DRC:c5_DRC+1]
  v_cvt_u32_f64 v[\q], v[c5_DN:c5_DN+1]
.endm
.macro c5_issue c, ta=c5_T0, tb=c5_T1, tc=c5_T2, pk=99
.if \pk != 99                                      ; (profile builds 8 - 12 keep their start stamp in s100:101, the helper build's pair)
  c5_prof_end 8, c5_HWSEQ
  c5_prof_end \pk, c5_HWSEQ
.endif
  ; the three model loads of context \c: the order-3 entry, the node's line (128 B at context << 7: pairs + flag word), the
  ; order-1 row (BN / B3 / B1 = the tables). Every instruction in front of the second load is on the symbol-to-symbol path:
  ; the entry's four address instructions first, NON (the line's offset, read at the next head) behind the loads.
  s_lshr_b32 s[\tb], s[\c], 2                   ; cr-ppm.c:66, the order-3 key of the context
  s_xor_b32 s[\tb], s[\tb], s[\c]
  s_and_b32 s[c5_K3N], s[\tb], 0x3fffff
  v_lshlrev_b32_e64 v[c5_AE], 1, s[c5_K3N]
  s_and_b32 s[\ta], s[\c], 0xffff
.if c5_swap
  v_lshl_add_u32 v[c5_AW], s[\ta], 7, v[c5_VLANE2]
  s_and_b32 s[\tc], s[\c], 0xff
  global_load_ushort v[c5_NW], v[c5_AW], s[c5_BN:c5_BN+1]
  global_load_ushort v[c5_FE], v[c5_AE], s[c5_B3:c5_B3+1]
.else
  global_load_ushort v[c5_FE], v[c5_AE], s[c5_B3:c5_B3+1]
  v_lshl_add_u32 v[c5_AW], s[\ta], 7, v[c5_VLANE2]
  s_and_b32 s[\tc], s[\c], 0xff
  global_load_ushort v[c5_NW], v[c5_AW], s[c5_BN:c5_BN+1]
.endif
  v_lshl_add_u32 v[c5_AR], s[\tc], 8, v[c5_VLANE4]
  s_lshl_b32 s[c5_NON], s[\ta], 7
  global_load_dword v[c5_FROW], v[c5_AR], s[c5_B1:c5_B1+1]
.if \pk != 99
  c5_prof_begin 9, c5_HWSEQ
.endif
.if c5_pf & 2
  ; the order-3 entries of every successor of \c: keys ((c << 8 | s) ^ (c << 8 | s) >> 2) & 0x3fffff, s = 0 .. 255 — one
  ; aligned group of 256 u16 entries (cr-ppm.c:66), its four lines asked for by lanes 0 .. 3
  s_lshl_b32 s[\ta], s[\c], 8
  s_lshr_b32 s[\tb], s[\ta], 2
  s_xor_b32 s[\ta], s[\ta], s[\tb]
  s_and_b32 s[\ta], s[\ta], 0x3fff00
  s_lshl_b32 s[\ta], s[\ta], 1
  v_lshl_add_u32 v[c5_PFA], v[c5_LANE], 7, s[\ta]
  s_mov_b64 exec, 0xf
  global_load_ubyte v[c5_PFD], v[c5_PFA], s[c5_B3:c5_B3+1]
  s_mov_b64 exec, -1
.endif
.endm
.macro c5_pick_sp u, check=1
  ; in-node symbol of a line (its lane OL): its count (FRQ) and (the count below it) x unit
.if \check
  s_cmp_lt_u32 s[c5_SS], 0x100
  s_cbranch_scc0 .Lc5_spicked_\u\()_\@
.endif
  v_readlane_b32 s[c5_FRQ], v[c5_SUM], s[c5_OL]
  v_readlane_b32 s[c5_T1], v[c5_P], s[c5_OL]
  v_mov_b32 v[c5_VFRQ], s[c5_FRQ]
  v_mul_lo_u32 v[c5_VLOWU], v[c5_VFRQ], v[c5_VUNIT]
  v_sub_u32 v[c5_VLOWU], s[c5_T1], v[c5_VLOWU]
.Lc5_spicked_\u\()_\@:
.endm
.macro c5_pick u, check=1
  ; in-node symbol SS: its count (FRQ, also as a scalar for the node update) and (the count below it) x unit
.if \check
  s_cmp_lt_u32 s[c5_SS], 0x100
  s_cbranch_scc0 .Lc5_picked_\u\()_\@
.endif
  s_lshr_b32 s[c5_OL], s[c5_SS], 2                 ; (OL and LOWER = the symbol's lane and its byte's shift: kept for the update)
  s_lshl_b32 s[c5_LOWER], s[c5_SS], 3
  s_and_b32 s[c5_LOWER], s[c5_LOWER], 24
  v_readlane_b32 s[c5_WW], v[c5_WX], s[c5_OL]
  v_readlane_b32 s[c5_T1], v[c5_EXCL], s[c5_OL]
  s_lshr_b32 s[c5_FRQ], s[c5_WW], s[c5_LOWER]
  s_and_b32 s[c5_FRQ], s[c5_FRQ], 0xff
  v_mov_b32 v[c5_VWW], s[c5_WW]
  v_mov_b32 v[c5_VFRQ], s[c5_FRQ]
  v_bfe_u32 v[c5_VWW], v[c5_VWW], 0, s[c5_LOWER]
  v_sad_u8 v[c5_VLOWU], v[c5_VWW], 0, s[c5_T1]
  v_mul_lo_u32 v[c5_VLOWU], v[c5_VLOWU], v[c5_VUNIT]
.Lc5_picked_\u\()_\@:
.endm
.macro c5_consume unit, lowu=c5_VLOWU, frq=c5_VFRQ
  ; range_decoder_decode (cr-rangecoder.c:91-99) with lower x unit in \lowu and the count in \frq
  v_sub_u32 v[c5_VCACHE], v[c5_VCACHE], v[\lowu]
  v_mul_lo_u32 v[c5_DM], v[\unit], v[\frq]
  v_mov_b32 v[c5_VCLO], v[c5_VIBHI]
  v_ffbh_u32 v[c5_DR], v[c5_DM]
  v_and_b32 v[c5_DR], 24, v[c5_DR]
  v_lshlrev_b32 v[c5_VRANGE], v[c5_DR], v[c5_DM]
  v_lshlrev_b64 v[c5_VCLO:c5_VCLO+1], v[c5_DR], v[c5_VCLO:c5_VCLO+1]
  v_lshlrev_b64 v[c5_VIBLO:c5_VIBLO+1], v[c5_DR], v[c5_VIBLO:c5_VIBLO+1]
  v_sub_u32 v[c5_VIBITS], v[c5_VIBITS], v[c5_DR]
  v_cmp_ge_u32 vcc, 32, v[c5_VIBITS]               ; 32 bits or fewer left: the caller refills on vccnz
.endm
.macro c5_refill
  v_readlane_b32 s[c5_T0], v[c5_WIN], s[c5_WIDX]
  v_sub_u32 v[c5_DR], 32, v[c5_VIBITS]
  v_mov_b32 v[c5_VTP+1], 0
  v_mov_b32 v[c5_VTP], s[c5_T0]
  s_add_u32 s[c5_WIDX], s[c5_WIDX], 1
  v_lshlrev_b64 v[c5_VTP:c5_VTP+1], v[c5_DR], v[c5_VTP:c5_VTP+1]
  v_add_u32 v[c5_VIBITS], 32, v[c5_VIBITS]
  s_cmp_ge_u32 s[c5_WIDX], 62                      ; the window is nearly used up: no further step (LIMIT = 0)
  v_or_b32 v[c5_VIBLO], v[c5_VIBLO], v[c5_VTP]
  v_or_b32 v[c5_VIBHI], v[c5_VIBHI], v[c5_VTP+1]
  s_cselect_b32 s[c5_LIMIT], 0, s[c5_LIMIT]
.endm
.macro c5_halve
  v_lshrrev_b32 v[c5_W], 1, v[c5_W]
  v_and_b32 v[c5_W], 0x7f7f7f7f, v[c5_W]
  v_xor_b32 v[c5_VT0], 0x01010101, v[c5_W]
  v_and_b32 v[c5_VT1], 0x7f7f7f7f, v[c5_VT0]
  v_add_u32 v[c5_VT1], 0x7f7f7f7f, v[c5_VT1]
  v_or_b32 v[c5_VT1], v[c5_VT1], v[c5_VT0]
  v_or_b32 v[c5_VT1], 0x7f7f7f7f, v[c5_VT1]
  v_not_b32 v[c5_VT1], v[c5_VT1]
  v_bcnt_u32_b32 v[c5_VT1], v[c5_VT1], 0
  c5_scan c5_VT0, c5_VT1
  v_readlane_b32 s[c5_T0], v[c5_VT0], 63
  s_add_u32 s[c5_T0], s[c5_T0], 1
  s_and_b32 s[c5_T0], s[c5_T0], 0xff
  s_and_b32 s[c5_T1], s[c5_SX], 0xff
  s_add_u32 s[c5_T1], s[c5_T1], 1
  s_lshr_b32 s[c5_T1], s[c5_T1], 1
  s_lshl_b32 s[c5_T0], s[c5_T0], 8
  s_or_b32 s[c5_SX], s[c5_T1], s[c5_T0]
.endm
.macro c5_halve_sp
  ; o2_model_update's halving pass (cr-o2model.c:72-84) on a line: counts floor-halved (a pair may fall to 0: as good as absent),
  ; count(256) = (count(256) + 1) / 2, count(257) = 1 + the pairs left with count 1
  v_and_b32 v[c5_VT0], v[c5_VMCNT], v[c5_PP]
  v_lshrrev_b32 v[c5_VT0], 1, v[c5_VT0]
  v_bfi_b32 v[c5_PP], v[c5_VMCNT], v[c5_VT0], v[c5_PP]
  v_cmp_eq_u32 vcc, 1, v[c5_VT0]
  s_bcnt1_i32_b64 s[c5_T0], vcc
  s_add_u32 s[c5_T0], s[c5_T0], 1
  s_and_b32 s[c5_T0], s[c5_T0], 0xff
  s_and_b32 s[c5_T1], s[c5_SX], 0xff
  s_add_u32 s[c5_T1], s[c5_T1], 1
  s_lshr_b32 s[c5_T1], s[c5_T1], 1
  s_lshl_b32 s[c5_T0], s[c5_T0], 8
  s_or_b32 s[c5_SX], s[c5_T1], s[c5_T0]
.endm
.macro c5_literal
.if c5_mode != 1
  s_lshr_b64 s[c5_X8LO:c5_X8LO+1], s[c5_X8LO:c5_X8LO+1], 8
  s_lshl_b32 s[c5_T0], s[c5_LIT], 24
  s_or_b32 s[c5_X8HI], s[c5_X8HI], s[c5_T0]
.endif
  s_mov_b64 s[c5_LB:c5_LB+1], s[c5_DST:c5_DST+1]
  s_mov_b32 s[c5_LOFF], s[c5_HAVE]
  s_add_u32 s[c5_HAVE], s[c5_HAVE], 1
.if c5_mode == 2
  s_cmp_le_u32 s[c5_HAVE], 16                      ; positions below 16 are never fed to the matcher
  s_cselect_b32 s[c5_LEARNED], s[c5_HAVE], s[c5_LEARNED]
.endif
.endm
.macro c5_gst op, a, d, b, off=0
.if c5_stpol == 0
  \op v[\a], v[\d], s[\b:\b+1] offset:\off
.elseif c5_stpol == 1
  \op v[\a], v[\d], s[\b:\b+1] offset:\off nt
.elseif c5_stpol == 2
  \op v[\a], v[\d], s[\b:\b+1] offset:\off sc1
.elseif c5_stpol == 3
  \op v[\a], v[\d], s[\b:\b+1] offset:\off sc0 sc1
.else
  \op v[\a], v[\d], s[\b:\b+1] offset:\off sc1 nt
.endif
.endm
.macro c5_o3_miss                                   ; ppm_update_o3(c), cr-ppm.c:75-80
  s_lshl_b32 s[c5_T0], s[c5_CONF], 2
  s_lshr_b64 s[c5_T0:c5_T0+1], s[c5_LUTM:c5_LUTM+1], s[c5_T0]
  s_and_b32 s[c5_CONF], s[c5_T0], 15
  s_cmp_eq_u32 s[c5_CONF], 0
  s_cselect_b32 s[c5_PRED], s[c5_SYM], s[c5_PRED]
  s_max_u32 s[c5_CONF], s[c5_CONF], 1
.endm
.macro c5_o3_hit                                    ; ppm_update_o3(-1), cr-ppm.c:81-83
  s_lshl_b32 s[c5_T0], s[c5_CONF], 2
  s_lshr_b64 s[c5_T0:c5_T0+1], s[c5_LUTH:c5_LUTH+1], s[c5_T0]
  s_and_b32 s[c5_CONF], s[c5_T0], 15
.endm
; a step's stores: only what the step changed. A line: the pairs (and flag halves) in the lanes of MW, one store; a dense
; node: its count word(s) in the lanes of MW and the flag word in its line. Both leave exec = 1 for the single-lane
; stores behind them (order-3 entry, output byte).
.macro c5_st_pairs
  v_add_u32 v[c5_SA], s[c5_NO], v[c5_VLANE2]
  s_mov_b64 exec, s[c5_MW:c5_MW+1]
  c5_gst global_store_short, c5_SA, c5_PP, c5_BN
  s_mov_b64 exec, 1
.endm
.macro c5_st_node
  s_mov_b64 exec, s[c5_MW:c5_MW+1]
  c5_gst global_store_dword, c5_VDA, c5_W, c5_ARENA
.if c5_dlds
  ds_write_b32 v[c5_VDL], v[c5_W] offset:0x110     ; the node's LDS copy (a slot past the LDS ones: the spare slot)
.endif
  s_mov_b64 exec, 1
.endm
.macro c5_st_flag
  s_lshl_b32 s[c5_T1], s[c5_GEN], 16
  v_mov_b32 v[c5_SA2], s[c5_NO]
  s_or_b32 s[c5_T1], s[c5_T1], s[c5_SX]
  v_mov_b32 v[c5_SD2], s[c5_T1]
  c5_gst global_store_dword, c5_SA2, c5_SD2, c5_BN, 124
.endm
.macro c5_st_o3_lit
  s_lshl_b32 s[c5_O3LV], s[c5_PRED], 8
  s_or_b32 s[c5_O3LV], s[c5_O3LV], s[c5_G3S]
  s_or_b32 s[c5_O3LV], s[c5_O3LV], s[c5_CONF]
  v_lshlrev_b32_e64 v[c5_SA3], 1, s[c5_K3]
  v_mov_b32 v[c5_SD3], s[c5_O3LV]
  c5_gst global_store_short, c5_SA3, c5_SD3, c5_B3
  s_mov_b32 s[c5_O3LK], s[c5_K3]
  v_mov_b32 v[c5_SA4], s[c5_LOFF]
  v_mov_b32 v[c5_SD4], s[c5_LIT]
  global_store_byte v[c5_SA4], v[c5_SD4], s[c5_LB:c5_LB+1]
  s_mov_b64 exec, -1
.endm
.macro c5_st_row
  v_lshl_add_u32 v[c5_SA5], s[c5_ROWI], 8, v[c5_VLANE4]
  c5_gst global_store_dword, c5_SA5, c5_ROWU, c5_B1
.endm
; end of a step that issued \k - 1 stores: the next step's node and order-3 loads are back when at most \k operations
; are out (its order-1 row, issued last and only read by an escape, and this step's stores)
.macro c5_tail k, u
  s_mov_b32 s[c5_CTX], s[c5_NCTX]
  c5_prof_end 9, c5_HWSEQ
.if c5_prof == 19                                  ; (19: how many steps find their model there: PACC counts the steps whose wait is under ~64 clocks, PCNT all)
  s_memtime s[c5_T0:c5_T0+1]
  s_waitcnt lgkmcnt(0)
  s_waitcnt vmcnt(\k)
  s_memtime s[c5_T2:c5_T2+1]
  s_waitcnt lgkmcnt(0)
  s_sub_u32 s[c5_T2], s[c5_T2], s[c5_T0]
  s_cmp_lt_u32 s[c5_T2], 110                       ; (the two stamps cost ~46 of them)
  s_cselect_b32 s[c5_T2], 1, 0
  v_add_u32 v[c5_PACC], s[c5_T2], v[c5_PACC]
  v_add_u32 v[c5_PCNT], 1, v[c5_PCNT]
.elseif c5_prof == 15                              ; (15 / 16: the wait for the order-3 entry alone, then what the node's line adds to it)
  s_memtime s[c5_T0:c5_T0+1]
  s_waitcnt lgkmcnt(0)
  s_waitcnt vmcnt(\k + 1)
  s_memtime s[c5_T2:c5_T2+1]
  s_waitcnt lgkmcnt(0)
  s_sub_u32 s[c5_T2], s[c5_T2], s[c5_T0]
  v_add_u32 v[c5_PACC], s[c5_T2], v[c5_PACC]
  v_add_u32 v[c5_PCNT], 1, v[c5_PCNT]
  s_waitcnt vmcnt(\k)
.elseif c5_prof == 16
  s_waitcnt vmcnt(\k + 1)
  s_memtime s[c5_T0:c5_T0+1]
  s_waitcnt lgkmcnt(0)
  s_waitcnt vmcnt(\k)
  s_memtime s[c5_T2:c5_T2+1]
  s_waitcnt lgkmcnt(0)
  s_sub_u32 s[c5_T2], s[c5_T2], s[c5_T0]
  v_add_u32 v[c5_PACC], s[c5_T2], v[c5_PACC]
  v_add_u32 v[c5_PCNT], 1, v[c5_PCNT]
.elseif c5_prof == 1
  s_memtime s[c5_T0:c5_T0+1]
  s_waitcnt lgkmcnt(0)
  s_waitcnt vmcnt(\k + ((c5_pf >> 1) & 1) - c5_hw)
  s_memtime s[c5_T2:c5_T2+1]
  s_waitcnt lgkmcnt(0)
  s_sub_u32 s[c5_T2], s[c5_T2], s[c5_T0]
  v_add_u32 v[c5_PACC], s[c5_T2], v[c5_PACC]
  v_add_u32 v[c5_PCNT], 1, v[c5_PCNT]
.else
  s_waitcnt vmcnt(\k + ((c5_pf >> 1) & 1) - c5_hw)   ; (with a helper wave the order-1 row is posted at the head: it has to be in as well)
.endif
  s_cmp_lt_u32 s[c5_HAVE], s[c5_LIMIT]             ; (a token that raises an event zeroes LIMIT: one test on the common path)
  s_cbranch_scc1 .Lc5_head_\u
  s_branch .Lc5_slow_tail_\u
.endm
.macro c5_prof_begin k, reg=c5_PF
.if c5_prof == \k
  s_memtime s[\reg:\reg+1]
  s_waitcnt lgkmcnt(0)
.endif
.endm
.macro c5_prof_end k, reg=c5_PF
.if c5_prof == \k
  s_memtime s[c5_T6:c5_T6+1]
  s_waitcnt lgkmcnt(0)
  s_sub_u32 s[c5_T6], s[c5_T6], s[\reg]
  v_add_u32 v[c5_PACC], s[c5_T6], v[c5_PACC]
  v_add_u32 v[c5_PCNT], 1, v[c5_PCNT]
.endif
.endm
.macro c5_lzp_round r, a, d, h, soff, pm, u
  ; cr_lzp_learn's second half for one table (crgpu_lzp.h), one round, not waited for: of the lanes in \pm (their
  ; compare-and-swap result is in \r) those that met their own key raise its entry, those that met another key
  ; try the next slot and stay in \pm; a lane that claimed an empty slot is done
  s_cmp_eq_u64 s[\pm:\pm+1], 0
  s_cbranch_scc1 .Lc5_rnd_done_\u\()_\@
  v_cmp_ne_u32 vcc, 0, v[\r+1]
  s_and_b64 s[c5_T0:c5_T0+1], vcc, s[\pm:\pm+1]
  v_cmp_eq_u32 vcc, v[\r+1], v[\d+1]
  s_and_b64 s[c5_T2:c5_T2+1], s[c5_T0:c5_T0+1], vcc
  s_andn2_b64 s[\pm:\pm+1], s[c5_T0:c5_T0+1], vcc
  s_mov_b64 exec, s[c5_T2:c5_T2+1]
  global_atomic_umax_x2 v[\a], v[\d:\d+1], s[c5_ARENA:c5_ARENA+1]
  s_mov_b64 exec, s[\pm:\pm+1]
  v_add_u32 v[\h], 1, v[\h]
  v_and_b32 v[\h], s[c5_LZM], v[\h]
  v_lshlrev_b32 v[\a], 3, v[\h]
  v_add_u32 v[\a], \soff, v[\a]
  global_atomic_cmpswap_x2 v[\r:\r+1], v[\a], v[\d:\d+3], s[c5_ARENA:c5_ARENA+1] sc0
  s_mov_b64 exec, -1
.Lc5_rnd_done_\u\()_\@:
.endm
.macro c5_lzp_finish r, a, d, h, soff, pm, u
  ; the same, waited for, until no lane of \pm is left
  s_cmp_eq_u64 s[\pm:\pm+1], 0
  s_cbranch_scc1 .Lc5_fin_done_\u\()_\@
.Lc5_fin_loop_\u\()_\@:
  c5_lzp_round \r, \a, \d, \h, \soff, \pm, \u
  s_waitcnt vmcnt(0)
  s_cmp_lg_u64 s[\pm:\pm+1], 0
  s_cbranch_scc1 .Lc5_fin_loop_\u\()_\@
.Lc5_fin_done_\u\()_\@:
.endm
.macro c5_prio u
  ; Two decoder waves on one SIMD do not share it evenly: at equal priority the OLDER wave issues nearly unimpeded and the younger
  ; one gets what is left (MI355X_MICROARCH.md, two waves per SIMD) — and the kernel lasts as long as its slowest block. They
  ; take turns instead (round 5): every 2^c5_fair output bytes a wave's priority flips between 3 and 2, in opposite phase for
  ; the even and the odd wave slots of a SIMD (HW_ID bit 0). Both stay above the 0 of whatever another stream runs beside
  ; the decoder. 2.6 percent off the bench stream with 4 096 bytes (profiles/r06i_paired_waves_priority.txt); called where the
  ; statement is entered and where 64 literals are learned.
.if c5_fair
  s_getreg_b32 s[c5_T0], hwreg(HW_REG_HW_ID, 0, 1)
  s_lshr_b32 s[c5_T1], s[c5_HAVE], c5_fair
  s_xor_b32 s[c5_T0], s[c5_T0], s[c5_T1]
  s_bitcmp1_b32 s[c5_T0], 0
  s_cbranch_scc1 .Lc5_prio_hi_\u\()_\@
  s_setprio 2
  s_branch .Lc5_prio_set_\u\()_\@
.Lc5_prio_hi_\u\()_\@:
  s_setprio 3
.Lc5_prio_set_\u\()_\@:
.endif
.endm
.macro c5_lzp_finish2 off8, off4, u
  ; both tables' walks together: a round of each, one wait for both, until no lane of either is left (the results of the
  ; operations in flight have been waited for when this starts)
.Lc5_fin2_loop_\u\()_\@:
  c5_lzp_round c5_R8, c5_A8, c5_D8, c5_VH8, \off8, c5_PM8, \u
  c5_lzp_round c5_R4, c5_A4, c5_D4, c5_VH4, \off4, c5_PM4, \u
  s_or_b64 s[c5_T0:c5_T0+1], s[c5_PM8:c5_PM8+1], s[c5_PM4:c5_PM4+1]
  s_cmp_eq_u64 s[c5_T0:c5_T0+1], 0
  s_cbranch_scc1 .Lc5_fin2_done_\u\()_\@
  s_waitcnt vmcnt(0)
  s_branch .Lc5_fin2_loop_\u\()_\@
.Lc5_fin2_done_\u\()_\@:
.endm
.macro c5_learn_masks lzsh
  v_cmp_ne_u32 vcc, 0, v[c5_PMV8]
  s_mov_b64 s[c5_PM8:c5_PM8+1], vcc
  v_cmp_ne_u32 vcc, 0, v[c5_PMV4]
  s_lshr_b32 s[c5_LZM], -1, \lzsh
  s_mov_b64 s[c5_PM4:c5_PM4+1], vcc
.endm
.macro c5_learn_drain off8, off4, lzsh, u
  ; a learn event still in flight: its walks to the end, waited for (before an event, the next learn event, an exit)
.if c5_alearn
  v_readfirstlane_b32 s[c5_T5], v[c5_FLV]
  s_cmp_eq_u32 s[c5_T5], 0
  s_cbranch_scc1 .Lc5_drained_\u\()_\@
  s_waitcnt vmcnt(0)
  c5_learn_masks \lzsh
  c5_lzp_finish2 \off8, \off4, \u
  v_mov_b32 v[c5_FLV], 0
.Lc5_drained_\u\()_\@:
.endif
.endm
.macro c5_lzp_probe c, dflt, h, e, la, soff, u
  ; cr_htab_get_from: the home slot holds another key (T2 = the key looked for + 1): walk on, wave-uniform
.Lc5_pr_loop_\u\()_\@:
  s_add_u32 s[\h], s[\h], 1
  s_and_b32 s[\h], s[\h], s[c5_LZM]
  s_lshl_b32 s[c5_T0], s[\h], 3
  s_add_u32 s[c5_T0], s[c5_T0], \soff
  v_mov_b32 v[\la], s[c5_T0]
  global_load_dwordx2 v[\e:\e+1], v[\la], s[c5_ARENA:c5_ARENA+1] sc1
  s_waitcnt vmcnt(0)
  v_readfirstlane_b32 s[c5_T1], v[\e+1]
  v_readfirstlane_b32 s[c5_T3], v[\e]
  s_cmp_eq_u32 s[c5_T1], 0
  s_cbranch_scc1 .Lc5_pr_dflt_\u\()_\@
  s_cmp_eq_u32 s[c5_T1], s[c5_T2]
  s_cbranch_scc0 .Lc5_pr_loop_\u\()_\@
  s_mov_b32 s[\c], s[c5_T3]
  s_branch .Lc5_pr_end_\u\()_\@
.Lc5_pr_dflt_\u\()_\@:
  s_mov_b32 s[\c], \dflt
.Lc5_pr_end_\u\()_\@:
.endm
; ---------------------------------------------------------------------------------------------------------------------
; One coding step, cr-ppm.c:169-235 + cr-coder.c:261-289, in two variants: sp = 1 the node is a line of pairs (PP),
; sp = 0 it is a dense slot (W, its dwords at VDA). u = the statement's unique label suffix, esc = the block's escape byte.
.macro c5_step sp, u, esc, ds
.Lc5_node_ok_\sp\()_\u:
  v_readfirstlane_b32 s[c5_O3E], v[c5_FE]
  s_cmp_eq_u32 s[c5_K3N], s[c5_O3LK]
  s_cselect_b32 s[c5_O3E], s[c5_O3LV], s[c5_O3E]   ; loaded before the previous step's store
  s_and_b32 s[c5_T1], s[c5_O3E], 0xf0
  s_cmp_eq_u32 s[c5_T1], s[c5_G3S]
  s_cselect_b32 s[c5_O3E], s[c5_O3E], 0            ; stale generation: the reference's zero-filled entry
  s_lshr_b32 s[c5_PRED], s[c5_O3E], 8
.if \sp && c5_hw
  ; the post for the helper wave: pairs, row (as the escape path would take it: the previous step's update if it is the same
  ; row), then the word that names them
  s_add_u32 s[c5_HWSEQ], s[c5_HWSEQ], 0x200
  s_and_b32 s[c5_ROWI], s[c5_CTX], 0xff
  v_mov_b32 v[c5_ROW], v[c5_FROW]
  s_cmp_eq_u32 s[c5_ROWI], s[c5_LRIDX]
  s_cbranch_scc1 .Lc5_hw_rowsame_\u
.Lc5_hw_row_ok_\u:
  ds_write_b16 v[c5_HWA2], v[c5_PP] offset:16
  ds_write_b32 v[c5_HWA4], v[c5_ROW] offset:144
  s_or_b32 s[c5_T1], s[c5_HWSEQ], s[c5_PRED]
  v_mov_b32 v[c5_VT1], s[c5_T1]
  s_mov_b32 s[c5_HWPOST], 1
  ds_write_b32 v[c5_HWB], v[c5_VT1]
.endif
.if \sp == 0
  s_mov_b32 s[c5_K3], s[c5_K3N]                    ; the key the order-3 entry was loaded with
  s_and_b32 s[c5_CONF], s[c5_O3E], 15
.endif
  ; ---------------------------------------------------------------- ppm_decode, cr-ppm.c:169-235
.if \sp
  v_and_b32 v[c5_CX], v[c5_VMCNT], v[c5_PP]            ; the pairs' counts (lanes 62 / 63: 0)
  v_lshrrev_b32 v[c5_VSYM], 8, v[c5_PP]
  v_or_b32 v[c5_VSYM], v[c5_VHI], v[c5_VSYM]           ; the pairs' symbols (lanes 62 / 63: above any byte)
  v_cmp_ne_u32 vcc, s[c5_PRED], v[c5_VSYM]
  s_and_b32 s[c5_FHIT], s[c5_SX], 0xff                 ; (every gap of the scan below carries instructions that do not depend on it)
  s_lshr_b32 s[c5_FESC], s[c5_SX], 8
  v_cndmask_b32 v[c5_SUM], 0, v[c5_CX], vcc            ; counts with the predicted byte taken out (cr-o2model.c:97)
  s_sub_u32 s[c5_T0], s[c5_HAVE], s[c5_LEARNED]
  v_cvt_f64_u32 v[c5_DN:c5_DN+1], v[c5_VRANGE]         ; (c5_vdiv_pre)
  v_add_u32_dpp v[c5_INCL], v[c5_SUM], v[c5_SUM] row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1
.if c5_pf & 1
  s_and_b32 s[c5_T2], s[c5_CTX], 0xff              ; the successors' lines: (last byte << 8 | symbol) << 7
  s_lshl_b32 s[c5_T2], s[c5_T2], 15
  s_lshl_b32 s[c5_T3], s[c5_PRED], 7
  s_add_u32 s[c5_T3], s[c5_T3], s[c5_T2]
.endif
  v_add_f64 v[c5_DN:c5_DN+1], v[c5_DN:c5_DN+1], 0.5
  v_mov_b32 v[c5_VFHIT], s[c5_FHIT]
.else
  s_and_b32 s[c5_T0], s[c5_PRED], 3
  s_lshl_b32 s[c5_T0], s[c5_T0], 3
  s_lshl_b32 s[c5_PM], 0xff, s[c5_T0]
  s_lshr_b32 s[c5_T1], s[c5_PRED], 2
  v_cmp_eq_u32 vcc, s[c5_T1], v[c5_LANE]
  v_mov_b32 v[c5_VT0], s[c5_PM]
  s_and_b32 s[c5_FHIT], s[c5_SX], 0xff
  s_lshr_b32 s[c5_FESC], s[c5_SX], 8
  v_cndmask_b32 v[c5_VPM], 0, v[c5_VT0], vcc       ; the predicted byte's place in its lane's word
  v_bfi_b32 v[c5_WX], v[c5_VPM], 0, v[c5_W]        ; counts with the predicted byte taken out (cr-o2model.c:97)
  v_sad_u8 v[c5_SUM], v[c5_WX], 0, 0
  v_mov_b32 v[c5_VFHIT], s[c5_FHIT]                ; (these two are the first scan step's wait states)
  v_mov_b32 v[c5_VFESC], s[c5_FESC]
  v_add_u32_dpp v[c5_INCL], v[c5_SUM], v[c5_SUM] row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1
  v_add_u32 v[c5_VHE], v[c5_VFHIT], v[c5_VFESC]
  s_sub_u32 s[c5_T0], s[c5_HAVE], s[c5_LEARNED]
  c5_vdiv_pre
.endif
  ; (the scan's wait states carry work that does not depend on the symbol: the position about to be decoded becomes
  ; pending with the 8 bytes in front of it - if the token turns out not to be a literal the lane is simply written again)
  v_add_u32_dpp v[c5_INCL], v[c5_INCL], v[c5_INCL] row_shr:2 row_mask:0xf bank_mask:0xf bound_ctrl:1
.if c5_mode != 1
  s_lshl_b64 exec, 1, s[c5_T0]
  v_mov_b32 v[c5_PENDLO], s[c5_X8LO]
  v_mov_b32 v[c5_PENDHI], s[c5_X8HI]
  s_mov_b64 exec, -1
.else
  s_nop 1
.endif
  v_add_u32_dpp v[c5_INCL], v[c5_INCL], v[c5_INCL] row_shr:4 row_mask:0xf bank_mask:0xf bound_ctrl:1
.if \sp
  s_mov_b32 s[c5_NDNO], s[c5_NO]
.else
  s_or_b32 s[c5_NDNO], s[c5_NO], 1                 ; (bit 0: the registers hold a dense node)
.endif
  s_mov_b32 s[c5_HALV], 0
  v_add_u32_dpp v[c5_INCL], v[c5_INCL], v[c5_INCL] row_shr:8 row_mask:0xf bank_mask:0xf bound_ctrl:1
.if \sp && (c5_pf & 1)
  v_mov_b32 v[c5_VFESC], s[c5_FESC]
  v_add_u32 v[c5_VHE], v[c5_VFHIT], v[c5_VFESC]
  s_mov_b32 s[c5_K3], s[c5_K3N]
  s_and_b32 s[c5_CONF], s[c5_O3E], 15
  v_cmp_le_u32_e64 s[c5_T4:c5_T4+1], c5_pfthr, v[c5_CX]
  v_lshl_add_u32 v[c5_PFA], v[c5_VSYM], 7, s[c5_T2]
  v_add_u32_dpp v[c5_INCL], v[c5_INCL], v[c5_INCL] row_bcast:15 row_mask:0xa bank_mask:0xf
  v_writelane_b32 v[c5_PFA], s[c5_T3], 63          ; (lane 63: the predicted byte's line)
  s_bitset1_b32 s[c5_T5], 31
  v_add_u32_dpp v[c5_INCL], v[c5_INCL], v[c5_INCL] row_bcast:31 row_mask:0xc bank_mask:0xf
  s_mov_b64 exec, s[c5_T4:c5_T4+1]
  v_readlane_b32 s[c5_BYTES], v[c5_INCL], 63
  global_load_ubyte v[c5_PFD], v[c5_PFA], s[c5_BN:c5_BN+1]
  s_mov_b64 exec, -1
.elseif \sp
  v_mov_b32 v[c5_VFESC], s[c5_FESC]
  v_add_u32 v[c5_VHE], v[c5_VFHIT], v[c5_VFESC]
  v_add_u32_dpp v[c5_INCL], v[c5_INCL], v[c5_INCL] row_bcast:15 row_mask:0xa bank_mask:0xf
  s_mov_b32 s[c5_K3], s[c5_K3N]                    ; the key the order-3 entry was loaded with
  s_and_b32 s[c5_CONF], s[c5_O3E], 15
  v_add_u32_dpp v[c5_INCL], v[c5_INCL], v[c5_INCL] row_bcast:31 row_mask:0xc bank_mask:0xf
  s_nop 0
  v_readlane_b32 s[c5_BYTES], v[c5_INCL], 63
.else
  s_nop 1
  v_add_u32_dpp v[c5_INCL], v[c5_INCL], v[c5_INCL] row_bcast:15 row_mask:0xa bank_mask:0xf
  s_nop 1
  v_add_u32_dpp v[c5_INCL], v[c5_INCL], v[c5_INCL] row_bcast:31 row_mask:0xc bank_mask:0xf
  s_nop 0
  v_readlane_b32 s[c5_BYTES], v[c5_INCL], 63
.endif
.if \sp
  v_add_u32 v[c5_VTOT], s[c5_BYTES], v[c5_VHE]
  c5_vdiv c5_VUNIT, s_nop 0
.else
  ; the four cumulative counts inside every lane's word: EXCL | C1 | C2 | C3 | INCL
  v_sub_u32 v[c5_EXCL], v[c5_INCL], v[c5_SUM]
  v_add_u32_sdwa v[c5_C1], v[c5_EXCL], v[c5_WX] dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_0
  v_add_u32 v[c5_VTOT], s[c5_BYTES], v[c5_VHE]
  v_add_u32_sdwa v[c5_C2], v[c5_C1], v[c5_WX] dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_1
  c5_vdiv c5_VUNIT, v_add_u32_sdwa v[c5_C3], v[c5_C2], v[c5_WX] dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_2
.endif
.if \sp
  ; a byte of the node: its pair = how many of the pairs' cumulative counts x unit do not exceed cache (a pair with
  ; count 0 repeats the boundary below it and is stepped over, o2_model_get_decode_symbol, cr-o2model.c:93-113). Lanes 62 / 63
  ; hold the byte total (their counts are 0), so ALL 64 lanes pass exactly when cache lies above the bytes (symbol 256 or 257):
  ; one ballot answers both questions, and the vector-to-scalar hop is paid once (round 5; before: a compare against
  ; unit x bytes and a branch on vcc in front of this one)
  v_mul_lo_u32 v[c5_P], v[c5_INCL], v[c5_VUNIT]
  v_cmp_ge_u32 vcc, v[c5_VCACHE], v[c5_P]
  s_nop 0
  s_bcnt1_i32_b64 s[c5_OL], vcc
  s_cmp_eq_u32 s[c5_OL], 64
  s_cbranch_scc1 .Lc5_not_in_node_\sp\()_\u
  s_nop 0
  v_readlane_b32 s[c5_SS], v[c5_VSYM], s[c5_OL]
.else
  v_mul_lo_u32 v[c5_VTB], v[c5_VUNIT], s[c5_BYTES]
  v_mul_lo_u32 v[c5_P], v[c5_INCL], v[c5_VUNIT]
  v_cmp_lt_u32 vcc, v[c5_VCACHE], v[c5_VTB]
  s_cbranch_vccz .Lc5_not_in_node_\sp\()_\u
  ; a byte of the node: its index = how many of the 256 cumulative counts x unit do not exceed cache (zero counts
  ; repeat the boundary below them and are stepped over, o2_model_get_decode_symbol, cr-o2model.c:93-113)
  v_mul_lo_u32 v[c5_C1], v[c5_C1], v[c5_VUNIT]
  v_mul_lo_u32 v[c5_C2], v[c5_C2], v[c5_VUNIT]
  v_mul_lo_u32 v[c5_C3], v[c5_C3], v[c5_VUNIT]
  v_cmp_ge_u32 vcc, v[c5_VCACHE], v[c5_P]
  v_cmp_ge_u32_e64 s[c5_T0:c5_T0+1], v[c5_VCACHE], v[c5_C1]
  v_cmp_ge_u32_e64 s[c5_T2:c5_T2+1], v[c5_VCACHE], v[c5_C2]
  v_cmp_ge_u32_e64 s[c5_T4:c5_T4+1], v[c5_VCACHE], v[c5_C3]
  s_bcnt1_i32_b64 s[c5_SS], vcc
  s_bcnt1_i32_b64 s[c5_T0], s[c5_T0:c5_T0+1]
  s_bcnt1_i32_b64 s[c5_T2], s[c5_T2:c5_T2+1]
  s_bcnt1_i32_b64 s[c5_T4], s[c5_T4:c5_T4+1]
  s_add_u32 s[c5_SS], s[c5_SS], s[c5_T0]
  s_add_u32 s[c5_T2], s[c5_T2], s[c5_T4]
  s_add_u32 s[c5_SS], s[c5_SS], s[c5_T2]
.endif
  ; ---------------------------------------------------------------- a byte of the node
.Lc5_consume_\sp\()_\u:
  s_mov_b32 s[c5_SYM], s[c5_SS]
  s_cmp_lg_u32 s[c5_AESC], 0
  s_cbranch_scc1 .Lc5_late_\sp\()_\u
  ; the common case (no escape byte pending): the next context is ctx << 8 | symbol whatever the symbol
  ; means, so the next step's loads go out before the coder state is even advanced
  s_lshl_b32 s[c5_NCTX], s[c5_CTX], 8
  s_or_b32 s[c5_NCTX], s[c5_NCTX], s[c5_SYM]
  c5_issue c5_NCTX, c5_T3, c5_T5, c5_T6, 10
.if \sp
  c5_pick_sp \u, 0
.else
  c5_pick \u, 0
.endif
  c5_consume c5_VUNIT
  s_cbranch_vccnz .Lc5_refill_a_\sp\()_\u
.Lc5_refilled_a_\sp\()_\u:
  s_mov_b32 s[c5_LRIDX], -1
  ; ---------------------------------------------------------------- what the symbol means, cr-coder.c:261-289
  s_mov_b32 s[c5_LIT], s[c5_SYM]
  s_cmp_eq_u32 s[c5_SYM], \esc
  s_cbranch_scc1 .Lc5_early_esc_\sp\()_\u
  c5_literal                                       ; a literal byte at `have`: pending LZP position, the 8 bytes in front
  ; ---------------------------------------------------------------- model updates, cr-ppm.c:199-232
.Lc5_upd_node_\sp\()_\u:
  s_lshl_b64 s[c5_MW:c5_MW+1], 1, s[c5_OL]
.if \sp
  s_mov_b64 exec, s[c5_MW:c5_MW+1]                 ; o2_model_update(sym, +1): the pair's lane as the search left it
  v_add_u32 v[c5_PP], 1, v[c5_PP]
.else
  s_lshl_b32 s[c5_T0], 1, s[c5_LOWER]              ; o2_model_update(sym, +1): lane and shift as c5_pick left them
  s_mov_b64 exec, s[c5_MW:c5_MW+1]
  v_add_u32 v[c5_W], s[c5_T0], v[c5_W]
.endif
  s_mov_b64 exec, -1
  s_cmp_ge_u32 s[c5_FRQ], 250
  s_cbranch_scc1 .Lc5_upd_halve_\sp\()_\u
  s_cmp_eq_u32 s[c5_FRQ], 1
  s_cbranch_scc1 .Lc5_upd_single_\sp\()_\u
  c5_o3_miss                                       ; the common case: hit / escape counts unchanged
.if \sp
  c5_st_pairs
.else
  c5_st_node
.endif
  c5_st_o3_lit
  c5_tail 4, \u

.if c5_align & 2
  .p2align 6
.endif
.Lc5_not_in_node_\sp\()_\u:                        ; symbol 256 (prediction hit) or 257 (escape)
.if \sp
  v_mul_lo_u32 v[c5_VTB], v[c5_VUNIT], s[c5_BYTES]   ; (unit x bytes: only the hit and the escape need it)
  ; an escape will want to know which bytes the node holds, in the order-1 row's layout (lane l = bytes 4l .. 4l + 3): the
  ; pairs' symbols are scattered through the wave's 256 bytes of LDS (all zero between steps) — set, read back, cleared
  ; again, three operations that go out together now and have come back when the escape path needs them
  ; (a lane without a pair — count 0 — aims at the spare byte behind the 256: no exec juggling)
.if c5_hw
  v_mul_lo_u32 v[c5_VLOWU], v[c5_VUNIT], v[c5_VFHIT]
  v_add_u32 v[c5_VLOWU], v[c5_VTB], v[c5_VLOWU]
  s_cmp_lg_u32 s[c5_HWPOST], 0                     ; (the helper wave has the pairs: it does the scatter)
  s_cbranch_scc1 .Lc5_hw_noscatter_\u
  v_add_u32 v[c5_VT3], v[c5_VLDB], v[c5_VSYM]
  v_cmp_ne_u32 vcc, 0, v[c5_CX]
  s_nop 1
  v_cndmask_b32 v[c5_VT3], v[c5_VLDX], v[c5_VT3], vcc
  ds_write_b8 v[c5_VT3], v[c5_VONE]
  ds_read_b32 v[c5_PRES], v[c5_VLDZ]
  ds_write_b8 v[c5_VT3], v[c5_VZERO]
.Lc5_hw_noscatter_\u:
.else
  v_add_u32 v[c5_VT3], v[c5_VLDB], v[c5_VSYM]
  v_cmp_ne_u32 vcc, 0, v[c5_CX]
  v_mul_lo_u32 v[c5_VLOWU], v[c5_VUNIT], v[c5_VFHIT]
  v_add_u32 v[c5_VLOWU], v[c5_VTB], v[c5_VLOWU]      ; (the byte counts + the hit count) x unit
  v_cndmask_b32 v[c5_VT3], v[c5_VLDX], v[c5_VT3], vcc
  ds_write_b8 v[c5_VT3], v[c5_VONE]
  ds_read_b32 v[c5_PRES], v[c5_VLDZ]
  ds_write_b8 v[c5_VT3], v[c5_VZERO]
.endif
.else
  v_mul_lo_u32 v[c5_VLOWU], v[c5_VUNIT], v[c5_VFHIT]
  v_add_u32 v[c5_VLOWU], v[c5_VTB], v[c5_VLOWU]      ; (the byte counts + the hit count) x unit
.endif
.Lc5_nin_test_\sp\()_\u:                           ; (a first-use node comes in here: nothing to scatter, unit x 1 is the boundary)
  v_cmp_lt_u32 vcc, v[c5_VCACHE], v[c5_VLOWU]
  s_cbranch_vccnz .Lc5_hit_\sp\()_\u
  ; ---------------------------------------------------------------- escape: order-1 step with exclusion, cr-ppm.c:209-232
  ; (the three kinds of step each run straight through to their own stores: a taken branch costs six instructions)
  s_movk_i32 s[c5_SS], 0x101
  c5_consume c5_VUNIT, c5_VLOWU, c5_VFESC
  s_cbranch_vccnz .Lc5_refill_e_\sp\()_\u
.Lc5_esc_start_\sp\()_\u:
  s_and_b32 s[c5_ROWI], s[c5_CTX], 0xff
  s_add_u32 s[c5_SX], s[c5_SX], 0x100              ; count(257) + 1, a byte (a node of 254 singletons leaves 255 behind a halving)
  s_and_b32 s[c5_SX], s[c5_SX], 0xffff
  s_lshr_b32 s[c5_T0], s[c5_SX], 8
  s_cmp_gt_u32 s[c5_T0], 250
  s_cbranch_scc1 .Lc5_esc_halve_\sp\()_\u
.Lc5_esc_go_\sp\()_\u:
.if \sp && c5_hw
  s_cmp_eq_u32 s[c5_HWPOST], 0                     ; nothing posted (a first-use node), or the node was just halved (its
  s_cbranch_scc1 .Lc5_esc_own_\u                   ; exclusion set is no longer what was posted): the coder's own path
  s_cmp_lg_u32 s[c5_HALV], 0
  s_cbranch_scc1 .Lc5_esc_own_\u
  c5_vdiv_pre
.Lc5_hw_poll_\u:
  ds_read_b32 v[c5_VT0], v[c5_HWB] offset:4
  s_waitcnt lgkmcnt(0)
  v_readfirstlane_b32 s[c5_T0], v[c5_VT0]
  s_cmp_lg_u32 s[c5_T0], s[c5_HWSEQ]
  s_cbranch_scc1 .Lc5_hw_poll_\u
  ds_read_b128 v[c5_HWR:c5_HWR+3], v[c5_HWA16] offset:400
  ds_read_b32 v[c5_VT1], v[c5_HWB] offset:8
  s_waitcnt lgkmcnt(0)
  v_mov_b32 v[c5_EXCL], v[c5_HWR]
  v_mov_b32 v[c5_FE], v[c5_HWR+1]
  v_mov_b32 v[c5_FO], v[c5_HWR+2]
  v_mov_b32 v[c5_INCL1], v[c5_HWR+3]
  v_readfirstlane_b32 s[c5_T4], v[c5_VT1]
  s_branch .Lc5_esc_sums_\u
.Lc5_esc_own_\u:
.endif
.if \sp
  s_and_b32 s[c5_T0], s[c5_PRED], 3                ; the predicted byte's place in its lane's word
  s_lshl_b32 s[c5_T0], s[c5_T0], 3
  s_lshl_b32 s[c5_PM], 0xff, s[c5_T0]
  s_lshr_b32 s[c5_T1], s[c5_PRED], 2
  v_cmp_eq_u32 vcc, s[c5_T1], v[c5_LANE]
  v_mov_b32 v[c5_VT0], s[c5_PM]
  s_nop 0
  v_cndmask_b32 v[c5_VPM], 0, v[c5_VT0], vcc
.endif
  s_cmp_eq_u32 s[c5_AESC], 1
  s_cbranch_scc1 .Lc5_esc_go_lzp_\sp\()_\u
  c5_prof_begin 20, c5_HWSEQ                       ; (20: an escape's wait for its order-1 row; 21: for the presence bytes out of LDS)
  s_waitcnt vmcnt(3 + ((c5_pf >> 1) & 1) + (\sp & c5_pf & 1))   ; this context's order-1 row (at least three stores went out behind it, and the prefetches)
  c5_prof_end 20, c5_HWSEQ
.Lc5_esc_row_in_\sp\()_\u:
  v_mov_b32 v[c5_ROW], v[c5_FROW]
  s_cmp_eq_u32 s[c5_ROWI], s[c5_LRIDX]
  s_cbranch_scc1 .Lc5_esc_rowsame_\sp\()_\u
.Lc5_esc_row_ok_\sp\()_\u:
.if \sp
.if c5_prof == 21
  s_memtime s[c5_HWSEQ:c5_HWSEQ+1]
.endif
  s_waitcnt lgkmcnt(0)
  c5_prof_end 21, c5_HWSEQ
  v_xor_b32 v[c5_VT0], 0x01010101, v[c5_PRES]      ; 0x01 in every byte the node does not hold ...
.else
  v_and_b32 v[c5_VT0], 0x7f7f7f7f, v[c5_W]         ; 0x01 in every byte of W that is zero ...
  v_add_u32 v[c5_VT0], 0x7f7f7f7f, v[c5_VT0]
  v_or_b32 v[c5_VT0], v[c5_VT0], v[c5_W]
  v_or_b32 v[c5_VT0], 0x7f7f7f7f, v[c5_VT0]
  v_not_b32 v[c5_VT0], v[c5_VT0]
  v_lshrrev_b32 v[c5_VT0], 7, v[c5_VT0]
.endif
  v_bfi_b32 v[c5_VT0], v[c5_VPM], 0, v[c5_VT0]     ; ... except the predicted byte: the candidates (cr-ppm.c:150-155)
  v_lshlrev_b32 v[c5_VT1], 8, v[c5_VT0]
  v_sub_u32 v[c5_KEEP], v[c5_VT1], v[c5_VT0]       ; x * 255: 0x01 -> 0xff in every byte
  v_and_b32 v[c5_ROWK], v[c5_ROW], v[c5_KEEP]
  v_sub_u32 v[c5_ROWK], v[c5_ROWK], v[c5_VT0]      ; count - 1 of every candidate (order-1 counts never drop below 1)
  v_and_b32 v[c5_FE], 0x00ff00ff, v[c5_ROWK]       ; bytes 0 and 2, bytes 1 and 3 as 16-bit fields
  v_lshrrev_b32 v[c5_FO], 8, v[c5_ROWK]
  v_and_b32 v[c5_VT1], 0x00ff00ff, v[c5_VT0]
  v_lshrrev_b32 v[c5_VT0], 8, v[c5_VT0]
  v_and_b32 v[c5_FO], 0x00ff00ff, v[c5_FO]
  v_and_b32 v[c5_VT0], 0x00ff00ff, v[c5_VT0]
  v_lshl_add_u32 v[c5_FE], v[c5_FE], 3, v[c5_VT1]  ; 8 (c - 1) + 1 = 8c - 7 per candidate (cr-ppm.c:98), 0 elsewhere
  v_lshl_add_u32 v[c5_FO], v[c5_FO], 3, v[c5_VT0]
  v_add_u32 v[c5_VT0], v[c5_FE], v[c5_FO]
  v_add_u32_sdwa v[c5_MINE], v[c5_VT0], v[c5_VT0] dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_0 src1_sel:WORD_1
  c5_vdiv_pre                                      ; (the scan's first wait states; the range is the escape's)
  v_add_u32_dpp v[c5_INCL1], v[c5_MINE], v[c5_MINE] row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1
  s_nop 1
  v_add_u32_dpp v[c5_INCL1], v[c5_INCL1], v[c5_INCL1] row_shr:2 row_mask:0xf bank_mask:0xf bound_ctrl:1
  s_nop 1
  v_add_u32_dpp v[c5_INCL1], v[c5_INCL1], v[c5_INCL1] row_shr:4 row_mask:0xf bank_mask:0xf bound_ctrl:1
  s_nop 1
  v_add_u32_dpp v[c5_INCL1], v[c5_INCL1], v[c5_INCL1] row_shr:8 row_mask:0xf bank_mask:0xf bound_ctrl:1
  s_nop 1
  v_add_u32_dpp v[c5_INCL1], v[c5_INCL1], v[c5_INCL1] row_bcast:15 row_mask:0xa bank_mask:0xf
  s_nop 1
  v_add_u32_dpp v[c5_INCL1], v[c5_INCL1], v[c5_INCL1] row_bcast:31 row_mask:0xc bank_mask:0xf
  s_nop 0
  v_readlane_b32 s[c5_T4], v[c5_INCL1], 63
  ; the four cumulative sums inside every lane: EXCL | C1 | C2 | C3 | INCL1
  v_sub_u32 v[c5_EXCL], v[c5_INCL1], v[c5_MINE]
.if \sp && c5_hw
.Lc5_esc_sums_\u:
.endif
  v_add_u32_sdwa v[c5_C1], v[c5_EXCL], v[c5_FE] dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:WORD_0
  v_mov_b32 v[c5_VTOT], s[c5_T4]
  v_add_u32_sdwa v[c5_C2], v[c5_C1], v[c5_FO] dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:WORD_0
  c5_vdiv c5_VUNIT1, v_add_u32_sdwa v[c5_C3], v[c5_C2], v[c5_FE] dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:WORD_1
  v_mul_lo_u32 v[c5_P], v[c5_INCL1], v[c5_VUNIT1]
  ; the symbol = how many of the 256 cumulative sums x unit do not exceed cache; P0 = the largest of them in every lane.
  ; (All 256 of them — the last one is total x unit — only with a damaged stream: tested on the count, behind the search; round 5,
  ; before: a multiplication, a compare and a branch on vcc in front of it)
  v_mul_lo_u32 v[c5_C1], v[c5_C1], v[c5_VUNIT1]
  v_mul_lo_u32 v[c5_C2], v[c5_C2], v[c5_VUNIT1]
  v_mul_lo_u32 v[c5_C3], v[c5_C3], v[c5_VUNIT1]
  v_cmp_ge_u32 vcc, v[c5_VCACHE], v[c5_P]
  v_cmp_ge_u32_e64 s[c5_T0:c5_T0+1], v[c5_VCACHE], v[c5_C1]
  v_cmp_ge_u32_e64 s[c5_T2:c5_T2+1], v[c5_VCACHE], v[c5_C2]
  v_cmp_ge_u32_e64 s[c5_T4:c5_T4+1], v[c5_VCACHE], v[c5_C3]
  s_bcnt1_i32_b64 s[c5_SYM], vcc
  s_bcnt1_i32_b64 s[c5_T6], s[c5_T0:c5_T0+1]
  s_bcnt1_i32_b64 s[c5_T7], s[c5_T2:c5_T2+1]
  s_bcnt1_i32_b64 s[c5_LOWER], s[c5_T4:c5_T4+1]
  s_add_u32 s[c5_SYM], s[c5_SYM], s[c5_T6]
  s_add_u32 s[c5_T7], s[c5_T7], s[c5_LOWER]
  s_add_u32 s[c5_SYM], s[c5_SYM], s[c5_T7]
  s_cmp_ge_u32 s[c5_SYM], 0x100
  s_cbranch_scc1 .Lc5_esc_corrupt_\sp\()_\u
.Lc5_esc_consume_\sp\()_\u:
  s_cmp_lg_u32 s[c5_AESC], 0
  s_cbranch_scc1 .Lc5_esc_noissue_\sp\()_\u
  s_lshl_b32 s[c5_NCTX], s[c5_CTX], 8
  s_or_b32 s[c5_NCTX], s[c5_NCTX], s[c5_SYM]
  c5_issue c5_NCTX, c5_T6, c5_T7, c5_LOWER, 12     ; (T0 .. T5 keep the search's three lane masks for P0 below)
.Lc5_esc_noissue_\sp\()_\u:
  ; P0 = the largest sum x unit below the cache in every lane: only the coder's advance wants it, so it waits until the loads are out
  ; (92 of 100 steps at 1 526 blocks end in a wait: what comes behind the loads is in its shadow)
  v_mul_lo_u32 v[c5_P0], v[c5_EXCL], v[c5_VUNIT1]
  v_cndmask_b32_e64 v[c5_P0], v[c5_P0], v[c5_C1], s[c5_T0:c5_T0+1]
  v_cndmask_b32_e64 v[c5_P0], v[c5_P0], v[c5_C2], s[c5_T2:c5_T2+1]
  v_cndmask_b32_e64 v[c5_P0], v[c5_P0], v[c5_C3], s[c5_T4:c5_T4+1]
  s_lshr_b32 s[c5_SL], s[c5_SYM], 2                ; the symbol's lane (SL) and its byte's shift (LOWER), kept for the
  s_and_b32 s[c5_LOWER], s[c5_SYM], 3              ; updates: (the sum below it) x unit, its order-1 count
  s_lshl_b32 s[c5_LOWER], s[c5_LOWER], 3
  v_readlane_b32 s[c5_T3], v[c5_P0], s[c5_SL]
  v_readlane_b32 s[c5_T2], v[c5_ROW], s[c5_SL]
  s_lshr_b32 s[c5_T2], s[c5_T2], s[c5_LOWER]
  s_and_b32 s[c5_T2], s[c5_T2], 0xff
  s_lshl_b32 s[c5_FRQ], s[c5_T2], 3
  s_sub_u32 s[c5_FRQ], s[c5_FRQ], 7
  v_mov_b32 v[c5_VLOWU], s[c5_T3]
  v_mov_b32 v[c5_VFRQ], s[c5_FRQ]
  c5_consume c5_VUNIT1
  s_cbranch_vccnz .Lc5_refill_b_\sp\()_\u
.Lc5_refilled_b_\sp\()_\u:
  s_lshl_b32 s[c5_T3], 1, s[c5_LOWER]              ; ppm_update_o1, cr-ppm.c:90-97
  v_mov_b32 v[c5_ROWU], v[c5_ROW]
  s_lshl_b64 exec, 1, s[c5_SL]
  v_add_u32 v[c5_ROWU], s[c5_T3], v[c5_ROWU]
  s_mov_b64 exec, -1
  s_cmp_ge_u32 s[c5_T2], 254
  s_cbranch_scc1 .Lc5_esc_rescale_\sp\()_\u
.Lc5_esc_done_\sp\()_\u:
  s_mov_b32 s[c5_LRIDX], s[c5_ROWI]
  s_cmp_lg_u32 s[c5_AESC], 0
  s_cbranch_scc1 .Lc5_tok_after_\sp\()_\u
  s_mov_b32 s[c5_LIT], s[c5_SYM]
  s_cmp_eq_u32 s[c5_SYM], \esc
  s_cbranch_scc1 .Lc5_early_esc_\sp\()_\u
  c5_literal                                       ; a literal byte at `have`: pending LZP position, the 8 bytes in front
.Lc5_upd_esc_\sp\()_\u:                            ; cr-ppm.c:160-162: the new byte enters the node unless it was just halved
  s_cmp_lg_u32 s[c5_HALV], 0
  s_cbranch_scc1 .Lc5_upd_esc_halved_\sp\()_\u
.if \sp
  ; a pair that holds the byte with count 0 (left by a halving; the unused pairs read as byte 0xff with count 0) is raised,
  ; else the pair is put in at its place in symbol order; a full line becomes a dense node first
  v_cmp_eq_u32 vcc, s[c5_SYM], v[c5_VSYM]
  s_nop 0
  s_cmp_lg_u64 vcc, 0
  s_cbranch_scc1 .Lc5_ins_found_\u
  v_readlane_b32 s[c5_T0], v[c5_PP], 61
  s_cmp_lg_u32 s[c5_T0], 0xff00
  s_cbranch_scc1 .Lc5_convert_\u
  v_cmp_gt_u32 vcc, s[c5_SYM], v[c5_VSYM]
  s_lshl_b32 s[c5_T6], s[c5_SYM], 8
  s_or_b32 s[c5_T6], s[c5_T6], 1
  s_bcnt1_i32_b64 s[c5_T0], vcc                    ; its place: the pairs with a smaller symbol
  s_lshl_b64 s[c5_T2:c5_T2+1], 1, s[c5_T0]
  s_sub_u32 s[c5_T4], s[c5_T2], 1
  s_subb_u32 s[c5_T5], s[c5_T3], 0
  s_not_b64 s[c5_T4:c5_T4+1], s[c5_T4:c5_T4+1]
  s_and_b32 s[c5_T5], s[c5_T5], 0x3fffffff          ; lanes place .. 61: the pairs that move up by one
  v_mov_b32_dpp v[c5_VT2], v[c5_PP] wave_shr:1 row_mask:0xf bank_mask:0xf
  s_mov_b64 exec, s[c5_T4:c5_T4+1]
  v_mov_b32 v[c5_PP], v[c5_VT2]
  s_mov_b64 exec, s[c5_T2:c5_T2+1]
  v_mov_b32 v[c5_PP], s[c5_T6]
  s_mov_b64 exec, -1
  s_mov_b64 s[c5_MW:c5_MW+1], s[c5_T4:c5_T4+1]
  s_branch .Lc5_upd_esc_st_\sp\()_\u
.Lc5_ins_found_\u:
  s_ff1_i32_b64 s[c5_T0], vcc
  s_lshl_b64 s[c5_MW:c5_MW+1], 1, s[c5_T0]
  s_mov_b64 exec, s[c5_MW:c5_MW+1]
  v_add_u32 v[c5_PP], 1, v[c5_PP]
  s_mov_b64 exec, -1
.else
  s_lshl_b32 s[c5_T0], 1, s[c5_LOWER]              ; the new byte's first count: lane and shift as the order-1 step left them
  s_lshl_b64 s[c5_MW:c5_MW+1], 1, s[c5_SL]
  s_mov_b64 exec, s[c5_MW:c5_MW+1]
  v_add_u32 v[c5_W], s[c5_T0], v[c5_W]
  s_mov_b64 exec, -1
.endif
.Lc5_upd_esc_st_\sp\()_\u:
  c5_o3_miss
.if \sp
  v_writelane_b32 v[c5_PP], s[c5_SX], 62
  s_bitset1_b32 s[c5_MW+1], 30
  c5_st_pairs
  c5_st_o3_lit
  c5_st_row
  c5_tail 5, \u
.else
  c5_st_node
  c5_st_flag
  c5_st_o3_lit
  c5_st_row
  c5_tail 6, \u
.endif
  ; ---------------------------------------------------------------- the predicted byte
.if c5_align & 4
  .p2align 6
.endif
.Lc5_hit_\sp\()_\u:
  s_movk_i32 s[c5_SS], 0x100
  s_mov_b32 s[c5_SYM], s[c5_PRED]
  s_cmp_lg_u32 s[c5_AESC], 0
  s_cbranch_scc1 .Lc5_late_hit_\sp\()_\u
  s_lshl_b32 s[c5_NCTX], s[c5_CTX], 8
  s_or_b32 s[c5_NCTX], s[c5_NCTX], s[c5_SYM]
  c5_issue c5_NCTX, c5_T0, c5_T1, c5_T2, 11
  c5_consume c5_VUNIT, c5_VTB, c5_VFHIT
  s_cbranch_vccnz .Lc5_refill_h_\sp\()_\u
.Lc5_refilled_h_\sp\()_\u:
  s_mov_b32 s[c5_LRIDX], -1
  s_mov_b32 s[c5_LIT], s[c5_SYM]
  s_cmp_eq_u32 s[c5_SYM], \esc
  s_cbranch_scc1 .Lc5_early_esc_\sp\()_\u
  c5_literal                                       ; a literal byte at `have`: pending LZP position, the 8 bytes in front
.Lc5_upd_hit_\sp\()_\u:                            ; o2_model_update(256, +1); ppm_update_o3(-1), cr-ppm.c:81-83
  s_add_u32 s[c5_SX], s[c5_SX], 1
  s_and_b32 s[c5_T0], s[c5_SX], 0xff
  s_cmp_gt_u32 s[c5_T0], 250
  s_cbranch_scc1 .Lc5_upd_hit_halve_\sp\()_\u
  c5_o3_hit                                        ; no byte count changed: the node's counts stay as they are in memory
.if \sp
  v_writelane_b32 v[c5_PP], s[c5_SX], 62
  s_mov_b32 s[c5_MW], 0
  s_mov_b32 s[c5_MW+1], 0x40000000
  c5_st_pairs
.else
  s_mov_b64 exec, 1
  c5_st_flag
.endif
  c5_st_o3_lit
  c5_tail 4, \u
  ; ================================================================ out of line
.Lc5_update_\sp\()_\u:                             ; (from the rare tokens: any of the three kinds)
  s_cmp_eq_u32 s[c5_SS], 0x100
  s_cbranch_scc1 .Lc5_upd_hit_\sp\()_\u
  s_cmp_eq_u32 s[c5_SS], 0x101
  s_cbranch_scc1 .Lc5_upd_esc_\sp\()_\u
  s_branch .Lc5_upd_node_\sp\()_\u
.Lc5_upd_hit_halve_\sp\()_\u:
.if \sp
  c5_halve_sp
.else
  c5_halve
.endif
  s_mov_b64 s[c5_MW:c5_MW+1], -1
  c5_o3_hit
.if \sp
  v_writelane_b32 v[c5_PP], s[c5_SX], 62
  c5_st_pairs
  c5_st_o3_lit
  c5_tail 4, \u
.else
  c5_st_node
  c5_st_flag
  c5_st_o3_lit
  c5_tail 5, \u
.endif
.Lc5_upd_esc_halved_\sp\()_\u:
  s_mov_b64 s[c5_MW:c5_MW+1], -1
  s_branch .Lc5_upd_esc_st_\sp\()_\u
.Lc5_esc_halve_\sp\()_\u:
.if \sp
  c5_halve_sp
.if c5_hw
  v_add_u32 v[c5_VT3], v[c5_VLDB], v[c5_VSYM]      ; (with a helper wave the step's own scatter may have been left out)
.endif
  v_and_b32 v[c5_VT2], v[c5_VMCNT], v[c5_PP]       ; which bytes the node holds NOW: counts may have fallen to zero (cr-ppm.c:146-155)
  v_cmp_ne_u32 vcc, 0, v[c5_VT2]
  s_nop 0
  s_mov_b64 exec, vcc
  ds_write_b8 v[c5_VT3], v[c5_VONE]
  s_mov_b64 exec, -1
  ds_read_b32 v[c5_PRES], v[c5_VLDZ]
  s_mov_b64 exec, vcc
  ds_write_b8 v[c5_VT3], v[c5_VZERO]
  s_mov_b64 exec, -1
.else
  c5_halve
.endif
  s_mov_b32 s[c5_HALV], 1
  s_branch .Lc5_esc_go_\sp\()_\u
.Lc5_esc_go_lzp_\sp\()_\u:                         ; the match token's six table operations went out behind the stores
  s_waitcnt vmcnt(9)
  s_branch .Lc5_esc_row_in_\sp\()_\u
.if \sp && c5_hw
.Lc5_hw_rowsame_\u:
  v_mov_b32 v[c5_ROW], v[c5_ROWU]
  s_branch .Lc5_hw_row_ok_\u
.endif
.Lc5_esc_rowsame_\sp\()_\u:                        ; this row was stored by the previous step, after this step's load went out
  v_mov_b32 v[c5_ROW], v[c5_ROWU]
  s_branch .Lc5_esc_row_ok_\sp\()_\u
.Lc5_esc_corrupt_\sp\()_\u:                        ; only a damaged stream gets here: stay inside the tables
  s_mov_b32 s[c5_SYM], 0
  v_mov_b32 v[c5_P0], 0
  s_branch .Lc5_esc_consume_\sp\()_\u
.Lc5_esc_rescale_\sp\()_\u:
  v_lshrrev_b32 v[c5_VT0], 1, v[c5_ROWU]
  v_and_b32 v[c5_VT0], 0x7f7f7f7f, v[c5_VT0]
  v_sub_u32 v[c5_ROWU], v[c5_ROWU], v[c5_VT0]
  s_branch .Lc5_esc_done_\sp\()_\u
.Lc5_refill_a_\sp\()_\u:
  c5_refill
  s_branch .Lc5_refilled_a_\sp\()_\u
.Lc5_refill_e_\sp\()_\u:
  c5_refill
  s_branch .Lc5_esc_start_\sp\()_\u
.Lc5_refill_h_\sp\()_\u:
  c5_refill
  s_branch .Lc5_refilled_h_\sp\()_\u
.Lc5_refill_b_\sp\()_\u:
  c5_refill
  s_branch .Lc5_refilled_b_\sp\()_\u
.Lc5_early_esc_\sp\()_\u:                          ; the escape byte: a match length or a 0 follows
.if c5_mode == 0
  s_mov_b32 s[c5_AESC], 1
  s_mov_b32 s[c5_EV], 6
  s_mov_b32 s[c5_LIMIT], 0                         ; (the step's end looks at EV only when HAVE < LIMIT fails; .Lc5_limit recomputes it)
.else
  s_mov_b32 s[c5_EV], 8                            ; (mode 1: the caller takes over once the symbol's model update is stored)
  s_mov_b32 s[c5_LIMIT], 0
  s_mov_b32 s[c5_NCTX], s[c5_CTX]
  s_mov_b32 s[c5_LIT], 0
.endif
  s_mov_b64 s[c5_LB:c5_LB+1], s[c5_ARENA:c5_ARENA+1]
  s_mov_b32 s[c5_LOFF], c5_OFF_SCR+512
  s_branch .Lc5_update_\sp\()_\u
.Lc5_late_hit_\sp\()_\u:                           ; the predicted byte behind an escape byte
  v_mov_b32 v[c5_VLOWU], v[c5_VTB]
  v_mov_b32 v[c5_VFRQ], v[c5_VFHIT]
.Lc5_late_\sp\()_\u:                               ; the symbol after an escape byte: 0 = the byte itself, else a match length
.if \sp
  c5_pick_sp \u
.else
  c5_pick \u
.endif
  c5_consume c5_VUNIT
  s_cbranch_vccz .Lc5_late_go_\sp\()_\u
  c5_refill
.Lc5_late_go_\sp\()_\u:
  s_mov_b32 s[c5_LRIDX], -1
.Lc5_tok_after_\sp\()_\u:
  s_mov_b32 s[c5_EV], s[c5_AESC]                   ; 1: the match token's table work is in flight, 7: it is not
  s_mov_b32 s[c5_AESC], 0
  s_cmp_eq_u32 s[c5_SYM], 0
  s_cbranch_scc0 .Lc5_tok_match_\sp\()_\u
  s_mov_b32 s[c5_LIT], \esc
  s_mov_b32 s[c5_EV], 0
  c5_literal
  s_lshl_b32 s[c5_NCTX], s[c5_CTX], 8
  s_or_b32 s[c5_NCTX], s[c5_NCTX], s[c5_LIT]
  c5_issue c5_NCTX
  s_branch .Lc5_update_\sp\()_\u
.Lc5_tok_match_\sp\()_\u:                          ; a match length: finish this symbol's model update first
  s_mov_b32 s[c5_LIMIT], 0
  s_mov_b32 s[c5_NCTX], s[c5_CTX]
  s_mov_b32 s[c5_LIT], 0
  s_mov_b64 s[c5_LB:c5_LB+1], s[c5_ARENA:c5_ARENA+1]
  s_mov_b32 s[c5_LOFF], c5_OFF_SCR+512
  s_branch .Lc5_update_\sp\()_\u
.Lc5_upd_single_\sp\()_\u:                         ; PPMX singleton rule, cr-ppm.c:136-138: count(257) - 1
  s_lshr_b32 s[c5_T0], s[c5_SX], 8
  s_sub_u32 s[c5_T0], s[c5_T0], 1
  s_and_b32 s[c5_T0], s[c5_T0], 0xff
  s_and_b32 s[c5_SX], s[c5_SX], 0xff
  s_lshl_b32 s[c5_T1], s[c5_T0], 8
  s_or_b32 s[c5_SX], s[c5_SX], s[c5_T1]
  s_cmp_gt_u32 s[c5_T0], 250
  s_cbranch_scc1 .Lc5_upd_halve_\sp\()_\u
.Lc5_upd_flag_\sp\()_\u:                           ; a byte of the node and a changed flag word
  c5_o3_miss
.if \sp
  v_writelane_b32 v[c5_PP], s[c5_SX], 62
  s_bitset1_b32 s[c5_MW+1], 30
  c5_st_pairs
  c5_st_o3_lit
  c5_tail 4, \u
.else
  c5_st_node
  c5_st_flag
  c5_st_o3_lit
  c5_tail 5, \u
.endif
.Lc5_upd_halve_\sp\()_\u:
.if \sp
  c5_halve_sp
.else
  c5_halve
.endif
  s_mov_b64 s[c5_MW:c5_MW+1], -1
  s_branch .Lc5_upd_flag_\sp\()_\u
.if \sp
.Lc5_convert_\u:
  ; the line is full and the byte is new: the node moves into the next free dense slot (its counts scattered through LDS into
  ; the 256-byte layout); the line keeps the flag word and gets the slot's number and the mark. This step's own update is the
  ; dense variant's.
  v_and_b32 v[c5_VT2], v[c5_VMCNT], v[c5_PP]
  v_add_u32 v[c5_VT3], v[c5_VLDB], v[c5_VSYM]
  v_cmp_ne_u32 vcc, 0, v[c5_VT2]
  v_readfirstlane_b32 s[c5_T0], v[c5_VDSLOT]
  s_nop 0
  s_mov_b64 exec, vcc
  ds_write_b8 v[c5_VT3], v[c5_VT2]
  s_mov_b64 exec, -1
  ds_read_b32 v[c5_W], v[c5_VLDZ]
  ds_write_b32 v[c5_VLDZ], v[c5_VZERO]             ; (the scratch is all zero between steps)
  v_add_u32 v[c5_VDSLOT], 1, v[c5_VDSLOT]
  s_cmp_ge_u32 s[c5_T0], \ds                       ; a slot behind the dense area: cannot happen within 2n coding steps (a damaged stream is
  s_cbranch_scc1 .Lc5_fail_\u                      ; held to them too), checked all the same before anything is stored there
  s_lshl_b32 s[c5_T1], s[c5_T0], 8
  v_add_u32 v[c5_VDA], s[c5_T1], v[c5_VDOFF4]
.if c5_dlds
  s_min_u32 s[c5_T1], s[c5_T0], c5_dlds
  s_lshl_b32 s[c5_T1], s[c5_T1], 8
  v_add_u32 v[c5_VDL], s[c5_T1], v[c5_VLDZ]
.endif
  s_and_b32 s[c5_T2], s[c5_T0], 0xffff
  s_lshr_b32 s[c5_T3], s[c5_T0], 16
  s_mov_b32 s[c5_T1], 0xffff
  v_writelane_b32 v[c5_PP], s[c5_T2], 0
  v_writelane_b32 v[c5_PP], s[c5_T3], 1
  v_writelane_b32 v[c5_PP], s[c5_T1], 61
  v_writelane_b32 v[c5_PP], s[c5_SX], 62
  v_add_u32 v[c5_SA], s[c5_NO], v[c5_VLANE2]
  global_store_short v[c5_SA], v[c5_PP], s[c5_BN:c5_BN+1]
  s_waitcnt lgkmcnt(0)
  global_store_dword v[c5_VDA], v[c5_W], s[c5_ARENA:c5_ARENA+1]
.if c5_dlds
  ds_write_b32 v[c5_VDL], v[c5_W] offset:0x110
.endif
  s_or_b32 s[c5_NDNO], s[c5_NO], 1
  s_branch .Lc5_upd_esc_0_\u
.endif
.endm
.endif
)ASM"

#define CR_V5_ASM_BODY R"ASM(
  s_mov_b64 s[c5_ARENA:c5_ARENA+1], %[arena]
  s_add_u32 s[c5_BN], s[c5_ARENA], c5_OFF_NODES
  s_addc_u32 s[c5_BN+1], s[c5_ARENA+1], 0
  s_add_u32 s[c5_B3], s[c5_ARENA], c5_OFF_O3D
  s_addc_u32 s[c5_B3+1], s[c5_ARENA+1], 0
  s_add_u32 s[c5_B1], s[c5_ARENA], c5_OFF_O1
  s_addc_u32 s[c5_B1+1], s[c5_ARENA+1], 0
  s_mov_b64 s[c5_DST:c5_DST+1], %[dst]
  s_mov_b32 s[c5_CTX], %[ctx]
  v_mov_b32 v[c5_VRANGE], %[range]
  v_mov_b32 v[c5_VCACHE], %[cache]
  v_mov_b32 v[c5_VIBLO], %[iblo]
  v_mov_b32 v[c5_VIBHI], %[ibhi]
  v_mov_b32 v[c5_VIBITS], %[ibits]
  s_mov_b32 s[c5_WIDX], %[widx]
  s_mov_b32 s[c5_HAVE], %[have]
  c5_prio %=
  s_mov_b32 s[c5_LEARNED], %[learned]
  s_cmp_lg_u32 %[aesc], 0
  s_cselect_b32 s[c5_AESC], 7, 0
  s_mov_b32 s[c5_X8LO], %[x8lo]
  s_mov_b32 s[c5_X8HI], %[x8hi]
  s_mov_b32 s[c5_TOTAL], %[total]
  s_mov_b32 s[c5_GEN], %[gen]
  s_lshl_b32 s[c5_G3S], %[g3], 4
.if c5_mode != 1
  s_add_u32 s[c5_T0], s[c5_LEARNED], 64            ; steps run while have < LIMIT = min(learned + 64, total)
  s_min_u32 s[c5_LIMIT], s[c5_T0], s[c5_TOTAL]
.else
  s_mov_b32 s[c5_LIMIT], s[c5_TOTAL]
.endif
  s_cmp_ge_u32 s[c5_WIDX], 62                      ; (an event other than the window's may have ended the last call with the window low)
  s_cselect_b32 s[c5_LIMIT], 0, s[c5_LIMIT]
  v_mov_b32 v[c5_PENDLO], %[plo]
  v_mov_b32 v[c5_PENDHI], %[phi]
  v_mov_b32 v[c5_WIN], %[win]
  s_mov_b32 s[c5_LUTM], 0x33322100
  s_mov_b32 s[c5_LUTM+1], 0x44444443
  s_mov_b32 s[c5_LUTH], 0x87654321
  s_mov_b32 s[c5_LUTH+1], 0xffedcba9
  s_mov_b32 s[c5_NDNO], -1
  s_mov_b32 s[c5_O3LK], -1
  s_mov_b32 s[c5_LRIDX], -1
  s_mov_b32 s[c5_EV], 0
  v_mov_b32 v[c5_PACC], 0
  v_mov_b32 v[c5_PCNT], 0
  v_mbcnt_lo_u32_b32 v[c5_LANE], -1, 0
  v_mbcnt_hi_u32_b32 v[c5_LANE], -1, v[c5_LANE]
  v_lshlrev_b32 v[c5_VLANE4], 2, v[c5_LANE]
  v_lshlrev_b32 v[c5_VLANE2], 1, v[c5_LANE]
  v_mov_b32 v[c5_VZERO], 0
  v_mov_b32 v[c5_VONE], 1
  v_mov_b32 v[c5_VT0], 0xff
  v_cmp_gt_u32 vcc, 62, v[c5_LANE]
  v_cndmask_b32 v[c5_VMCNT], 0, v[c5_VT0], vcc
  v_mov_b32 v[c5_VT1], 0x100
  s_nop 0
  v_cndmask_b32 v[c5_VHI], v[c5_VT1], v[c5_VZERO], vcc
  ; three words the C++ side keeps in the arena's scratch line (the statement is short of operand registers): where the dense
  ; slots start, the LDS address of the wave's 256 scratch bytes, the next free dense slot
  v_mov_b32 v[c5_VT0], c5_OFF_SCR+896
  global_load_dwordx3 v[c5_VT2:c5_VT2+2], v[c5_VT0], s[c5_ARENA:c5_ARENA+1]
  c5_issue c5_CTX                                  ; (the context's model goes out beside them: one round trip per entry, not two)
  s_waitcnt vmcnt(0)
  v_add_u32 v[c5_VDOFF4], v[c5_VT2], v[c5_VLANE4]
.if c5_hw
  v_add_u32 v[c5_HWB], 0x110, v[c5_VT3]            ; the mailbox behind the wave's scratch bytes
  s_mov_b32 s[c5_HWPOST], 0
  v_lshl_add_u32 v[c5_HWA2], v[c5_LANE], 1, v[c5_HWB]
  v_lshl_add_u32 v[c5_HWA4], v[c5_LANE], 2, v[c5_HWB]
  v_lshl_add_u32 v[c5_HWA16], v[c5_LANE], 4, v[c5_HWB]
  ds_read_b32 v[c5_VT0], v[c5_HWB] offset:12
  s_waitcnt lgkmcnt(0)
  v_readfirstlane_b32 s[c5_HWSEQ], v[c5_VT0]
.endif
  v_mov_b32 v[c5_VLDB], v[c5_VT3]
  v_add_u32 v[c5_VLDX], 0x100, v[c5_VT3]
  v_add_u32 v[c5_VLDZ], v[c5_VT3], v[c5_VLANE4]
  v_mov_b32 v[c5_VDSLOT], v[c5_VZERO]
  v_mov_b32 v[c5_VZERO], 0
.if c5_alearn
  v_mov_b32 v[c5_FLV], 0
.endif
  s_nop 0
  ds_write_b32 v[c5_VLDZ], v[c5_VZERO]             ; the wave's LDS scratch: all zero between steps
  s_branch .Lc5_after_event_%=                     ; (64 positions may be waiting to be learned right now)

.if c5_align & 1
  .p2align 6
.endif
.Lc5_head_%=:
  ; ---------------------------------------------------------------- this step's model: the node's line has arrived
  c5_prof_begin 8, c5_HWSEQ                        ; (8: head to the next context's loads, any step; 10 / 11 / 12: a byte of the node /
  c5_prof_begin 10, c5_HWSEQ                       ; the predicted byte / an escape; 9: from those loads to the wait at the step's end)
  c5_prof_begin 11, c5_HWSEQ
  c5_prof_begin 12, c5_HWSEQ
  c5_prof_begin 14, c5_HWSEQ                       ; (14: visits = first-use nodes)
  s_mov_b32 s[c5_NO], s[c5_NON]                    ; the line offset the loads of this context were issued with
  s_xor_b32 s[c5_T0], s[c5_NO], s[c5_NDNO]
  s_cmp_lt_u32 s[c5_T0], 2
  s_cbranch_scc1 .Lc5_same_%=                      ; the context came straight back: PP / W and SX are newer than memory
  v_readlane_b32 s[c5_T0], v[c5_NW], 63            ; the flag word's halves: generation | count(257) << 8 | count(256)
  v_readlane_b32 s[c5_SX], v[c5_NW], 62
  v_readlane_b32 s[c5_T1], v[c5_NW], 61
  v_mov_b32 v[c5_PP], v[c5_NW]
  s_cmp_lg_u32 s[c5_T0], s[c5_GEN]
  s_cbranch_scc1 .Lc5_fresh_%=                     ; stale tag: first use in this block (o2_model_init)
  s_cmp_eq_u32 s[c5_T1], 0xffff
  s_cbranch_scc1 .Lc5_load_dense_%=                ; the node has outgrown its line
  c5_step 1, %=, %[esc], %[dslots]
  c5_step 0, %=, %[esc], %[dslots]
.Lc5_same_%=:
  s_bitcmp1_b32 s[c5_NDNO], 0
  s_cbranch_scc1 .Lc5_node_ok_0_%=
  s_branch .Lc5_node_ok_1_%=
.Lc5_fresh_%=:                                     ; o2_model_init (cr-o2model.c:38-44), written out at once: the step's own stores
  c5_prof_end 14, c5_HWSEQ
.if c5_hw
  s_mov_b32 s[c5_HWPOST], 0
.endif
  v_mov_b32 v[c5_PP], 0xff00                       ; then only carry what it changes. Every pair unused, counts (256, 257) = (1, 1)
  s_mov_b32 s[c5_SX], 0x101
  s_nop 0
  v_writelane_b32 v[c5_PP], s[c5_SX], 62
  v_writelane_b32 v[c5_PP], s[c5_GEN], 63
  v_add_u32 v[c5_SA], s[c5_NO], v[c5_VLANE2]
  global_store_short v[c5_SA], v[c5_PP], s[c5_BN:c5_BN+1]
  ; an empty node (every sixth step of a bench block) holds no byte: the total is count(256) + count(257) = 2, so unit = range / 2,
  ; the prediction hit lies below unit and the escape above — no scan, no division. The step continues where a node's
  ; variant finds its symbol outside the node.
  s_mov_b32 s[c5_K3], s[c5_K3N]                    ; (the order-3 entry, as at the head of a variant)
  v_readfirstlane_b32 s[c5_T0], v[c5_FE]
  s_cmp_eq_u32 s[c5_K3], s[c5_O3LK]
  s_cselect_b32 s[c5_T0], s[c5_O3LV], s[c5_T0]
  s_and_b32 s[c5_T1], s[c5_T0], 0xf0
  s_cmp_eq_u32 s[c5_T1], s[c5_G3S]
  s_cselect_b32 s[c5_T0], s[c5_T0], 0
  s_lshr_b32 s[c5_PRED], s[c5_T0], 8
  s_and_b32 s[c5_CONF], s[c5_T0], 15
  s_mov_b32 s[c5_FHIT], 1
  s_mov_b32 s[c5_FESC], 1
  v_mov_b32 v[c5_VFHIT], 1
  v_mov_b32 v[c5_VFESC], 1
  v_mov_b32 v[c5_CX], 0
  v_or_b32 v[c5_VSYM], 0xff, v[c5_VHI]
  v_lshrrev_b32 v[c5_VUNIT], 1, v[c5_VRANGE]
  v_mov_b32 v[c5_VTB], 0
  s_mov_b32 s[c5_BYTES], 0
.if c5_mode != 1
  s_sub_u32 s[c5_T0], s[c5_HAVE], s[c5_LEARNED]    ; the position about to be decoded becomes pending, as in a variant's scan
  s_lshl_b64 exec, 1, s[c5_T0]
  v_mov_b32 v[c5_PENDLO], s[c5_X8LO]
  v_mov_b32 v[c5_PENDHI], s[c5_X8HI]
  s_mov_b64 exec, -1
.endif
.if c5_pf & 4
  s_and_b32 s[c5_T2], s[c5_CTX], 0xff
  s_lshl_b32 s[c5_T2], s[c5_T2], 15
  s_lshl_b32 s[c5_T3], s[c5_PRED], 7
  s_add_u32 s[c5_T3], s[c5_T3], s[c5_T2]
  v_mov_b32 v[c5_PFA], s[c5_T3]
  s_mov_b64 exec, 1
  global_load_ubyte v[c5_PFD], v[c5_PFA], s[c5_BN:c5_BN+1]
  s_mov_b64 exec, -1
.endif
  s_mov_b32 s[c5_NDNO], s[c5_NO]
  s_mov_b32 s[c5_HALV], 0
.if c5_hw
  s_branch .Lc5_not_in_node_1_%=
.else
  v_mov_b32 v[c5_PRES], 0                          ; (the node holds no byte: no presence scatter, and (0 + count(256)) x unit = unit)
  v_mov_b32 v[c5_VLOWU], v[c5_VUNIT]
  s_branch .Lc5_nin_test_1_%=
.endif
.Lc5_load_dense_%=:                                ; pairs 0 / 1 = the slot number: its 256 count bytes, one dword per lane
  v_readlane_b32 s[c5_T0], v[c5_NW], 0
  v_readlane_b32 s[c5_T1], v[c5_NW], 1
  s_lshl_b32 s[c5_T1], s[c5_T1], 16
  s_or_b32 s[c5_T0], s[c5_T0], s[c5_T1]
.if c5_dlds
  s_min_u32 s[c5_T1], s[c5_T0], c5_dlds
  s_lshl_b32 s[c5_T1], s[c5_T1], 8
  v_add_u32 v[c5_VDL], s[c5_T1], v[c5_VLDZ]
.endif
  s_lshl_b32 s[c5_T0], s[c5_T0], 8
  v_add_u32 v[c5_VDA], s[c5_T0], v[c5_VDOFF4]
  c5_prof_begin 13, c5_HWSEQ                       ; (13: a dense node's second round trip)
.if c5_dlds
  s_cmp_lt_u32 s[c5_T0], c5_dlds * 256
  s_cbranch_scc0 .Lc5_load_dense_far_%=
  ds_read_b32 v[c5_W], v[c5_VDL] offset:0x110
  s_waitcnt lgkmcnt(0)
  c5_prof_end 13, c5_HWSEQ
  s_branch .Lc5_node_ok_0_%=
.Lc5_load_dense_far_%=:
.endif
  global_load_dword v[c5_W], v[c5_VDA], s[c5_ARENA:c5_ARENA+1]
  s_waitcnt vmcnt(0)
  c5_prof_end 13, c5_HWSEQ
  s_branch .Lc5_node_ok_0_%=
.Lc5_slow_tail_%=:
  s_cmp_lg_u32 s[c5_EV], 0
  s_cbranch_scc1 .Lc5_event_%=
  s_branch .Lc5_limit_%=
.Lc5_after_event_%=:
  s_cmp_lt_u32 s[c5_HAVE], s[c5_LIMIT]
  s_cbranch_scc1 .Lc5_head_%=
.Lc5_limit_%=:
  ; rare from here: the window is running low, 64 positions are waiting to be learned, or the block is complete
.if c5_alearn
  ; ... or a learn event is in flight. Its last round's results are in (a step's end has waited for everything older than its
  ; own loads, an event has finished it): the lanes that met another key try their next slot, T5 = is any lane left
  v_readfirstlane_b32 s[c5_T5], v[c5_FLV]
  s_cmp_eq_u32 s[c5_T5], 0
  s_cbranch_scc1 .Lc5_limit_go_%=
  c5_prof_begin 17, c5_HWSEQ                       ; (17: a round of a learn event in flight)
  c5_learn_masks %[lzsh]
  c5_lzp_round c5_R8, c5_A8, c5_D8, c5_VH8, %[off8], c5_PM8, %=
  c5_lzp_round c5_R4, c5_A4, c5_D4, c5_VH4, %[off4], c5_PM4, %=
  v_cndmask_b32_e64 v[c5_PMV8], 0, 1, s[c5_PM8:c5_PM8+1]
  v_cndmask_b32_e64 v[c5_PMV4], 0, 1, s[c5_PM4:c5_PM4+1]
  s_or_b64 s[c5_T0:c5_T0+1], s[c5_PM8:c5_PM8+1], s[c5_PM4:c5_PM4+1]
  s_cmp_lg_u64 s[c5_T0:c5_T0+1], 0
  s_cselect_b32 s[c5_T5], 1, 0
  v_mov_b32 v[c5_FLV], s[c5_T5]
  c5_prof_end 17, c5_HWSEQ
.Lc5_limit_go_%=:
.endif
  s_cmp_ge_u32 s[c5_WIDX], 62
  s_cbranch_scc1 .Lc5_exit_window_%=
.if c5_mode != 1
  s_sub_u32 s[c5_T0], s[c5_HAVE], s[c5_LEARNED]
  s_cmp_ge_u32 s[c5_T0], 64
.if c5_mode == 0
  s_cbranch_scc1 .Lc5_learn_%=
.else
  s_cbranch_scc1 .Lc5_exit_learn_%=
.endif
  s_add_u32 s[c5_T0], s[c5_LEARNED], 64            ; (mode 2 moves `learned` during the first 16 positions)
.if c5_alearn
  s_min_u32 s[c5_T0], s[c5_T0], s[c5_TOTAL]
  s_cmp_lg_u32 s[c5_T5], 0
  s_cselect_b32 s[c5_LIMIT], 0, s[c5_T0]           ; (a learn event in flight: the next step's end comes here again)
  s_cmp_lt_u32 s[c5_HAVE], s[c5_T0]
.else
  s_min_u32 s[c5_LIMIT], s[c5_T0], s[c5_TOTAL]
  s_cmp_lt_u32 s[c5_HAVE], s[c5_LIMIT]
.endif
  s_cbranch_scc1 .Lc5_head_%=
.endif
  s_mov_b32 s[c5_EV], 4
  s_branch .Lc5_exit_%=

  ; ================================================================ match token, cr-coder.c:270-283
  ; The length symbol's step is complete (its stores are out). Short matches (< 64 bytes, source not overlapping
  ; the destination) are done here; everything else leaves through the event exit to the C++ around the statement.
.Lc5_event_%=:
  c5_learn_drain %[off8], %[off4], %[lzsh], %=
  s_cmp_eq_u32 s[c5_EV], 6
  s_cbranch_scc1 .Lc5_m_issue_%=                   ; the escape byte: a match token is about to follow, start its table work
  s_cmp_eq_u32 s[c5_EV], 1
  s_cbranch_scc1 .Lc5_m_checks_%=
  s_cmp_eq_u32 s[c5_EV], 7
  s_cbranch_scc0 .Lc5_exit_%=
.Lc5_m_checks_%=:
  c5_prof_begin 2
  s_add_u32 s[c5_T0], s[c5_HAVE], s[c5_SYM]
  s_cmp_gt_u32 s[c5_T0], s[c5_TOTAL]
  s_cbranch_scc1 .Lc5_m_slow_%=                    ; damaged stream: reported by the C++ side
  s_cmp_gt_u32 s[c5_T0], %[cap]
  s_cbranch_scc1 .Lc5_m_slow_%=
  s_cmp_ge_u32 s[c5_SYM], 64
  s_cbranch_scc1 .Lc5_m_slow_%=
  s_cmp_eq_u32 s[c5_EV], 7
  s_cbranch_scc1 .Lc5_m_issue_%=                   ; (the statement was left and re-entered since the escape byte)
  c5_prof_begin 3
  s_waitcnt vmcnt(3)                               ; everything but the length symbol's last three stores
  c5_prof_end 3
  s_branch .Lc5_m_back_%=
  ; ---- cr_lzp_learn_predict (crgpu_lzp.h), first half: matcher_update for the pending positions learned .. have-1
  ; (lane j holds the 8 bytes in front of learned + j) and matcher_getpos for `have` (X8) go out together. Issued
  ; at the escape byte, so the round trip runs under the length symbol's coding step.
.Lc5_m_issue_%=:
  s_sub_u32 s[c5_T0], s[c5_HAVE], s[c5_LEARNED]
  s_lshl_b64 s[c5_ACT:c5_ACT+1], 1, s[c5_T0]
  s_sub_u32 s[c5_ACT], s[c5_ACT], 1
  s_subb_u32 s[c5_ACT+1], s[c5_ACT+1], 0
  s_lshr_b32 s[c5_LZM], -1, %[lzsh]
  v_add_u32 v[c5_VQ], s[c5_LEARNED], v[c5_LANE]
  v_alignbit_b32 v[c5_VT0], v[c5_PENDHI], v[c5_PENDLO], 20     ; cr_key8: x ^ x >> 20 ^ x >> 40, 24 bits
  v_lshrrev_b32 v[c5_VT1], 8, v[c5_PENDHI]
  v_xor_b32 v[c5_VK8], v[c5_PENDLO], v[c5_VT0]
  v_xor_b32 v[c5_VK8], v[c5_VK8], v[c5_VT1]
  v_and_b32 v[c5_VK8], 0xffffff, v[c5_VK8]
  v_lshrrev_b32 v[c5_VT0], 6, v[c5_PENDHI]                     ; cr_key4: y ^ y >> 6 ^ y >> 12, 20 bits
  v_lshrrev_b32 v[c5_VT1], 12, v[c5_PENDHI]
  v_xor_b32 v[c5_VK4], v[c5_PENDHI], v[c5_VT0]
  v_xor_b32 v[c5_VK4], v[c5_VK4], v[c5_VT1]
  v_and_b32 v[c5_VK4], 0xfffff, v[c5_VK4]
  v_lshrrev_b32 v[c5_VK2], 16, v[c5_PENDHI]                    ; cr_key2
  s_mov_b32 s[c5_T0], 0x9e3779b1
  v_mul_lo_u32 v[c5_VH8], v[c5_VK8], s[c5_T0]
  v_mul_lo_u32 v[c5_VH4], v[c5_VK4], s[c5_T0]
  v_lshrrev_b32 v[c5_VH8], %[lzsh], v[c5_VH8]
  v_lshrrev_b32 v[c5_VH4], %[lzsh], v[c5_VH4]
  v_lshlrev_b32 v[c5_A8], 3, v[c5_VH8]
  v_lshlrev_b32 v[c5_A4], 3, v[c5_VH4]
  v_lshlrev_b32 v[c5_A2], 2, v[c5_VK2]
  v_add_u32 v[c5_A8], %[off8], v[c5_A8]
  v_add_u32 v[c5_A4], %[off4], v[c5_A4]
  v_add_u32 v[c5_A2], %[off2], v[c5_A2]
  v_mov_b32 v[c5_D8], v[c5_VQ]                                 ; {position, key + 1} to swap in, 0 to compare with
  v_add_u32 v[c5_D8+1], 1, v[c5_VK8]
  v_mov_b32 v[c5_D8+2], 0
  v_mov_b32 v[c5_D8+3], 0
  v_mov_b32 v[c5_D4], v[c5_VQ]
  v_add_u32 v[c5_D4+1], 1, v[c5_VK4]
  v_mov_b32 v[c5_D4+2], 0
  v_mov_b32 v[c5_D4+3], 0
  s_mov_b64 exec, s[c5_ACT:c5_ACT+1]
  s_cbranch_execnz .Lc5_m_atomics_%=
  s_mov_b64 exec, 1                                ; nothing pending: lane 0 aims at a scratch line, so that six operations
  v_mov_b32 v[c5_A8], c5_OFF_SCR+768               ; are in flight whatever the case (the waits of the next step count them)
  v_mov_b32 v[c5_A4], c5_OFF_SCR+776
  v_mov_b32 v[c5_A2], c5_OFF_SCR+784
.Lc5_m_atomics_%=:
  global_atomic_cmpswap_x2 v[c5_R8:c5_R8+1], v[c5_A8], v[c5_D8:c5_D8+3], s[c5_ARENA:c5_ARENA+1] sc0
  global_atomic_cmpswap_x2 v[c5_R4:c5_R4+1], v[c5_A4], v[c5_D4:c5_D4+3], s[c5_ARENA:c5_ARENA+1] sc0
  global_atomic_umax v[c5_A2], v[c5_VQ], s[c5_ARENA:c5_ARENA+1]
  s_mov_b64 exec, -1
  s_lshr_b64 s[c5_T0:c5_T0+1], s[c5_X8LO:c5_X8LO+1], 20        ; the same three keys of X8
  s_lshr_b32 s[c5_T2], s[c5_X8HI], 8
  s_xor_b32 s[c5_MK8], s[c5_X8LO], s[c5_T0]
  s_xor_b32 s[c5_MK8], s[c5_MK8], s[c5_T2]
  s_and_b32 s[c5_MK8], s[c5_MK8], 0xffffff
  s_lshr_b32 s[c5_T0], s[c5_X8HI], 6
  s_lshr_b32 s[c5_T1], s[c5_X8HI], 12
  s_xor_b32 s[c5_MK4], s[c5_X8HI], s[c5_T0]
  s_xor_b32 s[c5_MK4], s[c5_MK4], s[c5_T1]
  s_and_b32 s[c5_MK4], s[c5_MK4], 0xfffff
  s_lshr_b32 s[c5_MK2], s[c5_X8HI], 16
  s_mul_i32 s[c5_MH8], s[c5_MK8], 0x9e3779b1
  s_mul_i32 s[c5_MH4], s[c5_MK4], 0x9e3779b1
  s_lshr_b32 s[c5_MH8], s[c5_MH8], %[lzsh]
  s_lshr_b32 s[c5_MH4], s[c5_MH4], %[lzsh]
  s_lshl_b32 s[c5_T0], s[c5_MH8], 3
  s_lshl_b32 s[c5_T1], s[c5_MH4], 3
  s_lshl_b32 s[c5_T2], s[c5_MK2], 2
  s_add_u32 s[c5_T0], s[c5_T0], %[off8]
  s_add_u32 s[c5_T1], s[c5_T1], %[off4]
  s_add_u32 s[c5_T2], s[c5_T2], %[off2]
  v_mov_b32 v[c5_LA8], s[c5_T0]
  v_mov_b32 v[c5_LA4], s[c5_T1]
  v_mov_b32 v[c5_LA2], s[c5_T2]
  global_load_dwordx2 v[c5_E8:c5_E8+1], v[c5_LA8], s[c5_ARENA:c5_ARENA+1] sc1
  global_load_dwordx2 v[c5_E4:c5_E4+1], v[c5_LA4], s[c5_ARENA:c5_ARENA+1] sc1
  global_load_dword v[c5_E2], v[c5_LA2], s[c5_ARENA:c5_ARENA+1] sc1
  s_cmp_eq_u32 s[c5_EV], 6
  s_cbranch_scc1 .Lc5_m_issued_%=
  s_waitcnt vmcnt(0)
  ; ---- second half, at the match token: the scalars of the first half again (the coding step in between used the registers)
.Lc5_m_back_%=:
  s_sub_u32 s[c5_T0], s[c5_HAVE], s[c5_LEARNED]
  s_lshl_b64 s[c5_ACT:c5_ACT+1], 1, s[c5_T0]
  s_sub_u32 s[c5_ACT], s[c5_ACT], 1
  s_subb_u32 s[c5_ACT+1], s[c5_ACT+1], 0
  s_lshr_b32 s[c5_LZM], -1, %[lzsh]
  s_lshr_b64 s[c5_T0:c5_T0+1], s[c5_X8LO:c5_X8LO+1], 20
  s_lshr_b32 s[c5_T2], s[c5_X8HI], 8
  s_xor_b32 s[c5_MK8], s[c5_X8LO], s[c5_T0]
  s_xor_b32 s[c5_MK8], s[c5_MK8], s[c5_T2]
  s_and_b32 s[c5_MK8], s[c5_MK8], 0xffffff
  s_lshr_b32 s[c5_T0], s[c5_X8HI], 6
  s_lshr_b32 s[c5_T1], s[c5_X8HI], 12
  s_xor_b32 s[c5_MK4], s[c5_X8HI], s[c5_T0]
  s_xor_b32 s[c5_MK4], s[c5_MK4], s[c5_T1]
  s_and_b32 s[c5_MK4], s[c5_MK4], 0xfffff
  s_lshr_b32 s[c5_MK2], s[c5_X8HI], 16
  ; the inserts whose home slot held another key walk on; their rounds run under the two waits below
  s_mov_b64 s[c5_PM8:c5_PM8+1], s[c5_ACT:c5_ACT+1]
  s_mov_b64 s[c5_PM4:c5_PM4+1], s[c5_ACT:c5_ACT+1]
  c5_lzp_round c5_R8, c5_A8, c5_D8, c5_VH8, %[off8], c5_PM8, %=
  c5_lzp_round c5_R4, c5_A4, c5_D4, c5_VH4, %[off4], c5_PM4, %=
  c5_prof_begin 7
  ; ---- the three candidates: a pending position with the same key is the latest by construction, else the table's
  v_readfirstlane_b32 s[c5_E8K], v[c5_E8+1]
  v_readfirstlane_b32 s[c5_E8P], v[c5_E8]
  v_readfirstlane_b32 s[c5_E4K], v[c5_E4+1]
  v_readfirstlane_b32 s[c5_E4P], v[c5_E4]
  v_readfirstlane_b32 s[c5_C2], v[c5_E2]
  s_add_u32 s[c5_T2], s[c5_MK8], 1
  s_cmp_eq_u32 s[c5_E8K], s[c5_T2]
  s_cselect_b32 s[c5_C8], s[c5_E8P], 8
  s_cselect_b32 s[c5_T3], 0, s[c5_E8K]
  s_cmp_lg_u32 s[c5_T3], 0
  s_cbranch_scc1 .Lc5_m_probe8_%=
.Lc5_m_got8_%=:
  s_add_u32 s[c5_T2], s[c5_MK4], 1
  s_cmp_eq_u32 s[c5_E4K], s[c5_T2]
  s_cselect_b32 s[c5_C4], s[c5_E4P], 4
  s_cselect_b32 s[c5_T3], 0, s[c5_E4K]
  s_cmp_lg_u32 s[c5_T3], 0
  s_cbranch_scc1 .Lc5_m_probe4_%=
.Lc5_m_got4_%=:
  v_cmp_eq_u32 vcc, s[c5_MK8], v[c5_VK8]
  s_and_b64 s[c5_T0:c5_T0+1], vcc, s[c5_ACT:c5_ACT+1]
  v_cmp_eq_u32 vcc, s[c5_MK4], v[c5_VK4]
  s_and_b64 s[c5_T2:c5_T2+1], vcc, s[c5_ACT:c5_ACT+1]
  v_cmp_eq_u32 vcc, s[c5_MK2], v[c5_VK2]
  s_and_b64 s[c5_T4:c5_T4+1], vcc, s[c5_ACT:c5_ACT+1]
  s_or_b64 s[c5_T6:c5_T6+1], s[c5_T0:c5_T0+1], s[c5_T2:c5_T2+1]
  s_or_b64 s[c5_T6:c5_T6+1], s[c5_T6:c5_T6+1], s[c5_T4:c5_T4+1]
  s_cmp_lg_u64 s[c5_T6:c5_T6+1], 0
  s_cbranch_scc1 .Lc5_m_batch_%=
.Lc5_m_cand_%=:
  c5_prof_end 7
  ; ---- matcher_getpos' context checks (cr-matcher.c:59-73) and the source bytes of all three candidates, one round trip
  s_sub_u32 s[c5_T0], s[c5_HAVE], s[c5_C8]
  s_sub_u32 s[c5_T1], s[c5_HAVE], s[c5_C4]
  s_sub_u32 s[c5_T2], s[c5_HAVE], s[c5_C2]
  s_min_u32 s[c5_T0], s[c5_T0], s[c5_T1]
  s_min_u32 s[c5_T0], s[c5_T0], s[c5_T2]
  s_cmp_gt_u32 s[c5_SYM], s[c5_T0]
  s_cbranch_scc1 .Lc5_m_overlap_%=                 ; a source that runs into the destination repeats: C++ side
  s_sub_u32 s[c5_T0], s[c5_C8], 8
  s_sub_u32 s[c5_T1], s[c5_C4], 4
  v_mov_b32 v[c5_LA8], s[c5_T0]
  v_mov_b32 v[c5_LA4], s[c5_T1]
  global_load_dwordx2 v[c5_V8:c5_V8+1], v[c5_LA8], s[c5_DST:c5_DST+1]
  global_load_dword v[c5_V4], v[c5_LA4], s[c5_DST:c5_DST+1]
  s_lshl_b64 s[c5_LM:c5_LM+1], 1, s[c5_SYM]
  s_sub_u32 s[c5_LM], s[c5_LM], 1
  s_subb_u32 s[c5_LM+1], s[c5_LM+1], 0
  v_add_u32 v[c5_B8], s[c5_C8], v[c5_LANE]
  v_add_u32 v[c5_B4], s[c5_C4], v[c5_LANE]
  v_add_u32 v[c5_B2], s[c5_C2], v[c5_LANE]
  s_mov_b64 exec, s[c5_LM:c5_LM+1]
  global_load_ubyte v[c5_S8], v[c5_B8], s[c5_DST:c5_DST+1]
  global_load_ubyte v[c5_S4], v[c5_B4], s[c5_DST:c5_DST+1]
  global_load_ubyte v[c5_S2], v[c5_B2], s[c5_DST:c5_DST+1]
  s_mov_b64 exec, -1
  v_add_u32 v[c5_B8], s[c5_HAVE], v[c5_LANE]
  c5_prof_begin 4
  s_waitcnt vmcnt(0)
  c5_prof_end 4
  v_readfirstlane_b32 s[c5_T0], v[c5_V8]
  v_readfirstlane_b32 s[c5_T1], v[c5_V8+1]
  v_readfirstlane_b32 s[c5_T2], v[c5_V4]
  s_cmp_eq_u32 s[c5_T2], s[c5_X8HI]
  s_cselect_b64 s[c5_T4:c5_T4+1], -1, 0
  s_cmp_eq_u64 s[c5_T0:c5_T0+1], s[c5_X8LO:c5_X8LO+1]
  s_cselect_b64 s[c5_T6:c5_T6+1], -1, 0
  v_cndmask_b32_e64 v[c5_CPY], v[c5_S2], v[c5_S4], s[c5_T4:c5_T4+1]
  v_cndmask_b32_e64 v[c5_CPY], v[c5_CPY], v[c5_S8], s[c5_T6:c5_T6+1]
  s_mov_b64 exec, s[c5_LM:c5_LM+1]
  global_store_byte v[c5_B8], v[c5_CPY], s[c5_DST:c5_DST+1]
  s_mov_b64 exec, -1
  ; ---- the new context: the last four bytes pushed (cr-coder.c:279); they sit in the lanes that copied them
  s_cmp_lt_u32 s[c5_SYM], 4
  s_cbranch_scc1 .Lc5_m_short_%=
  s_sub_u32 s[c5_T0], s[c5_SYM], 4
  s_sub_u32 s[c5_T1], s[c5_SYM], 3
  s_sub_u32 s[c5_T2], s[c5_SYM], 2
  s_sub_u32 s[c5_T3], s[c5_SYM], 1
  v_readlane_b32 s[c5_T0], v[c5_CPY], s[c5_T0]
  v_readlane_b32 s[c5_T1], v[c5_CPY], s[c5_T1]
  v_readlane_b32 s[c5_T2], v[c5_CPY], s[c5_T2]
  v_readlane_b32 s[c5_T3], v[c5_CPY], s[c5_T3]
  s_lshl_b32 s[c5_T0], s[c5_T0], 24
  s_lshl_b32 s[c5_T1], s[c5_T1], 16
  s_lshl_b32 s[c5_T2], s[c5_T2], 8
  s_or_b32 s[c5_T0], s[c5_T0], s[c5_T1]
  s_or_b32 s[c5_T2], s[c5_T2], s[c5_T3]
  s_or_b32 s[c5_NCTX], s[c5_T0], s[c5_T2]
.Lc5_m_ctx_%=:
  c5_issue c5_NCTX
  c5_lzp_round c5_R8, c5_A8, c5_D8, c5_VH8, %[off8], c5_PM8, %=
  c5_lzp_round c5_R4, c5_A4, c5_D4, c5_VH4, %[off4], c5_PM4, %=
  ; ---- the copied positions become pending: lane i wrote byte have + i; XA = the 8 bytes ending there (X8 fills in
  ; from below), and the pending registers hold the 8 bytes in FRONT of each position, one lane further up
  v_lshlrev_b32 v[c5_XAHI], 24, v[c5_CPY]
  s_lshr_b32 s[c5_T0], s[c5_X8HI], 24
  v_mov_b32 v[c5_XT], s[c5_T0]
  s_bfe_u32 s[c5_T0], s[c5_X8HI], 0x80010
  v_mov_b32 v[c5_XU], s[c5_T0]
  v_mov_b32_dpp v[c5_XT], v[c5_CPY] wave_shr:1 row_mask:0xf bank_mask:0xf
  s_bfe_u32 s[c5_T0], s[c5_X8HI], 0x80008
  s_nop 0
  v_mov_b32_dpp v[c5_XU], v[c5_XT] wave_shr:1 row_mask:0xf bank_mask:0xf
  v_lshl_or_b32 v[c5_XAHI], v[c5_XT], 16, v[c5_XAHI]
  v_mov_b32 v[c5_XT], s[c5_T0]
  v_lshl_or_b32 v[c5_XAHI], v[c5_XU], 8, v[c5_XAHI]
  s_and_b32 s[c5_T0], s[c5_X8HI], 0xff
  v_mov_b32_dpp v[c5_XT], v[c5_XU] wave_shr:1 row_mask:0xf bank_mask:0xf
  v_mov_b32 v[c5_XU], s[c5_T0]
  v_or_b32 v[c5_XAHI], v[c5_XAHI], v[c5_XT]
  s_lshr_b32 s[c5_T0], s[c5_X8LO], 24
  v_mov_b32_dpp v[c5_XU], v[c5_XT] wave_shr:1 row_mask:0xf bank_mask:0xf
  v_mov_b32 v[c5_XT], s[c5_T0]
  v_lshlrev_b32 v[c5_XALO], 24, v[c5_XU]
  s_bfe_u32 s[c5_T0], s[c5_X8LO], 0x80010
  v_mov_b32_dpp v[c5_XT], v[c5_XU] wave_shr:1 row_mask:0xf bank_mask:0xf
  v_mov_b32 v[c5_XU], s[c5_T0]
  v_lshl_or_b32 v[c5_XALO], v[c5_XT], 16, v[c5_XALO]
  s_bfe_u32 s[c5_T0], s[c5_X8LO], 0x80008
  v_mov_b32_dpp v[c5_XU], v[c5_XT] wave_shr:1 row_mask:0xf bank_mask:0xf
  v_mov_b32 v[c5_XT], s[c5_T0]
  v_lshl_or_b32 v[c5_XALO], v[c5_XU], 8, v[c5_XALO]
  v_mov_b32 v[c5_PENDHI], s[c5_X8HI]
  v_mov_b32_dpp v[c5_XT], v[c5_XU] wave_shr:1 row_mask:0xf bank_mask:0xf
  v_mov_b32 v[c5_PENDLO], s[c5_X8LO]
  v_or_b32 v[c5_XALO], v[c5_XALO], v[c5_XT]
  s_sub_u32 s[c5_T0], s[c5_SYM], 1
  v_mov_b32_dpp v[c5_PENDHI], v[c5_XAHI] wave_shr:1 row_mask:0xf bank_mask:0xf
  s_nop 0
  v_mov_b32_dpp v[c5_PENDLO], v[c5_XALO] wave_shr:1 row_mask:0xf bank_mask:0xf
  v_readlane_b32 s[c5_X8HI], v[c5_XAHI], s[c5_T0]
  v_readlane_b32 s[c5_X8LO], v[c5_XALO], s[c5_T0]
  s_mov_b32 s[c5_LEARNED], s[c5_HAVE]
  s_add_u32 s[c5_HAVE], s[c5_HAVE], s[c5_SYM]
  s_mov_b32 s[c5_CTX], s[c5_NCTX]
  s_mov_b32 s[c5_EV], 0
  s_add_u32 s[c5_T0], s[c5_LEARNED], 64
  s_min_u32 s[c5_LIMIT], s[c5_T0], s[c5_TOTAL]
  s_cmp_ge_u32 s[c5_WIDX], 62
  s_cselect_b32 s[c5_LIMIT], 0, s[c5_LIMIT]
  c5_prof_begin 5
  s_waitcnt vmcnt(0)
  c5_prof_end 5
  c5_prof_begin 6
  c5_lzp_finish2 %[off8], %[off4], %=
  c5_prof_end 6
  c5_prof_end 2
  s_branch .Lc5_after_event_%=
.Lc5_m_short_%=:                                   ; fewer than four bytes: pushed one by one
  s_mov_b32 s[c5_NCTX], s[c5_CTX]
  v_readlane_b32 s[c5_T0], v[c5_CPY], 0
  s_lshl_b32 s[c5_NCTX], s[c5_NCTX], 8
  s_or_b32 s[c5_NCTX], s[c5_NCTX], s[c5_T0]
  s_cmp_eq_u32 s[c5_SYM], 1
  s_cbranch_scc1 .Lc5_m_ctx_%=
  v_readlane_b32 s[c5_T0], v[c5_CPY], 1
  s_lshl_b32 s[c5_NCTX], s[c5_NCTX], 8
  s_or_b32 s[c5_NCTX], s[c5_NCTX], s[c5_T0]
  s_cmp_eq_u32 s[c5_SYM], 2
  s_cbranch_scc1 .Lc5_m_ctx_%=
  v_readlane_b32 s[c5_T0], v[c5_CPY], 2
  s_lshl_b32 s[c5_NCTX], s[c5_NCTX], 8
  s_or_b32 s[c5_NCTX], s[c5_NCTX], s[c5_T0]
  s_branch .Lc5_m_ctx_%=
.Lc5_m_probe8_%=:
  s_mul_i32 s[c5_MH8], s[c5_MK8], 0x9e3779b1
  s_lshr_b32 s[c5_MH8], s[c5_MH8], %[lzsh]
  c5_lzp_probe c5_C8, 8, c5_MH8, c5_E8, c5_LA8, %[off8], %=
  s_branch .Lc5_m_got8_%=
.Lc5_m_probe4_%=:
  s_mul_i32 s[c5_MH4], s[c5_MK4], 0x9e3779b1
  s_lshr_b32 s[c5_MH4], s[c5_MH4], %[lzsh]
  c5_lzp_probe c5_C4, 4, c5_MH4, c5_E4, c5_LA4, %[off4], %=
  s_branch .Lc5_m_got4_%=
.Lc5_m_batch_%=:                                   ; T0:1 / T2:3 / T4:5 = pending lanes with X8's 8- / 4- / 2-byte key: the highest wins
  s_flbit_i32_b64 s[c5_E8K], s[c5_T0:c5_T0+1]
  s_flbit_i32_b64 s[c5_E8P], s[c5_T2:c5_T2+1]
  s_flbit_i32_b64 s[c5_E4K], s[c5_T4:c5_T4+1]
  s_add_u32 s[c5_E4P], s[c5_LEARNED], 63
  s_sub_u32 s[c5_E8K], s[c5_E4P], s[c5_E8K]
  s_sub_u32 s[c5_E8P], s[c5_E4P], s[c5_E8P]
  s_sub_u32 s[c5_E4K], s[c5_E4P], s[c5_E4K]
  s_cmp_lg_u64 s[c5_T0:c5_T0+1], 0
  s_cselect_b32 s[c5_C8], s[c5_E8K], s[c5_C8]
  s_cmp_lg_u64 s[c5_T2:c5_T2+1], 0
  s_cselect_b32 s[c5_C4], s[c5_E8P], s[c5_C4]
  s_cmp_lg_u64 s[c5_T4:c5_T4+1], 0
  s_cselect_b32 s[c5_C2], s[c5_E4K], s[c5_C2]
  s_branch .Lc5_m_cand_%=
.Lc5_m_overlap_%=:                                 ; get the pending positions into the tables and hand over with nothing pending
  s_waitcnt vmcnt(0)
  c5_lzp_finish2 %[off8], %[off4], %=
  s_mov_b32 s[c5_LEARNED], s[c5_HAVE]
.Lc5_m_slow_%=:
  s_mov_b32 s[c5_EV], 1
  s_branch .Lc5_exit_%=
.Lc5_m_issued_%=:
  s_mov_b32 s[c5_EV], 0
  s_branch .Lc5_after_event_%=

.if c5_mode == 0
  ; ---- 64 literals are waiting to be learned (cr_lzp_learn, crgpu_lzp.h; matcher_update, cr-matcher.c:75-96), round 5: here, not in
  ; the C++ around the statement — leaving it drains every store, and coming back costs two more dependent round trips (the
  ; scratch line, the context's model): ~5 000 clocks per 64 literals, a fourteenth of a lone block's time. The first half of
  ; .Lc5_m_issue (the inserts of the 64 pending lanes), waited for, collisions walked on; the model registers of the step that
  ; follows are not touched. With a match token's table work in flight (its registers are these) the old way out is taken.
.Lc5_learn_%=:
  s_cmp_lg_u32 s[c5_AESC], 0
  s_cbranch_scc1 .Lc5_exit_learn_%=
  c5_learn_drain %[off8], %[off4], %[lzsh], %=     ; (the one before, if its walks have outlasted 64 steps)
  s_lshr_b32 s[c5_LZM], -1, %[lzsh]
  v_add_u32 v[c5_VQ], s[c5_LEARNED], v[c5_LANE]
  v_alignbit_b32 v[c5_VT0], v[c5_PENDHI], v[c5_PENDLO], 20
  v_lshrrev_b32 v[c5_VT1], 8, v[c5_PENDHI]
  v_xor_b32 v[c5_VK8], v[c5_PENDLO], v[c5_VT0]
  v_xor_b32 v[c5_VK8], v[c5_VK8], v[c5_VT1]
  v_and_b32 v[c5_VK8], 0xffffff, v[c5_VK8]
  v_lshrrev_b32 v[c5_VT0], 6, v[c5_PENDHI]
  v_lshrrev_b32 v[c5_VT1], 12, v[c5_PENDHI]
  v_xor_b32 v[c5_VK4], v[c5_PENDHI], v[c5_VT0]
  v_xor_b32 v[c5_VK4], v[c5_VK4], v[c5_VT1]
  v_and_b32 v[c5_VK4], 0xfffff, v[c5_VK4]
  v_lshrrev_b32 v[c5_VK2], 16, v[c5_PENDHI]
  s_mov_b32 s[c5_T0], 0x9e3779b1
  v_mul_lo_u32 v[c5_VH8], v[c5_VK8], s[c5_T0]
  v_mul_lo_u32 v[c5_VH4], v[c5_VK4], s[c5_T0]
  v_lshrrev_b32 v[c5_VH8], %[lzsh], v[c5_VH8]
  v_lshrrev_b32 v[c5_VH4], %[lzsh], v[c5_VH4]
  v_lshlrev_b32 v[c5_A8], 3, v[c5_VH8]
  v_lshlrev_b32 v[c5_A4], 3, v[c5_VH4]
  v_lshlrev_b32 v[c5_A2], 2, v[c5_VK2]
  v_add_u32 v[c5_A8], %[off8], v[c5_A8]
  v_add_u32 v[c5_A4], %[off4], v[c5_A4]
  v_add_u32 v[c5_A2], %[off2], v[c5_A2]
  v_mov_b32 v[c5_D8], v[c5_VQ]
  v_add_u32 v[c5_D8+1], 1, v[c5_VK8]
  v_mov_b32 v[c5_D8+2], 0
  v_mov_b32 v[c5_D8+3], 0
  v_mov_b32 v[c5_D4], v[c5_VQ]
  v_add_u32 v[c5_D4+1], 1, v[c5_VK4]
  v_mov_b32 v[c5_D4+2], 0
  v_mov_b32 v[c5_D4+3], 0
  global_atomic_cmpswap_x2 v[c5_R8:c5_R8+1], v[c5_A8], v[c5_D8:c5_D8+3], s[c5_ARENA:c5_ARENA+1] sc0
  global_atomic_cmpswap_x2 v[c5_R4:c5_R4+1], v[c5_A4], v[c5_D4:c5_D4+3], s[c5_ARENA:c5_ARENA+1] sc0
  global_atomic_umax v[c5_A2], v[c5_VQ], s[c5_ARENA:c5_ARENA+1]
.if c5_alearn
  v_mov_b32 v[c5_PMV8], 1                          ; every lane's first round is out; the steps' ends take it from here
  v_mov_b32 v[c5_PMV4], 1
  v_mov_b32 v[c5_FLV], 1
  s_mov_b32 s[c5_LEARNED], s[c5_HAVE]
  c5_prio %=
  s_mov_b32 s[c5_EV], 0
  s_mov_b32 s[c5_T5], 1
  s_branch .Lc5_limit_go_%=
.endif
  s_mov_b64 s[c5_PM8:c5_PM8+1], -1
  s_mov_b64 s[c5_PM4:c5_PM4+1], -1
  s_waitcnt vmcnt(0)
  c5_lzp_finish2 %[off8], %[off4], %=              ; (both tables' collision rounds share their round trips)
  s_mov_b32 s[c5_LEARNED], s[c5_HAVE]
  c5_prio %=
  s_add_u32 s[c5_T0], s[c5_LEARNED], 64
  s_min_u32 s[c5_LIMIT], s[c5_T0], s[c5_TOTAL]
  s_cmp_ge_u32 s[c5_WIDX], 62
  s_cselect_b32 s[c5_LIMIT], 0, s[c5_LIMIT]
  s_mov_b32 s[c5_EV], 0
  s_branch .Lc5_after_event_%=
.endif
.Lc5_fail_%=:
  s_mov_b32 s[c5_EV], 5
  s_branch .Lc5_exit_%=
.Lc5_exit_learn_%=:
  s_mov_b32 s[c5_EV], 2
  s_branch .Lc5_exit_%=
.Lc5_exit_window_%=:
  s_mov_b32 s[c5_EV], 3
.Lc5_exit_%=:
  c5_learn_drain %[off8], %[off4], %[lzsh], %=
  s_waitcnt vmcnt(0) lgkmcnt(0)                    ; (a presence read may still be out: its register must not be written once the statement has ended)
  s_mov_b32 %[ctx], s[c5_CTX]
  v_readfirstlane_b32 %[range], v[c5_VRANGE]
  v_readfirstlane_b32 %[cache], v[c5_VCACHE]
  v_readfirstlane_b32 %[iblo], v[c5_VIBLO]
  v_readfirstlane_b32 %[ibhi], v[c5_VIBHI]
  v_readfirstlane_b32 %[ibits], v[c5_VIBITS]
  s_mov_b32 %[widx], s[c5_WIDX]
  s_mov_b32 %[have], s[c5_HAVE]
  s_mov_b32 %[learned], s[c5_LEARNED]
  s_mov_b32 %[aesc], s[c5_AESC]
  s_mov_b32 %[x8lo], s[c5_X8LO]
  s_mov_b32 %[x8hi], s[c5_X8HI]
  s_mov_b32 %[ev], s[c5_EV]
  s_mov_b32 %[sym], s[c5_SYM]
  v_mov_b32 %[pacc], v[c5_PACC]
  v_mov_b32 %[pcnt], v[c5_PCNT]
  v_mov_b32 %[plo], v[c5_PENDLO]
  v_mov_b32 %[phi], v[c5_PENDHI]
  v_mov_b32 v[c5_VT0], c5_OFF_SCR+904
  global_store_dword v[c5_VT0], v[c5_VDSLOT], s[c5_ARENA:c5_ARENA+1]
.if c5_hw
  v_mov_b32 v[c5_VT1], s[c5_HWSEQ]
  ds_write_b32 v[c5_HWB], v[c5_VT1] offset:12
.endif
  s_waitcnt vmcnt(0) lgkmcnt(0)
)ASM"

#define CR_V5_CLOBBERS \
    "s34", "s35", "s36", "s37", "s38", "s39", "s40", "s41", "s42", "s43", "s44", "s45", "s46", "s47", "s48", "s49", \
    "s50", "s51", "s52", "s53", "s54", "s55", "s56", "s57", "s58", "s59", "s60", "s61", "s62", "s63", "s64", "s65", "s66", "s67", \
    "s68", "s69", "s70", "s71", "s72", "s73", "s74", "s75", "s76", "s77", "s78", "s79", "s80", "s81", "s82", "s83", "s84", "s85", \
    "s86", "s87", "s88", "s89", "s90", "s91", "s92", "s93", "s94", "s95", "s96", "s97", "s98", "s99", \
    "v32", "v33", "v34", "v35", "v36", "v37", "v38", "v39", "v40", "v41", "v42", "v43", "v44", "v45", "v46", "v47", "v48", "v49", \
    "v50", "v51", "v52", "v53", "v54", "v55", "v56", "v57", "v58", "v59", "v60", "v61", "v62", "v63", "v64", "v65", "v66", "v67", \
    "v68", "v69", "v70", "v71", "v72", "v73", "v74", "v75", "v76", "v77", "v78", "v79", "v80", "v81", "v82", "v83", "v84", "v85", \
    "v86", "v87", "v88", "v89", "v90", "v91", "v92", "v93", "v94", "v95", "v96", "v97", "v98", "v99", "v100", "v101", "v102", "v103", \
    "v104", "v105", "v106", "v107", "v108", "v109", "v110", "v111", "v112", "v113", "v114", "v115", "v116", "v117", "v118", "v119", "v120", \
    "v121", "v122", "v123", "v124", "v125", "v126", "v127", "v128", "v129", "v130", "v131", "v132", "v133", "v134", "v135", "v136", "v137", "v138", \
    "v139", "v140", "v141", "v142", "v143", "v144", "v145", "s100", "s101", "vcc", "scc", "memory"

/* The helper wave of CRGPU_OPT_DECODER_HELPER (the mailbox's layout: CR_V5_ASM_MODE_HW above): waits for a post, computes what the
 * escape path of crgpu_rop5.h's step computes between "the node's bytes as a presence map" and "the scan's total" — same
 * instructions, same order, cr-ppm.c:209-211 — and leaves it in the mailbox under the post's sequence number. Runs at the
 * lowest priority (the coder waves of other blocks share its SIMD) and ends on the stop word. */
CR_DEV void cr_rop_decode_helper(uint32_t mb_at) {
    asm volatile(R"ASM(
  s_setprio 0
  v_mbcnt_lo_u32_b32 v32, -1, 0
  v_mbcnt_hi_u32_b32 v32, -1, v32                  ; v32 lane
  v_mov_b32 v33, %[mb]                             ; v33 mailbox
  v_lshl_add_u32 v34, v32, 1, v33                  ; v34 + 16: the lane's pair
  v_lshl_add_u32 v35, v32, 2, v33                  ; v35 + 144: the lane's row word
  v_lshl_add_u32 v36, v32, 4, v33                  ; v36 + 400: the lane's four results
  v_mov_b32 v50, 0xff
  v_cmp_gt_u32 vcc, 62, v32
  v_mov_b32 v51, 0x100
  v_mov_b32 v40, 0                                 ; v40 zero, v39 one
  v_cndmask_b32 v37, 0, v50, vcc                   ; v37 0xff below lane 62
  v_cndmask_b32 v38, v51, v40, vcc                 ; v38 0x100 from lane 62 on
  v_mov_b32 v39, 1
  v_add_u32 v41, 0x590, v33                        ; v41 the helper's own 256 scratch bytes (+1424), v43 the spare byte, v42 the lane's dword
  v_add_u32 v43, 0x100, v41
  v_lshl_add_u32 v42, v32, 2, v41
  s_nop 0
  ds_write_b32 v42, v40
  s_mov_b32 s35, 0                                 ; s35 the last sequence number answered
.Lch_poll_%=:
  ds_read_b32 v60, v33
  s_waitcnt lgkmcnt(0)
  v_readfirstlane_b32 s34, v60                     ; s34 the post word
  s_cmp_eq_u32 s34, -1
  s_cbranch_scc1 .Lch_exit_%=
  s_andn2_b32 s36, s34, 0x1ff                      ; s36 its sequence number
  s_cmp_eq_u32 s36, s35
  s_cbranch_scc0 .Lch_work_%=
  s_branch .Lch_poll_%=                            ; (a tight poll: with s_sleep 1 here the kernel takes a third of a percent longer)
.Lch_work_%=:
  ds_read_u16 v44, v34 offset:16                   ; v44 pair, v45 row
  ds_read_b32 v45, v35 offset:144
  s_and_b32 s37, s34, 0xff                         ; s37 the predicted byte
  s_and_b32 s38, s37, 3
  s_lshl_b32 s38, s38, 3
  s_lshl_b32 s39, 0xff, s38                        ; s39 its byte in its lane's word
  s_lshr_b32 s38, s37, 2
  s_waitcnt lgkmcnt(0)
  v_and_b32 v46, v37, v44                          ; v46 the pairs' counts
  v_lshrrev_b32 v47, 8, v44
  v_or_b32 v47, v38, v47                           ; v47 the pairs' symbols
  v_add_u32 v48, v41, v47
  v_cmp_ne_u32 vcc, 0, v46
  v_mov_b32 v50, s39
  s_nop 0
  v_cndmask_b32 v48, v43, v48, vcc
  ds_write_b8 v48, v39
  ds_read_b32 v49, v42                             ; v49 0x01 in every byte the node holds
  ds_write_b8 v48, v40
  v_cmp_eq_u32 vcc, s38, v32
  s_nop 1
  v_cndmask_b32 v52, 0, v50, vcc                   ; v52 the predicted byte's place
  s_waitcnt lgkmcnt(0)
  v_xor_b32 v50, 0x01010101, v49
  v_bfi_b32 v50, v52, 0, v50                       ; the candidates (cr-ppm.c:150-155)
  v_lshlrev_b32 v51, 8, v50
  v_sub_u32 v53, v51, v50
  v_and_b32 v54, v45, v53
  v_sub_u32 v54, v54, v50                          ; count - 1 of every candidate
  v_and_b32 v57, 0x00ff00ff, v54
  v_lshrrev_b32 v58, 8, v54
  v_and_b32 v51, 0x00ff00ff, v50
  v_lshrrev_b32 v50, 8, v50
  v_and_b32 v58, 0x00ff00ff, v58
  v_and_b32 v50, 0x00ff00ff, v50
  v_lshl_add_u32 v57, v57, 3, v51                  ; v57 / v58: 8 c - 7 of the even / odd bytes as 16-bit fields
  v_lshl_add_u32 v58, v58, 3, v50
  v_add_u32 v50, v57, v58
  v_add_u32_sdwa v55, v50, v50 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_0 src1_sel:WORD_1
  s_nop 1
  v_add_u32_dpp v59, v55, v55 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1
  s_nop 1
  v_add_u32_dpp v59, v59, v59 row_shr:2 row_mask:0xf bank_mask:0xf bound_ctrl:1
  s_nop 1
  v_add_u32_dpp v59, v59, v59 row_shr:4 row_mask:0xf bank_mask:0xf bound_ctrl:1
  s_nop 1
  v_add_u32_dpp v59, v59, v59 row_shr:8 row_mask:0xf bank_mask:0xf bound_ctrl:1
  s_nop 1
  v_add_u32_dpp v59, v59, v59 row_bcast:15 row_mask:0xa bank_mask:0xf
  s_nop 1
  v_add_u32_dpp v59, v59, v59 row_bcast:31 row_mask:0xc bank_mask:0xf
  s_nop 0
  v_readlane_b32 s38, v59, 63                      ; the total
  v_sub_u32 v56, v59, v55                          ; v56 .. v59: below the lane, even weights, odd weights, prefix
  v_mov_b32 v50, s38
  v_mov_b32 v51, s36
  ds_write_b128 v36, v[56:59] offset:400
  ds_write_b32 v33, v50 offset:8
  ds_write_b32 v33, v51 offset:4
  s_mov_b32 s35, s36
  s_branch .Lch_poll_%=
.Lch_exit_%=:
  s_waitcnt lgkmcnt(0)
)ASM"
                 :: [mb] "s"(mb_at)
                 : "s34", "s35", "s36", "s37", "s38", "s39", "v32", "v33", "v34", "v35", "v36", "v37", "v38", "v39", "v40", "v41", "v42", "v43", "v44",
                   "v45", "v46", "v47", "v48", "v49", "v50", "v51", "v52", "v53", "v54", "v55", "v56", "v57", "v58", "v59", "v60", "vcc", "scc", "memory");
}

template <int HELPER, int DL>
CR_DEV uint32_t cr_rop_decode_v5(const uint8_t* src_, uint32_t n, uint8_t* dst_, uint32_t cap, uint8_t* arena_,
                                 const CrArenaLayout& L, uint32_t lds_scratch, u64* st) {
    const uint8_t* const src = cr_uni_ptr(src_);
    uint8_t* const dst = cr_uni_ptr(dst_);
    uint8_t* const arena = cr_uni_ptr(arena_);
    n = cr_uni(n); cap = cr_uni(cap);
    cr_stamp(st, 0);
    const uint32_t lane = cr_lane();
    if (n < CR_ROP_HEADER) return 0xFFFFFFFFu;
    if (src[0] == 0) {                                                   /* cr-coder.c:243-248 */
        uint32_t raw = n - CR_ROP_HEADER;
        if (raw > cap) return 0xFFFFFFFFu;
        for (uint32_t i = lane; i < raw; i += CRGPU_WAVE) dst[i] = src[CR_ROP_HEADER + i];
        return raw;
    }
    const uint32_t total = cr_uni((uint32_t)src[4] | ((uint32_t)src[5] << 8) | ((uint32_t)src[6] << 16) | ((uint32_t)src[7] << 24));
    const uint32_t esc = cr_uni(src[8]);
    if (total > cap || total < CR_LZP_SKIP || total > L.max_block) return 0xFFFFFFFFu;
    if (lane < CR_LZP_SKIP) dst[lane] = src[9u + lane];                  /* cr-coder.c:251-254 */

    CrLzp z;
    cr_lzp_attach(z, arena, L, cr_log2_ceil_pow2(2u * total, 1024u, L.cap_lz));
    cr_lzp_reset(z);
    uint32_t g3_;
    const uint32_t gen = cr_uni(cr_v3_reset(arena, L, g3_));
    const uint32_t g3 = cr_uni(g3_);
    /* where the dense slots start, the LDS address of the wave's 256 scratch bytes, the next free dense slot: the statement
     * reads them from the arena's scratch line (and writes the slot counter back there when it is left) */
    if (lane == 0) {
        uint32_t* scr = reinterpret_cast<uint32_t*>(arena + CRGPU_OFF_SCRATCH + 896u);
        scr[0] = (uint32_t)L.off_dense; scr[1] = lds_scratch; scr[2] = 0u;
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
    cr_wave_sync();
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");

    /* coded bytes: cache = bytes 1..4 (range_decoder_init, cr-rangecoder.c:81-89), then a 64-bit shift register
     * refilled one dword at a time from a 256-byte big-endian register window (crgpu_dec.h) */
    const uint8_t* const payload = src + CR_ROP_HEADER;
    const uint32_t psize = n - CR_ROP_HEADER;
    uint32_t wbase = 0, win = cr_v4_window(payload, psize, 0u);
    uint32_t cache = cr_lane_get(win, 0), range = 0xFFFFFFFFu;
    uint32_t ib_hi = cr_lane_get(win, 1), ib_lo = cr_lane_get(win, 2), ibits = 64, widx = 3;
    uint32_t ctx = 0, have = CR_LZP_SKIP, learned = CR_LZP_SKIP, after_esc = 0;
    u64 x8 = *reinterpret_cast<const cr_u64u*>(src + 10);                 /* the 8 bytes in front of the write position */
    uint32_t x8_lo = cr_uni((uint32_t)x8), x8_hi = cr_uni((uint32_t)(x8 >> 32));
    uint32_t pend_lo = 0, pend_hi = 0;                                   /* lane j: those 8 bytes for position learned + j */
    const uint32_t off8 = cr_uni((uint32_t)L.off_lz8), off4 = cr_uni((uint32_t)L.off_lz4), off2 = cr_uni((uint32_t)L.off_lz2);
    const uint32_t lzsh = cr_uni(z.shift);
    const uint32_t dslots = cr_uni(L.dense_slots);
    cr_stamp(st, 4);
    if (st && lane == 0)                                                 /* tools/dec_blocks.py: where the wave runs (HW_ID | XCC_ID << 32) */
        st[6] = (u64)(uint32_t)__builtin_amdgcn_s_getreg(63492) | ((u64)(uint32_t)__builtin_amdgcn_s_getreg(63508) << 32);
#ifdef CR_V5_PROF
    u64 pf_wait = 0, pf_steps = 0, pf_calls = 0;
    const u64 pf_t0 = __builtin_amdgcn_s_memtime();
#endif

    while (have < total) {                                               /* cr-coder.c:259-290 */
        uint32_t ev, sym, pacc, pcnt;
#define CR_V5_STATEMENT(mode_) \
        asm volatile(mode_ CR_V5_ASM_DEFS CR_V5_ASM_MACROS CR_V5_ASM_BODY \
                     : [ctx] "+s"(ctx), [range] "+s"(range), [cache] "+s"(cache), [iblo] "+s"(ib_lo), [ibhi] "+s"(ib_hi), \
                       [ibits] "+s"(ibits), [widx] "+s"(widx), [have] "+s"(have), [learned] "+s"(learned), [aesc] "+s"(after_esc), \
                       [x8lo] "+s"(x8_lo), [x8hi] "+s"(x8_hi), [ev] "=&s"(ev), [sym] "=&s"(sym), [plo] "+v"(pend_lo), [phi] "+v"(pend_hi), \
                       [pacc] "=&v"(pacc), [pcnt] "=&v"(pcnt) \
                     : [win] "v"(win), [arena] "s"(arena), [dst] "s"(dst), [total] "s"(total), [gen] "s"(gen), [g3] "s"(g3), [esc] "s"(esc), \
                       [cap] "s"(cap), [off8] "s"(off8), [off4] "s"(off4), [off2] "s"(off2), [lzsh] "s"(lzsh), [dslots] "s"(dslots) \
                     : CR_V5_CLOBBERS)
        if (HELPER) CR_V5_STATEMENT(CR_V5_ASM_MODE_HW); else if (DL) CR_V5_STATEMENT(CR_V5_ASM_MODE_DL(0)); else CR_V5_STATEMENT(CR_V5_ASM_MODE(0));
#undef CR_V5_STATEMENT
        ev = cr_uni(ev);
#ifdef CR_V5_PROF
        pf_wait += cr_uni(pacc); pf_steps += cr_uni(pcnt); pf_calls++;
#endif
        if (ev == CR_V5_EV_DONE) break;
        if (ev == CR_V5_EV_LEARN) {
            cr_lzp_learn(z, ((u64)pend_hi << 32) | pend_lo, learned + lane);
            learned = have;
            continue;
        }
        if (ev == CR_V5_EV_WINDOW) {
            wbase += widx * 4u;
            win = cr_v4_window(payload, psize, wbase);
            widx = 0;
            continue;
        }
        if (ev != CR_V5_EV_MATCH) return 0xFFFFFFFFu;                    /* step limit: a damaged stream */
        {
            const uint32_t len = cr_uni(sym);
            const u64 x8v = ((u64)x8_hi << 32) | x8_lo;
            if (have + len > total || have + len > cap) return 0xFFFFFFFFu;   /* corrupt stream */
            uint32_t c8, c4, c2;
            cr_lzp_learn_predict(z, ((u64)pend_hi << 32) | pend_lo, learned, have - learned, x8v, c8, c4, c2);
            learned = have;
            /* matcher_getpos' two context checks (cr-matcher.c:59-73) and the first 64 source bytes of all
             * three candidates in one round trip; a source that overlaps the destination repeats with
             * period have - from (byte-serial copy, cr-coder.c:277-279) */
            const uint32_t p8 = have - c8, p4 = have - c4, p2 = have - c2;
            const uint32_t r8 = (len > p8) ? lane % p8 : lane, r4 = (len > p4) ? lane % p4 : lane, r2 = (len > p2) ? lane % p2 : lane;
            const u64 v8 = *reinterpret_cast<const cr_u64u*>(dst + c8 - 8);
            const uint32_t v4 = *reinterpret_cast<const cr_u32u*>(dst + c4 - 4);
            uint32_t s8 = 0, s4 = 0, s2 = 0;
            if (lane < len) { s8 = dst[c8 + r8]; s4 = dst[c4 + r4]; s2 = dst[c2 + r2]; }
            uint32_t from = c2, mine = s2;
            if (v8 == x8v) { from = c8; mine = s8; }
            else if (v4 == x8_hi) { from = c4; mine = s4; }
            from = cr_uni(from);
            if (lane < len) dst[have + lane] = (uint8_t)mine;
            const uint32_t period = have - from;
            for (uint32_t i0 = CRGPU_WAVE; i0 < len; i0 += CRGPU_WAVE) {
                uint32_t i = i0 + lane;
                if (i < len) {
                    uint32_t r = i < period ? i : i % period;
                    mine = dst[from + r];
                    dst[have + i] = (uint8_t)mine;
                }
            }
            /* only the last four pushes survive in the 32-bit context; they sit in the lanes that copied them */
            if (len >= 4u && ((len - 1u) & 63u) >= 3u) {
                uint32_t l3 = (len - 1u) & 63u;
                ctx = (cr_lane_get(mine, l3 - 3u) << 24) | (cr_lane_get(mine, l3 - 2u) << 16) |
                      (cr_lane_get(mine, l3 - 1u) << 8) | cr_lane_get(mine, l3);
            } else {
                cr_wave_sync();
                uint32_t k = len < 4u ? len : 4u;
                for (uint32_t i = len - k; i < len; i++) ctx = (ctx << 8) | cr_uni(dst[have + i]);
            }
            if (len < CRGPU_WAVE) {
                /* the copied positions become pending: lane i held byte have+i; xa = the 8 bytes ending there */
                uint32_t t = mine & 0xffu;
                u64 xa = (u64)t << 56;
#pragma unroll
                for (uint32_t k = 1; k < 8u; k++) {
                    t = cr_shift_up1(t, (uint32_t)(x8v >> (8u * (8u - k))) & 0xffu);
                    xa |= (u64)t << (8u * (7u - k));
                }
                pend_lo = cr_shift_up1((uint32_t)xa, x8_lo);             /* lane 0: position have, lane j: have+j */
                pend_hi = cr_shift_up1((uint32_t)(xa >> 32), x8_hi);
                const u64 nx = cr_lane_get64(xa, len - 1u);
                x8_lo = (uint32_t)nx; x8_hi = (uint32_t)(nx >> 32);
                have += len;
            } else {
                cr_wave_sync();
                for (uint32_t q0 = have; q0 < have + len; q0 += CRGPU_WAVE) {
                    uint32_t q = q0 + lane;
                    if (q < have + len) cr_lzp_learn(z, *reinterpret_cast<const cr_u64u*>(dst + q - 8), q);
                }
                have += len;
                learned = have;
                const u64 nx = *reinterpret_cast<const cr_u64u*>(dst + have - 8);
                x8_lo = cr_uni((uint32_t)nx); x8_hi = cr_uni((uint32_t)(nx >> 32));
            }
            ctx = cr_uni(ctx);
            /* the copy's stores need not be waited for here: nothing reads the output before the next match
             * token, and the asm statement drains every store before it hands one over */
        }
    }
#ifdef CR_V5_PROF
    if (st && lane == 0) { st[8] = __builtin_amdgcn_s_memtime() - pf_t0; st[9] = pf_wait; st[10] = 0; st[11] = pf_calls; st[12] = pf_steps; st[13] = 0; }
#endif
    cr_stamp(st, 5);
    return have;
}

#endif

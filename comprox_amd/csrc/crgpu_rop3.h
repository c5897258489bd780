/*
 * comprox_amd/csrc/crgpu_rop3.h — comprop lzdecode for the batched API, third layout of the step.
 *
 * Reference: /root/reference/src/ropmain/cr-coder.c:231-292 (lzdecode), src/cr-ppm.c:169-235
 * (ppm_decode), src/cr-rangecoder.c:81-104 (range decoder), src/cr-o2model.c:93-113 (symbol search).
 *
 * Same wave-per-datablock, write-through step as cr_rop_decode_lean (crgpu_rop.h); what changes is
 * what sits on the serial chain of a block:
 *   - the order-3 predictor is DIRECT-indexed by the reference's 22-bit key (cr-ppm.c:66): one u16
 *     {byte, 4-bit generation, confidence} per key, 8 MB per resident workgroup. No hashing, no probe
 *     group, no slot search; a stale generation reads as the reference's zero-filled entry, so the
 *     per-block reset is one increment (the table is wiped every 15th block of a workgroup);
 *   - range decoder without the second division: the reference computes target = cache / unit and
 *     looks for cum <= target < cum + frq; cum <= floor(cache / unit) <=> cum * unit <= cache, and
 *     cum * unit <= range < 2^32, so the search compares per-lane products with `cache` instead. The
 *     owner lane is the popcount of one ballot;
 *   - the next step's loads are issued TWICE at most: at the top of a step for the context that follows
 *     if the order-3 prediction hits (known before any arithmetic), and after the symbol is known only
 *     when it turned out different. A hit step has its successor's model in flight for the whole step;
 *   - the step's four loads and five stores are issued from inline assembly with explicit wait counts:
 *     the compiler's wait-count pass merges the states of the paths into a loop head conservatively and
 *     ended up draining the previous step's stores (a full store round trip per symbol) in front of
 *     every step. The loads go out as one group once the symbol is known, the five stores follow after
 *     the register updates, and the wait at the bottom of the step is "all but the last five" = the loads
 *     only. Addresses are 32-bit offsets from the arena base (saddr form), no 64-bit address arithmetic.
 */
#ifndef CRGPU_ROP3_H
#define CRGPU_ROP3_H

#include "crgpu_rop.h"

#define CR_O3D_ENTRIES (1u << 22)        /* cr-ppm.c:66: 22-bit key */

/* ppm_model_free + ppm_model_init (cr-ppm.c:34-57) for the direct-indexed tables: node generation,
 * order-3 generation, order-1 rows = 1. Returns the node generation; o3gen by reference. */
CR_DEV uint32_t cr_v3_reset(uint8_t* arena, const CrArenaLayout& L, uint32_t& o3gen) {
    uint32_t* dir = reinterpret_cast<uint32_t*>(arena + L.off_dir);
    uint32_t g = cr_uni(dir[0]) + 1u;
    uint32_t g3 = cr_uni(dir[3]) + 1u;
    if (g > 0xffffu) {
        cr_fill(arena + L.off_nodes, (u64)65536u * CRGPU_NODE_BYTES, 0u);
        g = 1u;
    }
    if (g3 > 15u) {
        cr_fill(arena + L.off_o3d, (u64)CR_O3D_ENTRIES * 2u, 0u);
        g3 = 1u;
    }
    cr_wave_sync();
    if (cr_lane() == 0) { dir[0] = g; dir[3] = g3; }
    cr_fill(arena + L.off_o1, 65536u, 0x01010101u);
    o3gen = g3;
    return g;
}

/* Diagnostic build only (-DCR_V3_PROF): shader-clock sums per block in the stats slots 8..13
 * (loop total, wait for the next model, match tokens, their count, steps, escapes). */
#ifdef CR_V3_PROF
#define CR_V3_T(var_) const u64 var_ = __builtin_amdgcn_s_memtime()
#define CR_V3_ACC(acc_, from_) do { acc_ += __builtin_amdgcn_s_memtime() - (from_); } while (0)
#else
#define CR_V3_T(var_) do { } while (0)
#define CR_V3_ACC(acc_, from_) do { } while (0)
#endif

/* SPEC = 1: issue the loads of the context that follows a prediction hit at the top of the step */
template <int SPEC>
CR_DEV uint32_t cr_rop_decode_v3(const uint8_t* src, uint32_t n, uint8_t* dst, uint32_t cap, uint8_t* arena,
                                 const CrArenaLayout& L, u64* st) {
    cr_stamp(st, 0);
    const uint32_t lane = cr_lane();
    if (n < CR_ROP_HEADER) return 0xFFFFFFFFu;
    if (src[0] == 0) {                                                   /* cr-coder.c:243-248 */
        uint32_t raw = n - CR_ROP_HEADER;
        if (raw > cap) return 0xFFFFFFFFu;
        for (uint32_t i = lane; i < raw; i += CRGPU_WAVE) dst[i] = src[CR_ROP_HEADER + i];
        return raw;
    }
    const uint32_t total = (uint32_t)src[4] | ((uint32_t)src[5] << 8) | ((uint32_t)src[6] << 16) | ((uint32_t)src[7] << 24);
    const uint32_t esc = src[8];
    if (total > cap || total < CR_LZP_SKIP || total > L.max_block) return 0xFFFFFFFFu;
    if (lane < CR_LZP_SKIP) dst[lane] = src[9u + lane];                  /* cr-coder.c:251-254 */

    CrLzp z;
    cr_lzp_attach(z, arena, L, cr_log2_ceil_pow2(2u * total, 1024u, L.cap_lz));
    cr_lzp_reset(z);
    uint32_t g3_;
    const uint32_t gen = cr_uni(cr_v3_reset(arena, L, g3_));
    const uint32_t g3 = cr_uni(g3_);
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
    cr_wave_sync();
    uint8_t* const nodes = arena + L.off_nodes;
    uint16_t* const o3d = reinterpret_cast<uint16_t*>(arena + L.off_o3d);
    uint8_t* const o1 = arena + L.off_o1;
    uint8_t* const scratch = arena + L.off_dir + 4096u;                   /* 1 KiB nobody reads */

    CrSource in;
    cr_source_init(in, src + CR_ROP_HEADER, n - CR_ROP_HEADER);
    CrRc rc; cr_rc_dec_init(rc, in);

    uint32_t ctx = 0;
    uint32_t nd_key = 0xFFFFFFFFu, nd_w = 0, nd_x = 0;                   /* the node of the previous step, as stored */
    uint32_t o3_lk = 0xFFFFFFFFu, o3_lv = 0;                             /* last order-3 store: key, value */
    uint32_t lr_idx = 0xFFFFFFFFu, lr_row = 0;                           /* last order-1 row store */
    uint32_t have = CR_LZP_SKIP, learned = CR_LZP_SKIP, after_esc = 0;
    u64 x8 = *reinterpret_cast<const cr_u64u*>(src + 10);
    x8 = ((u64)cr_uni((uint32_t)(x8 >> 32)) << 32) | cr_uni((uint32_t)x8);
    u64 pend_x = 0;
    cr_stamp(st, 4);

    /* two register sets for a step's model: S = issued at the top of the previous step for the context
     * that follows a prediction hit, N = issued once the previous symbol was known (only if different) */
    uint32_t s_w = 0, s_x = 0, s_row = 0, s_e = 0, n_w = 0, n_x = 0, n_row = 0, n_e = 0;
    uint32_t spec_ctx = 0;
    /* lane-varying parts of the load offsets, once */
    const uint32_t vo_nodes = (uint32_t)L.off_nodes + lane * 4u, vo_o1 = (uint32_t)L.off_o1 + lane * 4u;
    const uint32_t so_nodes = (uint32_t)L.off_nodes, so_o3d = (uint32_t)L.off_o3d, so_scr = (uint32_t)L.off_dir + 4096u;
#define CR_V3_ISSUE(c_, w_, x_, e_, row_) do { \
        const uint32_t no_ = ((c_) & 0xffffu) * CRGPU_NODE_BYTES; \
        const uint32_t aw_ = vo_nodes + no_, ax_ = so_nodes + no_; \
        const uint32_t ae_ = so_o3d + ((cr_o3_key(c_) << 1) & ~3u), ar_ = vo_o1 + (((c_) & 0xffu) << 8); \
        asm volatile("global_load_dword %0, %4, %8\n\t" \
                     "global_load_dword %1, %5, %8 offset:256\n\t" \
                     "global_load_dword %2, %6, %8\n\t" \
                     "global_load_dword %3, %7, %8" \
                     : "+&v"(w_), "+&v"(x_), "+&v"(e_), "+&v"(row_) \
                     : "v"(aw_), "v"(ax_), "v"(ae_), "v"(ar_), "s"(arena) : "memory"); \
    } while (0)
    /* Wait until all but the step's five stores are complete — that is every load of both sets — THEN copy
     * the chosen set into the current-step registers. The copy sits inside the statement: anything the
     * compiler places in front of it (it copies tied operands) could read registers a load is still going
     * to write. Both sets are plain inputs of both forms, so their registers stay allocated until every
     * load has landed; nothing is in flight into a register once this statement has run. */
#define CR_V3_TAKE(behind_, pick_s_) do { \
        const u64 m_ = cr_uni((pick_s_) ? 1u : 0u) ? ~0ull : 0ull; \
        asm volatile("s_waitcnt vmcnt(" #behind_ ")\n\t" \
                     "v_cndmask_b32_e64 %0, %4, %8, %12\n\tv_cndmask_b32_e64 %1, %5, %9, %12\n\t" \
                     "v_cndmask_b32_e64 %2, %6, %10, %12\n\tv_cndmask_b32_e64 %3, %7, %11, %12" \
                     : "=&v"(f_w), "=&v"(f_x), "=&v"(f_e), "=&v"(f_row) \
                     : "v"(n_w), "v"(n_x), "v"(n_e), "v"(n_row), "v"(s_w), "v"(s_x), "v"(s_e), "v"(s_row), "s"(m_) : "memory"); \
    } while (0)
    CR_V3_ISSUE(ctx, n_w, n_x, n_e, n_row);
    uint32_t f_w, f_x, f_e, f_row;                                       /* the current step's model, as loaded */
    CR_V3_TAKE(0, false);
    /* every vector-memory operation of this wave has completed (hidden loads must not land in registers
     * the compiler has given to somebody else; hidden stores must be readable) */
#define CR_V3_DRAIN() asm volatile("s_waitcnt vmcnt(0)" ::: "memory")

#ifdef CR_V3_PROF
    u64 pf_take = 0, pf_match = 0, pf_nmatch = 0, pf_steps = 0, pf_esc = 0;
#endif
    CR_V3_T(pf_t0);
    while (have < total) {                                               /* cr-coder.c:259-290 */
        cr_rc_pin(rc);
        in.pos = cr_uni(in.pos); in.base = cr_uni(in.base);
        ctx = cr_uni(ctx); nd_key = cr_uni(nd_key); nd_x = cr_uni(nd_x);
        o3_lk = cr_uni(o3_lk); o3_lv = cr_uni(o3_lv); lr_idx = cr_uni(lr_idx);
        /* ---- this step's model */
        const uint32_t fx = cr_uni(f_x);
        uint32_t e = cr_uni(f_e);                                        /* the aligned word holding the u16 entry */
        const uint32_t key = ctx & 0xffffu;
        const bool same = key == nd_key, live = (fx >> 16) == gen;       /* stale tag: node not yet used in this block (o2_model_init) */
        uint32_t w = same ? nd_w : (live ? f_w : 0u);
        uint32_t x = same ? nd_x : (live ? (fx & 0xffffu) : 0x0101u);
        const uint32_t w_was = w;                                        /* what memory holds (nothing valid for a node's first use) */
        const uint32_t k3 = cr_o3_key(ctx);
        e = (e >> ((k3 & 1u) << 4)) & 0xffffu;
        if (k3 == o3_lk) e = o3_lv;                                      /* loaded before the previous step's store */
        const bool e_live = ((e >> 4) & 15u) == g3;                      /* stale: the reference's zeroed entry */
        uint32_t pred = e_live ? (e >> 8) & 0xffu : 0u, conf = e_live ? e & 15u : 0u;
        const uint32_t row_idx = ctx & 0xffu;
        uint32_t row = (row_idx == lr_idx) ? lr_row : f_row;
        if (SPEC) {                                                      /* the context that follows if the prediction hits: its loads go out now */
            spec_ctx = (ctx << 8) | pred;
            CR_V3_ISSUE(spec_ctx, s_w, s_x, s_e, s_row);
        }

        /* ---- ppm_decode, cr-ppm.c:169-235 */
        const uint32_t f_hit = x & 0xffu, f_esc = (x >> 8) & 0xffu;
        uint32_t wx = w;
        if (lane == (pred >> 2)) wx &= ~(0xffu << ((pred & 3u) * 8u));
        const uint32_t mysum = cr_bytesum(wx);
        const uint32_t incl = cr_scan_incl(mysum);
        const uint32_t bytes = cr_lane_get(incl, 63);
        const uint32_t unit = cr_uni(rc.range / (bytes + f_hit + f_esc));           /* cr-rangecoder.c:101-104 */
        const uint32_t cache = rc.cache;
        const uint32_t tb = bytes * unit;
        uint32_t s, lower, frq;
        if (cache < tb) {
            /* first lane whose inclusive cumulative count, times unit, exceeds cache */
            const uint32_t ol = (uint32_t)__builtin_popcountll(cr_ballot(incl * unit <= cache));
            const uint32_t ww = cr_lane_get(wx, ol), upto = cr_lane_get(incl, ol);
            const uint32_t b0 = ww & 0xffu, b1 = (ww >> 8) & 0xffu, b2 = (ww >> 16) & 0xffu, b3 = ww >> 24;
            const uint32_t a0 = upto - (b0 + b1 + b2 + b3), a1 = a0 + b0, a2 = a1 + b1, a3 = a2 + b2;
            uint32_t j;
            if (cache < a1 * unit) { j = 0; lower = a0; frq = b0; }
            else if (cache < a2 * unit) { j = 1; lower = a1; frq = b1; }
            else if (cache < a3 * unit) { j = 2; lower = a2; frq = b2; }
            else { j = 3; lower = a3; frq = b3; }
            s = ol * 4u + j;
        } else if (cache < tb + f_hit * unit) {
            s = 256u; lower = bytes; frq = f_hit;
        } else {
            s = 257u; lower = bytes + f_hit; frq = f_esc;
        }
        rc.range = unit;
        cr_rc_dec_consume(rc, lower, frq, in);
        uint32_t sym = s == 256u ? pred : s;
        uint32_t halved = 0;
        uint8_t* row_dst = scratch;
        if (s == 257u) {                                                 /* cr-ppm.c:209-232 */
            const uint32_t ne = (f_esc + 1u) & 0xffu;
            x = (x & 0x00ffu) | (ne << 8);
            if (ne > 250u) { cr_lean_halve(w, x); halved = 1; }
            uint32_t keep = cr_zero_bytes(w);
            if (lane == (pred >> 2)) keep &= ~(0xffu << ((pred & 3u) * 8u));
            const uint32_t mine = cr_o1_weight_sum(row, keep);
            const uint32_t incl1 = cr_scan_incl(mine);
            const uint32_t all = cr_lane_get(incl1, 63);
            const uint32_t unit1 = cr_uni(rc.range / all);
            const uint32_t cache1 = rc.cache;
            uint32_t got = 0, lo = 0, fo = 1;
            if (cache1 < all * unit1) {
                const uint32_t ol = (uint32_t)__builtin_popcountll(cr_ballot(incl1 * unit1 <= cache1));
                const uint32_t rw = cr_lane_get(row, ol), kp = cr_lane_get(keep, ol), upto = cr_lane_get(incl1, ol);
                const uint32_t q0 = (kp & 0x000000ffu) ? ((rw & 0xffu) * 8u - 7u) : 0u;
                const uint32_t q1 = (kp & 0x0000ff00u) ? (((rw >> 8) & 0xffu) * 8u - 7u) : 0u;
                const uint32_t q2 = (kp & 0x00ff0000u) ? (((rw >> 16) & 0xffu) * 8u - 7u) : 0u;
                const uint32_t q3 = (kp & 0xff000000u) ? ((rw >> 24) * 8u - 7u) : 0u;
                const uint32_t a0 = upto - (q0 + q1 + q2 + q3), a1 = a0 + q0, a2 = a1 + q1, a3 = a2 + q2;
                uint32_t j;
                if (cache1 < a1 * unit1) { j = 0; lo = a0; }
                else if (cache1 < a2 * unit1) { j = 1; lo = a1; }
                else if (cache1 < a3 * unit1) { j = 2; lo = a2; }
                else { j = 3; lo = a3; }
                got = ol * 4u + j;
                fo = ((rw >> (8u * j)) & 0xffu) * 8u - 7u;
            }
            rc.range = unit1;
            cr_rc_dec_consume(rc, lo, fo, in);
            sym = got;
            /* ppm_update_o1, cr-ppm.c:90-97 */
            const uint32_t cur = cr_table_byte(row, sym);
            if (lane == (sym >> 2)) row += 1u << ((sym & 3u) * 8u);
            if (cur + 1u >= 255u) row -= (row >> 1) & 0x7f7f7f7fu;
            row_dst = o1 + (row_idx << 8);
            lr_idx = row_idx; lr_row = row;
        }
        sym = cr_uni(sym);

        /* ---- what the symbol means (cr-coder.c:261-289), before the next context's loads go out */
        uint32_t newctx = (ctx << 8) | sym;
        uint8_t* lit_dst = scratch + 512u;
        uint32_t lit = 0;
        if (!after_esc) {
            if (sym == esc) after_esc = 1;
            else { lit_dst = dst + have; lit = sym; }
        } else {
            after_esc = 0;
            if (sym == 0u) { lit_dst = dst + have; lit = esc; newctx = (ctx << 8) | esc; }
            else {
                const uint32_t len = sym;
                if (have + len > total || have + len > cap) { CR_V3_DRAIN(); return 0xFFFFFFFFu; }  /* corrupt stream */
                CR_V3_T(pf_m0);
                CR_V3_DRAIN();                                           /* the literals' stores are readable */
                cr_wave_sync();
                uint32_t c8, c4, c2;
                cr_lzp_learn_predict(z, pend_x, learned, have - learned, x8, c8, c4, c2);
                learned = have;
                const uint32_t p8 = have - c8, p4 = have - c4, p2 = have - c2;
                const uint32_t r8 = (len > p8) ? lane % p8 : lane, r4 = (len > p4) ? lane % p4 : lane, r2 = (len > p2) ? lane % p2 : lane;
                const u64 v8 = *reinterpret_cast<const cr_u64u*>(dst + c8 - 8);
                const uint32_t v4 = *reinterpret_cast<const cr_u32u*>(dst + c4 - 4);
                uint32_t s8 = 0, s4 = 0, s2 = 0;
                if (lane < len) { s8 = dst[c8 + r8]; s4 = dst[c4 + r4]; s2 = dst[c2 + r2]; }
                uint32_t from = c2, mine = s2;
                if (v8 == x8) { from = c8; mine = s8; }
                else if (v4 == (uint32_t)(x8 >> 32)) { from = c4; mine = s4; }
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
                if (len >= 4u && ((len - 1u) & 63u) >= 3u) {
                    uint32_t l3 = (len - 1u) & 63u;
                    newctx = (cr_lane_get(mine, l3 - 3u) << 24) | (cr_lane_get(mine, l3 - 2u) << 16) |
                             (cr_lane_get(mine, l3 - 1u) << 8) | cr_lane_get(mine, l3);
                } else {
                    cr_wave_sync();
                    newctx = ctx;
                    uint32_t k = len < 4u ? len : 4u;
                    for (uint32_t i = len - k; i < len; i++) newctx = (newctx << 8) | cr_uni(dst[have + i]);
                }
                if (len < CRGPU_WAVE) {
                    uint32_t t = mine & 0xffu;
                    u64 xa = (u64)t << 56;
#pragma unroll
                    for (uint32_t k = 1; k < 8u; k++) {
                        t = cr_shift_up1(t, (uint32_t)(x8 >> (8u * (8u - k))) & 0xffu);
                        xa |= (u64)t << (8u * (7u - k));
                    }
                    const uint32_t lo = cr_shift_up1((uint32_t)xa, (uint32_t)x8), hi = cr_shift_up1((uint32_t)(xa >> 32), (uint32_t)(x8 >> 32));
                    pend_x = ((u64)hi << 32) | lo;
                    x8 = cr_lane_get64(xa, len - 1u);
                    have += len;
                } else {
                    cr_wave_sync();
                    for (uint32_t q0 = have; q0 < have + len; q0 += CRGPU_WAVE) {
                        uint32_t q = q0 + lane;
                        if (q < have + len) cr_lzp_learn(z, *reinterpret_cast<const cr_u64u*>(dst + q - 8), q);
                    }
                    have += len;
                    learned = have;
                    x8 = *reinterpret_cast<const cr_u64u*>(dst + have - 8);
                    x8 = ((u64)cr_uni((uint32_t)(x8 >> 32)) << 32) | cr_uni((uint32_t)x8);
                }
                cr_wave_sync();
                CR_V3_DRAIN();
#ifdef CR_V3_PROF
                CR_V3_ACC(pf_match, pf_m0); pf_nmatch++;
#endif
            }
        }
        const uint32_t have_at = have;
        if (lit_dst != scratch + 512u) {                                 /* a literal byte at `have` (register bookkeeping only) */
            if (lane == have - learned) pend_x = x8;
            x8 = (x8 >> 8) | ((u64)lit << 56);
            have++;
            if (have - learned == CRGPU_WAVE) { cr_lzp_learn(z, pend_x, learned + lane); learned = have; }
        }

        /* ---- next step's loads */
        newctx = cr_uni(newctx);
        const bool use_s = SPEC && newctx == spec_ctx;                   /* the speculative set already is that context */
        if (!use_s) CR_V3_ISSUE(newctx, n_w, n_x, n_e, n_row);

        /* ---- model updates (cr-ppm.c:199-232), in registers */
        if (s == 256u) {
            const uint32_t hv = ((x & 0xffu) + 1u) & 0xffu;
            x = (x & 0xff00u) | hv;
            if (hv > 250u) cr_lean_halve(w, x);
            conf += (conf < 15u) ? 1u : 0u;
        } else {
            if (s < 256u) {
                if (lane == (s >> 2)) w += 1u << ((s & 3u) * 8u);
                if (frq + 1u > 250u) cr_lean_halve(w, x);
                else if (frq + 1u == 2u) {
                    const uint32_t ne = (((x >> 8) & 0xffu) - 1u) & 0xffu;
                    x = (x & 0x00ffu) | (ne << 8);
                    if (ne > 250u) cr_lean_halve(w, x);
                }
            } else if (!halved) {
                if (lane == (sym >> 2)) w += 1u << ((sym & 3u) * 8u);
            }
            uint32_t c = (uint32_t)(conf > 1u) + (uint32_t)(conf > 2u) + (uint32_t)(conf > 4u) + (uint32_t)(conf > 8u);
            if (c == 0u) { pred = (s < 256u) ? s : sym; c = 1u; }
            conf = c;
        }
        /* ---- the step's five stores, each by every lane (a lane whose word did not change, and every
         * lane of a step without an order-1 update, aims at a scratch word instead) */
        {
            const uint32_t no = key * CRGPU_NODE_BYTES;
            const uint32_t a_w = (w != w_was || !(same || live)) ? vo_nodes + no : so_scr + 256u;
            const uint32_t a_x = so_nodes + no, v_x = x | (gen << 16);
            const uint32_t val3 = (pred << 8) | (g3 << 4) | conf;
            const uint32_t a_e = so_o3d + (k3 << 1);
            const uint32_t a_r = (row_dst == scratch) ? so_scr : vo_o1 + (row_idx << 8);
            const bool is_lit = lit_dst != scratch + 512u;
            const uint8_t* const l_base = is_lit ? dst : arena;
            const uint32_t a_l = is_lit ? have_at : so_scr + 512u;
            asm volatile("global_store_dword %0, %1, %9\n\t"
                         "global_store_dword %2, %3, %9 offset:256\n\t"
                         "global_store_short %4, %5, %9\n\t"
                         "global_store_dword %6, %7, %9\n\t"
                         "global_store_byte %8, %10, %11"
                         :: "v"(a_w), "v"(w), "v"(a_x), "v"(v_x), "v"(a_e), "v"(val3), "v"(a_r), "v"(row), "v"(a_l),
                            "s"(arena), "v"(lit), "s"(l_base) : "memory");
            nd_key = key; nd_w = w; nd_x = x;
            o3_lk = k3; o3_lv = val3;
        }
        /* ---- the model of the next step: the loads are older than the five stores */
        CR_V3_T(pf_w0);
        CR_V3_TAKE(5, use_s);
#ifdef CR_V3_PROF
        CR_V3_ACC(pf_take, pf_w0); pf_steps++; pf_esc += (s == 257u);
#endif
        ctx = newctx;
    }
    CR_V3_DRAIN();
#ifdef CR_V3_PROF
    if (st && lane == 0) { st[8] = __builtin_amdgcn_s_memtime() - pf_t0; st[9] = pf_take; st[10] = pf_match; st[11] = pf_nmatch; st[12] = pf_steps; st[13] = pf_esc; }
#endif
#undef CR_V3_ISSUE
#undef CR_V3_TAKE
#undef CR_V3_DRAIN
    cr_stamp(st, 5);
    return have;
}

#endif

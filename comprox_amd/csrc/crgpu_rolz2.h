/*
 * comprox_amd/csrc/crgpu_rolz2.h — comprolz lzencode for the batched API on the comprop encoder's kernel pipeline.
 *
 * Reference: /root/reference/src/rolzmain/cr-coder.c:150-250 (lzencode).
 *
 * The parse is already there (k_rolz_match left a rank / length for every position), so which positions start a
 * token only depends on the lengths, exactly as in comprop's k_rop_events: 64 positions per step, the few matches
 * of a step resolved with scalar bit operations, one event {four bytes in front, symbol} per token. The single side
 * stream (match length, ring rank; a zero length for a literal escape byte) is coded in token order by the same
 * wave. k_rop_links / _o3 / _o2 / _o1 produce the triples, k_rolz_rc codes them and assembles the block.
 */
#ifndef CRGPU_ROLZ2_H
#define CRGPU_ROLZ2_H

#include "crgpu_rolz.h"
#include "crgpu_rop2.h"

#define CR_ROLZC_CODES 4     /* V.ctr words: number of side-stream codes, bytes of the side stream */
#define CR_ROLZC_BSIDE 5

CR_DEV void cr_rolz_emit_events(const uint8_t* src, uint32_t n, const CrRolzTables& T, uint8_t* side, CrEvViews& V, CrRoxShared& sh) {
    const uint32_t lane = cr_lane();
    if (n == 0) {                                             /* (the one-wave coder's empty stored block) */
        if (lane == 0) { V.ctr[0] = 0; V.ctr[1] = 0; V.ctr[2] = 0; V.ctr[3] = 0x400u; }
        return;
    }
    const uint32_t esc = cr_pick_escape(src, n, sh.hist);
    cr_rolz_side_reset(sh);
    cr_wave_sync();
    CrSink s_side; s_side.dst = side; s_side.n = 0;
    CrRc rc_side; cr_rc_init(rc_side);
    const uint32_t head4 = n >= 4u ? __builtin_bswap32(*reinterpret_cast<const cr_u32u*>(src))
                                   : ((uint32_t)src[0] << 24) | ((n > 1u ? (uint32_t)src[1] : 0u) << 16) | ((n > 2u ? (uint32_t)src[2] : 0u) << 8);
    uint32_t nev = 0, skip_until = 1, codes = 0;              /* the first byte travels in the header */
    /* A step's operands are fetched two steps ahead into two register slots (round 4: unconditional loads with a clamped
     * index — behind `if (p < n)` they are branches the compiler cannot count, and a step began with a memory round trip —, a
     * slot reloaded as soon as its values are taken out, the loop unrolled by the two slots so that nothing is copied). */
    struct Slot { uint32_t c, ctx, rank, len; };
    const auto fetch = [&](Slot& sl, uint32_t base) __attribute__((always_inline)) {
        const uint32_t p = base + lane, q = p < n ? p : n - 1u;
        sl.c = src[q];
        /* (a block of 1 .. 3 bytes has no such word — head4 covers p < 5 —: its lanes read the match kernel's scratch instead of
         * running up to 3 bytes past a caller's allocation; a select on the address, the load stays unconditional) */
        const uint8_t* const cp = n >= 4u ? src + (q >= 4u ? q - 4u : 0u) : reinterpret_cast<const uint8_t*>(T.len);
        sl.ctx = *reinterpret_cast<const cr_u32u*>(cp);
        sl.rank = T.rank[q];
        sl.len = T.len[q];
    };
    const auto step = [&](Slot& sl, uint32_t base) __attribute__((always_inline)) {
        const uint32_t p = base + lane;
        uint32_t c = 0, ctx = 0, rank = 0xffu, len = 1;
        if (p < n) {
            c = sl.c;
            /* the context starts empty at position 1 (cr-coder.c:188): below position 5 it holds src[1 .. p-1] only */
            if (p >= 5u) ctx = __builtin_bswap32(sl.ctx);
            else ctx = (head4 >> (8u * (4u - p))) & ((1u << (8u * (p - 1u))) - 1u);
            if (p >= CR_ROLZ_WARM && p + CR_ROLZ_TAIL < n) { rank = sl.rank; len = sl.len; }
        }
        fetch(sl, base + 2u * CRGPU_WAVE);
        const bool is_match = rank != 0xffu;
        const u64 mm = cr_ballot(is_match);
        u64 starts = 0;
        uint32_t cur = skip_until > base ? skip_until - base : 0u;
        while (cur < 64u) {
            const u64 rest = mm >> cur << cur;
            if (!rest) { starts |= ~0ull << cur; cur = 64u; break; }
            const uint32_t l = (uint32_t)__builtin_ctzll(rest);
            starts |= (~0ull << cur) & (l == 63u ? ~0ull : ((2ull << l) - 1ull));
            cur = l + cr_lane_get(len, l);
        }
        skip_until = base + cur;
        if (n - base < 64u) starts &= (1ull << (n - base)) - 1ull;
        const bool start = (starts >> lane) & 1ull;
        const bool coded_match = start && is_match;
        const uint32_t incl = cr_scan_incl(start ? 1u : 0u);
        if (start) {
            const uint32_t e = nev + incl - 1u;
            V.ev_ctx[e] = ctx;
            V.ev_sym[e] = (uint16_t)((coded_match ? esc : c) | CR_EV_LAST);
        }
        nev += cr_lane_get(incl, 63);
        /* the side stream, in token order (cr-coder.c:205-229) */
        for (u64 todo = cr_ballot(coded_match || (start && c == esc)); todo; todo &= todo - 1ull) {
            const uint32_t l = (uint32_t)__builtin_ctzll(todo);
            if ((mm >> l) & 1ull) {
                cr_side_encode(sh, CR_ROLZ_M_LEN, cr_lane_get(len, l), 4u, rc_side, s_side);
                cr_side_encode(sh, CR_ROLZ_M_IDX, cr_lane_get(rank, l), 4u, rc_side, s_side);
            } else {
                cr_side_encode(sh, CR_ROLZ_M_LEN, 0u, 4u, rc_side, s_side);
            }
            codes++;
        }
    };
    Slot sa, sb;
    fetch(sa, 1u); fetch(sb, 1u + CRGPU_WAVE);
    for (uint32_t base = 1; base < n; base += 2u * CRGPU_WAVE) {
        step(sa, base);
        if (base + CRGPU_WAVE < n) step(sb, base + CRGPU_WAVE);
    }
    cr_rc_pin(rc_side); cr_rc_flush(rc_side, s_side);
    if (lane == 0) {
        V.ctr[0] = nev; V.ctr[1] = 0; V.ctr[2] = 0; V.ctr[3] = esc;
        V.ctr[CR_ROLZC_CODES] = codes; V.ctr[CR_ROLZC_BSIDE] = s_side.n;
    }
}

/* main stream from the triples, then the block (cr-coder.c:233-250) */
CR_DEV uint32_t cr_rolz_finish(const uint8_t* src, uint32_t n, uint8_t* dst, const uint8_t* side, CrEvViews& V, u64* ring) {
    const uint32_t lane = cr_lane();
    const uint32_t info = cr_uni(V.ctr[3]);
    if (info & 0x400u) { if (lane < CR_ROLZ_HEADER) dst[lane] = 0; return CR_ROLZ_HEADER; }
    const uint32_t esc = info & 0xffu;
    uint32_t got = cr_code_events_fast(n, dst + CR_ROLZ_HEADER, CR_ROLZ_HEADER, V, ring);
    if (got == 0u) got = cr_code_events(n, dst + CR_ROLZ_HEADER, CR_ROLZ_HEADER, V);
    if (got == 0xFFFFFFFFu) {
        cr_wave_sync();
        cr_rolz_store_raw(src, n, dst);
        return CR_ROLZ_HEADER + n;
    }
    cr_wave_sync();
    const uint32_t codes = cr_uni(V.ctr[CR_ROLZC_CODES]), b_side = cr_uni(V.ctr[CR_ROLZC_BSIDE]);
    const uint32_t o_side = CR_ROLZ_HEADER + got;
    if ((u64)o_side + b_side > cr_bound_rolz(n)) return 0xFFFFFFFFu;         /* cannot happen (crgpu_device.h); never write past the slot */
    for (uint32_t i = lane; i < b_side; i += CRGPU_WAVE) dst[o_side + i] = side[i];
    if (lane < CR_ROLZ_HEADER) {                                         /* cr-coder.c:241-245 */
        const uint32_t fields[4] = {(uint32_t)src[0] | (1u << 8) | (esc << 16), n, codes, o_side};
        dst[lane] = (uint8_t)(fields[lane >> 2] >> (8u * (lane & 3u)));
    }
    return o_side + b_side;
}

#endif

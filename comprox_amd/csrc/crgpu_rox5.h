/*
 * comprox_amd/csrc/crgpu_rox5.h — comprox lzdecode for the batched API with the PPM main stream in assembly.
 *
 * Reference: /root/reference/src/roxmain/cr-coder.c:390-526 (lzdecode), :347-368 (distance symbols).
 *
 * The main stream is a plain sequence of ppm_decode symbols: a byte other than the block's escape byte is a
 * literal (stored, pushed into the context), the escape byte announces a match whose length / distance come from
 * the three side streams. crgpu_rop5.h's statement runs the literals in its mode 1 and hands over at every
 * escape byte; the side streams (small adaptive models in LDS, crgpu_rox.h) and the copy stay in C++.
 * Model tables: the decoder's direct-indexed layout of crgpu_dec.h / crgpu_rop5.h (same generation counters
 * as the one-wave coder, so both can use one arena).
 */
#ifndef CRGPU_ROX5_H
#define CRGPU_ROX5_H

#include "crgpu_rop5.h"
#include "crgpu_rox.h"

CR_DEV uint32_t cr_rox_decode_v5(const uint8_t* src_, uint32_t n, uint8_t* dst_, uint32_t cap, uint8_t* arena_,
                                 const CrArenaLayout& L, CrRoxShared& sh, uint32_t lds_scratch) {
    const uint8_t* const src = cr_uni_ptr(src_);
    uint8_t* const dst = cr_uni_ptr(dst_);
    uint8_t* const arena = cr_uni_ptr(arena_);
    n = cr_uni(n); cap = cr_uni(cap);
    const uint32_t lane = cr_lane();
    if (n < CR_ROX_HEADER) return 0xFFFFFFFFu;
    if (src[0] == 0) {
        uint32_t raw = n - CR_ROX_HEADER;
        if (raw > cap) return 0xFFFFFFFFu;
        for (uint32_t i = lane; i < raw; i += CRGPU_WAVE) dst[i] = src[CR_ROX_HEADER + i];
        return raw;
    }
    const uint32_t long_min = cr_uni(src[1]), esc = cr_uni(src[2]);
    uint32_t hw[7];
    for (int k = 0; k < 7; k++) hw[k] = cr_uni((uint32_t)src[4 + 4 * k] | ((uint32_t)src[5 + 4 * k] << 8) | ((uint32_t)src[6 + 4 * k] << 16) | ((uint32_t)src[7 + 4 * k] << 24));
    const uint32_t total = hw[0], o_spos = hw[4], o_pos = hw[5], o_len = hw[6];
    if (total > cap || total > L.max_block || o_spos < CR_ROX_HEADER || o_spos > o_pos || o_pos > o_len || o_len > n) return 0xFFFFFFFFu;

    cr_side_reset(sh);
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

    CrSource in_spos, in_pos, in_len;
    cr_source_init(in_spos, src + o_spos, n - o_spos);
    cr_source_init(in_pos, src + o_pos, n - o_pos);
    cr_source_init(in_len, src + o_len, n - o_len);
    CrRc rc_spos, rc_pos, rc_len;
    cr_rc_dec_init(rc_spos, in_spos); cr_rc_dec_init(rc_pos, in_pos); cr_rc_dec_init(rc_len, in_len);

    /* main stream: the coder state of crgpu_rop5.h (range_decoder_init reads five bytes, the first drops out) */
    const uint8_t* const payload = src + CR_ROX_HEADER;
    const uint32_t psize = n - CR_ROX_HEADER;
    uint32_t wbase = 0, win = cr_v4_window(payload, psize, 0u);
    uint32_t cache = cr_lane_get(win, 0), range = 0xFFFFFFFFu;
    uint32_t ib_hi = cr_lane_get(win, 1), ib_lo = cr_lane_get(win, 2), ibits = 64, widx = 3;
    uint32_t ctx = 0, have = 0, learned = 0, after_esc = 0, x8_lo = 0, x8_hi = 0, pend_lo = 0, pend_hi = 0;
    const uint32_t zero = 0;
    uint32_t prev_dist = 0;

    const uint32_t dslots = cr_uni(L.dense_slots);
    while (have < total) {                                               /* cr-coder.c:459-523 */
        uint32_t ev, sym, pacc, pcnt;
        /* (the side-stream code below leaves the compiler unsure that these are wave-uniform) */
        ctx = cr_uni(ctx); range = cr_uni(range); cache = cr_uni(cache); ib_lo = cr_uni(ib_lo); ib_hi = cr_uni(ib_hi); ibits = cr_uni(ibits);
        widx = cr_uni(widx); have = cr_uni(have); learned = cr_uni(learned); after_esc = cr_uni(after_esc); x8_lo = cr_uni(x8_lo); x8_hi = cr_uni(x8_hi);
        asm volatile(CR_V5_SIDE_MODE(1) CR_V5_ASM_DEFS CR_V5_ASM_MACROS CR_V5_ASM_BODY
                     : [ctx] "+s"(ctx), [range] "+s"(range), [cache] "+s"(cache), [iblo] "+s"(ib_lo), [ibhi] "+s"(ib_hi),
                       [ibits] "+s"(ibits), [widx] "+s"(widx), [have] "+s"(have), [learned] "+s"(learned), [aesc] "+s"(after_esc),
                       [x8lo] "+s"(x8_lo), [x8hi] "+s"(x8_hi), [ev] "=&s"(ev), [sym] "=&s"(sym), [plo] "+v"(pend_lo), [phi] "+v"(pend_hi),
                       [pacc] "=&v"(pacc), [pcnt] "=&v"(pcnt)
                     : [win] "v"(win), [arena] "s"(arena), [dst] "s"(dst), [total] "s"(total), [gen] "s"(gen), [g3] "s"(g3), [esc] "s"(esc),
                       [cap] "s"(cap), [off8] "s"(zero), [off4] "s"(zero), [off2] "s"(zero), [lzsh] "s"(zero), [dslots] "s"(dslots)
                     : CR_V5_CLOBBERS);
        ev = cr_uni(ev);
        (void)sym; (void)pacc; (void)pcnt;
        if (ev == CR_V5_EV_DONE) break;
        if (ev == CR_V5_EV_WINDOW) {
            wbase += widx * 4u;
            win = cr_v4_window(payload, psize, wbase);
            widx = 0;
            continue;
        }
        if (ev != CR_V5_EV_ESC) return 0xFFFFFFFFu;
        /* the escape byte: length, then distance (cr-coder.c:462-500) */
        uint32_t len = cr_uni(cr_side_decode(sh, CR_SIDE_LEN, 30u, rc_len, in_len)), dist = 0;
        if (len == 0u) {                                                 /* the escape byte itself */
            if (have >= cap) return 0xFFFFFFFFu;
            if (lane == 0) dst[have] = (uint8_t)esc;
            ctx = cr_uni((ctx << 8) | esc);
            have = cr_uni(have + 1u);
            continue;
        }
        if (len < long_min) {
            dist = cr_uni(cr_side_decode(sh, CR_SIDE_SPOS, 1u, rc_spos, in_spos));
        } else {                                                         /* cr-coder.c:347-368 */
            uint32_t v = 0, s2 = 0, j = 0;
            while (j < 2u && (s2 = cr_uni(cr_side_decode(sh, CR_SIDE_POS + j, 1u << (2u * j), rc_pos, in_pos))) >= 128u) { v += (s2 - 128u) << (7u * j); j++; }
            if (j < 2u) {
                dist = (v + (s2 << (7u * j))) / 8u;
            } else {
                while (j < 5u && (s2 = cr_uni(cr_side_decode(sh, CR_SIDE_POS + j, 1u << (2u * j), rc_pos, in_pos))) >= 64u) { v += (s2 - 64u) << (6u * j + 2u); j++; }
                dist = (v + (s2 << (6u * j + 2u))) / 8u;
            }
        }
        if (len > 1u) {                                                  /* cr-coder.c:503-514 */
            const uint32_t dd = cr_uni(dist > 0u ? dist : prev_dist);
            if (dd == 0u || dd > have || have + len > total || have + len > cap) return 0xFFFFFFFFu;
            prev_dist = dd;
            const uint32_t from = have - dd;
            uint32_t mine = 0;
            for (uint32_t i0 = 0; i0 < len; i0 += CRGPU_WAVE) {
                uint32_t i = i0 + lane;
                if (i < len) {
                    uint32_t r = i < dd ? i : i % dd;
                    mine = dst[from + r];
                    dst[have + i] = (uint8_t)mine;
                }
            }
            if (len >= 4u && ((len - 1u) & 63u) >= 3u) {
                uint32_t l3 = (len - 1u) & 63u;
                ctx = (cr_lane_get(mine, l3 - 3u) << 24) | (cr_lane_get(mine, l3 - 2u) << 16) | (cr_lane_get(mine, l3 - 1u) << 8) | cr_lane_get(mine, l3);
            } else {
                cr_wave_sync();
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                uint32_t k = len < 4u ? len : 4u;
                for (uint32_t i = len - k; i < len; i++) ctx = (ctx << 8) | cr_uni(dst[have + i]);
            }
            ctx = cr_uni(ctx);
            have = cr_uni(have + len);
        } else {                                                         /* a coded length of 1 (no encoder writes it): the escape byte as a literal */
            if (have >= cap) return 0xFFFFFFFFu;
            if (lane == 0) dst[have] = (uint8_t)esc;
            ctx = cr_uni((ctx << 8) | esc);
            have = cr_uni(have + 1u);
        }
    }
    return have;
}

#endif

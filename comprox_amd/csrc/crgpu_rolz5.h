/*
 * comprox_amd/csrc/crgpu_rolz5.h — comprolz lzdecode for the batched API with the PPM main stream in assembly.
 *
 * Reference: /root/reference/src/rolzmain/cr-coder.c:293-378 (lzdecode), :265-277 (length / rank symbols),
 * src/rolzmain/cr-matcher.c:66-84 (matcher_update), :126-146 (matcher_getpos).
 *
 * Literals run in crgpu_rop5.h's statement (mode 2: every stored byte becomes a pending matcher position, lane j of
 * the pending registers holding the 8 bytes in front of position fed + j); the statement ends at the escape byte
 * and when 64 positions are pending. Feeding the rings / rows, the rank lookup and the copy are crgpu_rolz.h's.
 */
#ifndef CRGPU_ROLZ5_H
#define CRGPU_ROLZ5_H

#include "crgpu_rop5.h"
#include "crgpu_rolz.h"

CR_DEV uint32_t cr_rolz_decode_v5(const uint8_t* src_, uint32_t n, uint8_t* dst_, uint32_t cap, const CrRolzTables& T, uint32_t* row_head,
                                  uint8_t* arena_, const CrArenaLayout& L, CrRoxShared& sh, uint32_t* hist, uint32_t lds_scratch, u64* st) {
    const uint8_t* const src = cr_uni_ptr(src_);
    uint8_t* const dst = cr_uni_ptr(dst_);
    uint8_t* const arena = cr_uni_ptr(arena_);
    n = cr_uni(n); cap = cr_uni(cap);
    const uint32_t lane = cr_lane();
    if (n < CR_ROLZ_HEADER) return 0xFFFFFFFFu;
    if (src[1] == 0) {                                                   /* cr-coder.c:303-308 */
        uint32_t raw = n - CR_ROLZ_HEADER;
        if (raw > cap) return 0xFFFFFFFFu;
        for (uint32_t i = lane; i < raw; i += CRGPU_WAVE) dst[i] = src[CR_ROLZ_HEADER + i];
        return raw;
    }
    const uint32_t esc = cr_uni(src[2]);
    uint32_t hw[3];
    for (int k = 0; k < 3; k++) hw[k] = cr_uni((uint32_t)src[4 + 4 * k] | ((uint32_t)src[5 + 4 * k] << 8) | ((uint32_t)src[6 + 4 * k] << 16) | ((uint32_t)src[7 + 4 * k] << 24));
    const uint32_t total = hw[0], o_side = hw[2];
    uint32_t codes = hw[1];
    if (total == 0 || total > cap || total > L.max_block || o_side < CR_ROLZ_HEADER || o_side > n) return 0xFFFFFFFFu;
    const bool ctx4 = total >= 4194304u;                                 /* using_ctx4, cr-coder.c:314 */

    cr_rolz_side_reset(sh);
    uint32_t g3_;
    const uint32_t gen = cr_uni(cr_v3_reset(arena, L, g3_));
    const uint32_t g3 = cr_uni(g3_);
    cr_fill(reinterpret_cast<uint8_t*>(T.ring_head), (u64)CR_ROLZ_BUCKETS * 4u, 0u);     /* matcher_init */
    for (uint32_t k = lane; k < 256u; k += CRGPU_WAVE) row_head[k] = 0;
    if (lane == 0) dst[0] = src[0];
    /* where the dense slots start, the LDS address of the wave's 256 scratch bytes, the next free dense slot: the statement
     * reads them from the arena's scratch line (and writes the slot counter back there when it is left) */
    if (lane == 0) {
        uint32_t* scr = reinterpret_cast<uint32_t*>(arena + CRGPU_OFF_SCRATCH + 896u);
        scr[0] = (uint32_t)L.off_dense; scr[1] = lds_scratch; scr[2] = 0u;
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
    cr_wave_sync();
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");

    CrSource in_side;
    cr_source_init(in_side, src + o_side, n - o_side);
    CrRc rc_side;
    cr_rc_dec_init(rc_side, in_side);

    const uint8_t* const payload = src + CR_ROLZ_HEADER;
    const uint32_t psize = n - CR_ROLZ_HEADER;
    uint32_t wbase = 0, win = cr_v4_window(payload, psize, 0u);
    uint32_t cache = cr_lane_get(win, 0), range = 0xFFFFFFFFu;
    uint32_t ib_hi = cr_lane_get(win, 1), ib_lo = cr_lane_get(win, 2), ibits = 64, widx = 3;
    uint32_t ctx = 0, have = 1, fed = 1, after_esc = 0;                  /* positions < fed are in the tables (or below 16: never fed) */
    uint32_t x8_lo = 0, x8_hi = cr_uni((uint32_t)src[0]) << 24;          /* the 8 bytes in front of the write position */
    uint32_t pend_lo = 0, pend_hi = 0;                                   /* lane j: those 8 bytes for position fed + j */
    const uint32_t zero = 0;
    /* ring links as plain links (hist == nullptr: block sizes above 1 MiB) or with a history per position */
#define CR_FEED(x_, q0_, np_) do { if (hist) cr_rolz_feed_hist(T, hist, row_head, x_, q0_, np_, ctx4); else cr_rolz_feed(T, row_head, x_, q0_, np_, ctx4); } while (0)
#ifdef CR_ROLZ5_PROF
    u64 pf_asm = 0, pf_feed = 0, pf_get = 0, pf_side = 0, pf_n = 0, pf_rank = 0;
    const u64 pf_t0 = __builtin_amdgcn_s_memtime();
#define CR_PF(acc_, ...) do { const u64 t_ = __builtin_amdgcn_s_memtime(); __VA_ARGS__; acc_ += __builtin_amdgcn_s_memtime() - t_; } while (0)
#else
#define CR_PF(acc_, ...) do { __VA_ARGS__; } while (0)
#endif

    const uint32_t dslots = cr_uni(L.dense_slots);
    while (have < total) {                                               /* cr-coder.c:334-375 */
        uint32_t ev, sym, pacc, pcnt;
        ctx = cr_uni(ctx); range = cr_uni(range); cache = cr_uni(cache); ib_lo = cr_uni(ib_lo); ib_hi = cr_uni(ib_hi); ibits = cr_uni(ibits);
        widx = cr_uni(widx); have = cr_uni(have); fed = cr_uni(fed); after_esc = cr_uni(after_esc); x8_lo = cr_uni(x8_lo); x8_hi = cr_uni(x8_hi);
        CR_PF(pf_asm,
        asm volatile(CR_V5_SIDE_MODE(2) CR_V5_ASM_DEFS CR_V5_ASM_MACROS CR_V5_ASM_BODY
                     : [ctx] "+s"(ctx), [range] "+s"(range), [cache] "+s"(cache), [iblo] "+s"(ib_lo), [ibhi] "+s"(ib_hi),
                       [ibits] "+s"(ibits), [widx] "+s"(widx), [have] "+s"(have), [learned] "+s"(fed), [aesc] "+s"(after_esc),
                       [x8lo] "+s"(x8_lo), [x8hi] "+s"(x8_hi), [ev] "=&s"(ev), [sym] "=&s"(sym), [plo] "+v"(pend_lo), [phi] "+v"(pend_hi),
                       [pacc] "=&v"(pacc), [pcnt] "=&v"(pcnt)
                     : [win] "v"(win), [arena] "s"(arena), [dst] "s"(dst), [total] "s"(total), [gen] "s"(gen), [g3] "s"(g3), [esc] "s"(esc),
                       [cap] "s"(cap), [off8] "s"(zero), [off4] "s"(zero), [off2] "s"(zero), [lzsh] "s"(zero), [dslots] "s"(dslots)
                     : CR_V5_CLOBBERS));
        ev = cr_uni(ev);
        (void)sym; (void)pacc; (void)pcnt;
        if (ev == CR_V5_EV_DONE) break;
        if (ev == CR_V5_EV_WINDOW) {
            wbase += widx * 4u;
            win = cr_v4_window(payload, psize, wbase);
            widx = 0;
            continue;
        }
        if (ev == CR_V5_EV_LEARN) {
            CR_PF(pf_feed, CR_FEED(((u64)pend_hi << 32) | pend_lo, fed, CRGPU_WAVE));
            fed = have;
            continue;
        }
        if (ev != CR_V5_EV_ESC) return 0xFFFFFFFFu;
        uint32_t len = 0, rank = 0;
        if (codes > 0u) {                                                /* cr-coder.c:265-277 */
            codes--;
            CR_PF(pf_side,
            len = cr_uni(cr_side_decode(sh, CR_ROLZ_M_LEN, 4u, rc_side, in_side));
            if (len > 0u) rank = cr_uni(cr_side_decode(sh, CR_ROLZ_M_IDX, 4u, rc_side, in_side)));
        }
        if (len == 0u) {                                                 /* the escape byte itself */
            if (have >= cap) return 0xFFFFFFFFu;
            if (lane == 0) dst[have] = (uint8_t)esc;
            if (have >= CR_ROLZ_WARM) { if (lane == have - fed) { pend_lo = x8_lo; pend_hi = x8_hi; } } else fed = have + 1u;
            x8_lo = cr_uni((x8_lo >> 8) | (x8_hi << 24));
            x8_hi = cr_uni((x8_hi >> 8) | (esc << 24));
            have = cr_uni(have + 1u);
            ctx = cr_uni((ctx << 8) | esc);
            if (have - fed == CRGPU_WAVE) { CR_FEED(((u64)pend_hi << 32) | pend_lo, fed, CRGPU_WAVE); fed = have; }
            continue;
        }
        if (have + len > total || have + len > cap || have < CR_ROLZ_WARM) return 0xFFFFFFFFu;   /* corrupt stream */
        if (have > fed) CR_PF(pf_feed, CR_FEED(((u64)pend_hi << 32) | pend_lo, fed, have - fed));
        fed = have;
        cr_wave_sync();                                                  /* the literals' stores and the links are readable */
        const u64 x8 = ((u64)x8_hi << 32) | x8_lo;
        uint32_t from = 0;
        CR_PF(pf_get, from = cr_uni(hist ? cr_rolz_getpos_hist(T, hist, row_head, rank, have, x8, ctx4) : cr_rolz_getpos(T, row_head, rank, have, x8, ctx4)));
#ifdef CR_ROLZ5_PROF
        pf_n++; pf_rank += rank < CR_ROLZ_RING ? rank : rank - CR_ROLZ_RING;
#endif
        if (from == CR_ROLZ_NONE || from >= have) return 0xFFFFFFFFu;
        const uint32_t period = have - from;
        uint32_t mine = 0;
        for (uint32_t i0 = 0; i0 < len; i0 += CRGPU_WAVE) {              /* byte-serial copy semantics, cr-coder.c:355-358 */
            uint32_t i = i0 + lane;
            if (i < len) {
                uint32_t r = i < period ? i : i % period;
                mine = dst[from + r];
                dst[have + i] = (uint8_t)mine;
            }
        }
        {   /* len >= 5: the last four pushes sit in the lanes that copied them */
            const uint32_t l3 = (len - 1u) & 63u;
            if (l3 >= 3u) {
                ctx = (cr_lane_get(mine, l3 - 3u) << 24) | (cr_lane_get(mine, l3 - 2u) << 16) | (cr_lane_get(mine, l3 - 1u) << 8) | cr_lane_get(mine, l3);
            } else {
                cr_wave_sync();
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                for (uint32_t i = len - 4u; i < len; i++) ctx = (ctx << 8) | cr_uni(dst[have + i]);
            }
            ctx = cr_uni(ctx);
        }
        if (len < CRGPU_WAVE) {
            /* the copied positions become pending: lane i held byte have+i; xa = the 8 bytes ending there */
            uint32_t t = mine & 0xffu;
            u64 xa = (u64)t << 56;
#pragma unroll
            for (uint32_t k = 1; k < 8u; k++) {
                t = cr_shift_up1(t, (uint32_t)(x8 >> (8u * (8u - k))) & 0xffu);
                xa |= (u64)t << (8u * (7u - k));
            }
            pend_lo = cr_shift_up1((uint32_t)xa, x8_lo);                 /* lane 0: position have, lane j: have+j */
            pend_hi = cr_shift_up1((uint32_t)(xa >> 32), x8_hi);
            const u64 nx = cr_lane_get64(xa, len - 1u);
            x8_lo = cr_uni((uint32_t)nx); x8_hi = cr_uni((uint32_t)(nx >> 32));
            have = cr_uni(have + len);
        } else {
            cr_wave_sync();
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            for (uint32_t q0 = have; q0 < have + len; q0 += CRGPU_WAVE) {
                const uint32_t q = q0 + lane, np = have + len - q0 < CRGPU_WAVE ? have + len - q0 : CRGPU_WAVE;
                u64 xq = 0;
                if (q < have + len) xq = *reinterpret_cast<const cr_u64u*>(dst + q - 8);
                CR_FEED(xq, q0, np);
            }
            have = cr_uni(have + len);
            fed = have;
            const u64 nx = *reinterpret_cast<const cr_u64u*>(dst + have - 8);
            x8_lo = cr_uni((uint32_t)nx); x8_hi = cr_uni((uint32_t)(nx >> 32));
        }
    }
#ifdef CR_ROLZ5_PROF
    if (st && lane == 0) { st[8] = __builtin_amdgcn_s_memtime() - pf_t0; st[9] = pf_asm; st[10] = pf_feed; st[11] = pf_get; st[12] = pf_side; st[13] = pf_n; st[14] = pf_rank; }
#endif
#undef CR_PF
#undef CR_FEED
    (void)st;
    return have;
}

#endif

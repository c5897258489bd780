/*
 * comprox_amd/csrc/crgpu_rop4.h — comprop lzdecode for the batched API, straight-line step.
 *
 * Reference: /root/reference/src/ropmain/cr-coder.c:231-292 (lzdecode), src/cr-ppm.c:169-235
 * (ppm_decode), src/cr-rangecoder.c:81-104 (range decoder), src/cr-o2model.c:54-71,93-113.
 *
 * One wavefront per datablock; a block is a chain of ~43 000 dependent ppm_decode steps, and in-kernel
 * clock stamps (tools/dec_profile.py) show where a step's ~2 500 clocks went in the previous layouts:
 * 12 % waiting for the next context's model, 20 % in match tokens, the rest executing ~300 instructions
 * at 4.1 clocks each (tools/issue_probe.hip: a lone wave issues one instruction per ~4 clocks whatever its
 * kind) plus ~25 clocks for every TAKEN branch. So this version is written for instruction count and
 * fall-through:
 *   - tables as in crgpu_rop3.h: order-2 nodes direct-indexed with a generation tag, order-3 predictor
 *     direct-indexed by the reference's 22-bit key (u16 entries, 4-bit generation), order-1 rows dense;
 *   - the four loads of the next step and the five stores of this one are issued from inline assembly
 *     with an exact wait ("all but the last five": the loads only); addresses are 32-bit offsets from the
 *     arena base;
 *   - one division per coding step: the symbol search compares cumulative-count x unit products with the
 *     coder's `cache` (cum <= cache / unit  <=>  cum * unit <= cache; products stay below 2^32);
 *   - everything that is rare is an out-of-line branch that is normally NOT taken (first use of a node, a
 *     context that comes straight back, count halving, refilling the input window, learning 64 pending
 *     LZP positions, match tokens); what is common is selects;
 *   - the coded bytes arrive through a 64-bit scalar shift register (refilled 4 bytes at a time from a
 *     256-byte register window), so renormalisation is one count-leading-zeros and shifts, no byte loop.
 */
#ifndef CRGPU_ROP4_H
#define CRGPU_ROP4_H

#include "crgpu_rop3.h"

#define CR_LIKELY(x)   __builtin_expect(!!(x), 1)
#define CR_UNLIKELY(x) __builtin_expect(!!(x), 0)

#ifdef CR_V4_PROF
#define CR_V4_T(var_) const u64 var_ = __builtin_amdgcn_s_memtime()
#define CR_V4_ACC(acc_, from_) do { acc_ += __builtin_amdgcn_s_memtime() - (from_); } while (0)
#else
#define CR_V4_T(var_) do { } while (0)
#define CR_V4_ACC(acc_, from_) do { } while (0)
#endif

/* big-endian view of the coded bytes: lane l holds payload bytes [1 + base + 4l, +4), first byte in the
 * top bits; bytes past the end read as zero (CrSource does the same for the other decoders) */
CR_DEV uint32_t cr_v4_window(const uint8_t* payload, uint32_t size, uint32_t base) {
    const uint32_t o = 1u + base + cr_lane() * 4u;
    uint32_t v = 0;
    if (o + 4u <= size) v = *reinterpret_cast<const cr_u32u*>(payload + o);
    else for (uint32_t j = 0; j < 4; j++) if (o + j < size) v |= (uint32_t)payload[o + j] << (8 * j);
    cr_drain_loads();
    return __builtin_bswap32(v);
}

/* a wave-uniform pointer the compiler knows to be uniform (it came out of a vector load of one address) */
template <typename T>
CR_DEV T* cr_uni_ptr(T* p) {
    const u64 v = reinterpret_cast<u64>(p);
    return reinterpret_cast<T*>(((u64)cr_uni((uint32_t)(v >> 32)) << 32) | cr_uni((uint32_t)v));
}

CR_DEV uint32_t cr_rop_decode_v4(const uint8_t* src_, uint32_t n, uint8_t* dst_, uint32_t cap, uint8_t* arena_,
                                 const CrArenaLayout& L, u64* st) {
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
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
    cr_wave_sync();

    /* ---- coded bytes: range_decoder_init (cr-rangecoder.c:81-89) shifts five bytes through a 32-bit
     * register, i.e. cache = bytes 1..4; the shift register `ib` holds what follows, first byte on top */
    const uint8_t* const payload = src + CR_ROP_HEADER;
    const uint32_t psize = n - CR_ROP_HEADER;
    uint32_t wbase = 0, win = cr_v4_window(payload, psize, 0u);
    uint32_t cache = cr_lane_get(win, 0), range = 0xFFFFFFFFu;
    u64 ib = (u64)cr_lane_get(win, 1) << 32;
    uint32_t ibits = 32, widx = 2;
#define CR_V4_REFILL() do { \
        if (CR_UNLIKELY(widx == 64u)) { wbase += 256u; win = cr_v4_window(payload, psize, wbase); widx = 0; } \
        ib |= (u64)cr_lane_get(win, widx) << (32u - ibits); \
        widx++; ibits += 32u; \
    } while (0)
    /* range_decoder_decode, cr-rangecoder.c:91-99, with range = unit already divided */
#define CR_V4_CONSUME(lower_, frq_, unit_) do { \
        cache -= (lower_) * (unit_); \
        const uint32_t r2_ = (unit_) * (frq_); \
        const uint32_t n8_ = (uint32_t)__builtin_clz(r2_) & 0x18u; \
        range = r2_ << n8_; \
        cache = (uint32_t)(((((u64)cache << 32) | (ib >> 32)) << n8_) >> 32); \
        ib <<= n8_; ibits -= n8_; \
        if (CR_UNLIKELY(ibits <= 32u)) CR_V4_REFILL(); \
    } while (0)
    CR_V4_REFILL();

    uint32_t ctx = 0;
    uint32_t nd_key = 0xFFFFFFFFu, nd_w = 0, nd_x = 0;                   /* the node of the previous step, as stored */
    uint32_t o3_lk = 0xFFFFFFFFu, o3_lv = 0;                             /* the previous step's order-3 store: key, value */
    uint32_t lr_idx = 0xFFFFFFFFu, lr_row = 0;                           /* the previous step's order-1 row store (none: ~0) */
    uint32_t have = CR_LZP_SKIP, learned = CR_LZP_SKIP, after_esc = 0;
    u64 x8 = *reinterpret_cast<const cr_u64u*>(src + 10);                 /* the 8 bytes in front of the write position */
    x8 = ((u64)cr_uni((uint32_t)(x8 >> 32)) << 32) | cr_uni((uint32_t)x8);
    u64 pend_x = 0;                                                      /* lane j: those 8 bytes for position learned + j */
    cr_stamp(st, 4);

    const uint32_t vo_nodes = (uint32_t)L.off_nodes + lane * 4u, vo_o1 = (uint32_t)L.off_o1 + lane * 4u;
    const uint32_t so_nodes = (uint32_t)L.off_nodes, so_o3d = (uint32_t)L.off_o3d, so_scr = (uint32_t)L.off_dir + 4096u;
    uint32_t n_w = 0, n_x = 0, n_e = 0, n_row = 0;                       /* the loads' destination registers */
    uint32_t f_w, f_x, f_e, f_row;                                       /* the current step's model, as loaded */
    /* the four loads of a step's model: node word of this lane, node flag word, the aligned word holding
     * the order-3 entry, this lane's word of the order-1 row */
#define CR_V4_ISSUE(c_) do { \
        const uint32_t no_ = ((c_) & 0xffffu) * CRGPU_NODE_BYTES; \
        const uint32_t aw_ = vo_nodes + no_, ax_ = so_nodes + no_; \
        const uint32_t ae_ = so_o3d + ((cr_o3_key(c_) << 1) & ~3u), ar_ = vo_o1 + (((c_) & 0xffu) << 8); \
        asm volatile("global_load_dword %0, %4, %8\n\t" \
                     "global_load_dword %1, %5, %8 offset:256\n\t" \
                     "global_load_dword %2, %6, %8\n\t" \
                     "global_load_dword %3, %7, %8" \
                     : "+&v"(n_w), "+&v"(n_x), "+&v"(n_e), "+&v"(n_row) \
                     : "v"(aw_), "v"(ax_), "v"(ae_), "v"(ar_), "s"(arena) : "memory"); \
    } while (0)
    /* Wait until all but the last `behind_` vector-memory operations are complete, THEN copy the loaded
     * set into the current-step registers. The copy is inside the statement: whatever the compiler places
     * in front of an asm statement (it copies tied operands) could read a register a load still has to
     * write. Nothing is in flight into a register once this has run. */
#define CR_V4_TAKE(behind_) \
        asm volatile("s_waitcnt vmcnt(" #behind_ ")\n\t" \
                     "v_mov_b32 %0, %4\n\tv_mov_b32 %1, %5\n\tv_mov_b32 %2, %6\n\tv_mov_b32 %3, %7" \
                     : "=&v"(f_w), "=&v"(f_x), "=&v"(f_e), "=&v"(f_row) : "v"(n_w), "v"(n_x), "v"(n_e), "v"(n_row) : "memory")
    /* hidden stores are readable / nothing is in flight */
#define CR_V4_DRAIN() asm volatile("s_waitcnt vmcnt(0)" ::: "memory")
    CR_V4_ISSUE(ctx);
    CR_V4_TAKE(0);

#ifdef CR_V4_PROF
    u64 pf_take = 0, pf_match = 0, pf_nmatch = 0, pf_steps = 0, pf_esc = 0;
#endif
    CR_V4_T(pf_t0);
    while (have < total) {                                               /* cr-coder.c:259-290 */
        ctx = cr_uni(ctx); range = cr_uni(range); cache = cr_uni(cache);
        ib = ((u64)cr_uni((uint32_t)(ib >> 32)) << 32) | cr_uni((uint32_t)ib);
        ibits = cr_uni(ibits); widx = cr_uni(widx); wbase = cr_uni(wbase);
        nd_key = cr_uni(nd_key); nd_x = cr_uni(nd_x); o3_lk = cr_uni(o3_lk); o3_lv = cr_uni(o3_lv); lr_idx = cr_uni(lr_idx);
        have = cr_uni(have); learned = cr_uni(learned); after_esc = cr_uni(after_esc);

        /* ---- this step's model */
        const uint32_t key = ctx & 0xffffu;
        const uint32_t fx = cr_uni(f_x);
        uint32_t w = f_w, x = fx & 0xffffu;
        bool store_all = false;
        if (CR_UNLIKELY((fx >> 16) != gen)) { w = 0; x = 0x0101u; store_all = true; }   /* stale tag: first use in this block = o2_model_init */
        if (CR_UNLIKELY(key == nd_key)) { w = nd_w; x = nd_x; store_all = false; }       /* loaded before the previous step's stores */
        const uint32_t w_was = w;
        const uint32_t k3 = cr_o3_key(ctx);
        uint32_t e = (cr_uni(f_e) >> ((k3 & 1u) << 4)) & 0xffffu;
        if (CR_UNLIKELY(k3 == o3_lk)) e = o3_lv;
        uint32_t pred = e >> 8, conf = e & 15u;
        if (CR_UNLIKELY(((e >> 4) & 15u) != g3)) { pred = 0; conf = 0; }  /* stale: the reference's zero-filled table */
        const uint32_t row_idx = ctx & 0xffu;
        uint32_t row = f_row;
        if (CR_UNLIKELY(row_idx == lr_idx)) row = lr_row;

        /* ---- ppm_decode, cr-ppm.c:169-235 */
        const uint32_t f_hit = x & 0xffu, f_esc = x >> 8;
        const uint32_t pmask = 0xffu << ((pred & 3u) * 8u);
        const uint32_t wx = (lane == (pred >> 2)) ? (w & ~pmask) : w;
        const uint32_t mysum = cr_bytesum(wx);
        const uint32_t incl = cr_scan_incl(mysum);
        const uint32_t bytes = cr_lane_get(incl, 63);
        const uint32_t unit = cr_uni(range / (bytes + f_hit + f_esc));                   /* cr-rangecoder.c:101-104 */
        const uint32_t tb = bytes * unit;
        uint32_t s, lower, frq;
        if (cache < tb) {
            /* first lane whose inclusive cumulative count, times unit, exceeds cache */
            const uint32_t ol = (uint32_t)__builtin_popcountll(cr_ballot(incl * unit <= cache));
            const uint32_t ww = cr_lane_get(wx, ol), a0 = cr_lane_get(incl, ol) - cr_lane_get(mysum, ol);
            const uint32_t b0 = ww & 0xffu, b1 = (ww >> 8) & 0xffu, b2 = (ww >> 16) & 0xffu;
            const uint32_t a1 = a0 + b0, a2 = a1 + b1, a3 = a2 + b2;
            const uint32_t c1 = cache >= a1 * unit, c2 = cache >= a2 * unit, c3 = cache >= a3 * unit;
            const uint32_t j = c1 + c2 + c3;
            lower = c3 ? a3 : (c2 ? a2 : (c1 ? a1 : a0));
            frq = (ww >> (8u * j)) & 0xffu;
            s = ol * 4u + j;
        } else {
            const bool hit = cache < tb + f_hit * unit;
            s = hit ? 256u : 257u;
            lower = hit ? bytes : bytes + f_hit;
            frq = hit ? f_hit : f_esc;
        }
        CR_V4_CONSUME(lower, frq, unit);
        uint32_t sym = s == 256u ? pred : s;
        uint32_t esc_halved = 0;
        if (s == 257u) {                                                 /* cr-ppm.c:209-232 */
            const uint32_t ne = (f_esc + 1u) & 0xffu;
            x = (x & 0x00ffu) | (ne << 8);
            if (CR_UNLIKELY(ne > 250u)) { cr_lean_halve(w, x); esc_halved = 1; }
            uint32_t keep = cr_zero_bytes(w);
            if (lane == (pred >> 2)) keep &= ~pmask;
            const uint32_t mine = cr_o1_weight_sum(row, keep);
            const uint32_t incl1 = cr_scan_incl(mine);
            const uint32_t all = cr_lane_get(incl1, 63);
            const uint32_t unit1 = cr_uni(range / all);
            uint32_t got = 0, lo = 0, fo = 1;
            if (CR_LIKELY(cache < all * unit1)) {
                const uint32_t ol = (uint32_t)__builtin_popcountll(cr_ballot(incl1 * unit1 <= cache));
                const uint32_t rw = cr_lane_get(row, ol), kp = cr_lane_get(keep, ol), a0 = cr_lane_get(incl1, ol) - cr_lane_get(mine, ol);
                const uint32_t q0 = (kp & 0x000000ffu) ? ((rw & 0xffu) * 8u - 7u) : 0u;
                const uint32_t q1 = (kp & 0x0000ff00u) ? (((rw >> 8) & 0xffu) * 8u - 7u) : 0u;
                const uint32_t q2 = (kp & 0x00ff0000u) ? (((rw >> 16) & 0xffu) * 8u - 7u) : 0u;
                const uint32_t a1 = a0 + q0, a2 = a1 + q1, a3 = a2 + q2;
                const uint32_t c1 = cache >= a1 * unit1, c2 = cache >= a2 * unit1, c3 = cache >= a3 * unit1;
                const uint32_t j = c1 + c2 + c3;
                lo = c3 ? a3 : (c2 ? a2 : (c1 ? a1 : a0));
                got = ol * 4u + j;
                fo = ((rw >> (8u * j)) & 0xffu) * 8u - 7u;
            }
            CR_V4_CONSUME(lo, fo, unit1);
            sym = got;
            /* ppm_update_o1, cr-ppm.c:90-97 */
            const uint32_t cur = cr_table_byte(row, sym);
            if (lane == (sym >> 2)) row += 1u << ((sym & 3u) * 8u);
            if (CR_UNLIKELY(cur + 1u >= 255u)) row -= (row >> 1) & 0x7f7f7f7fu;
            lr_idx = row_idx; lr_row = row;
        } else {
            lr_idx = 0xFFFFFFFFu;
        }
        sym = cr_uni(sym);

        /* ---- what the symbol means (cr-coder.c:261-289) */
        const bool esc_tok = !after_esc && sym == esc;                   /* escape byte: a length or a 0 follows */
        const bool lit_esc = after_esc && sym == 0u;                     /* ... 0: the escape byte itself is the literal */
        const bool is_match = after_esc && sym != 0u;
        const bool is_lit = !(esc_tok || is_match);
        const uint32_t lit = lit_esc ? esc : sym;
        uint32_t newctx = (ctx << 8) | lit;                              /* the escape token pushes `esc`, which is sym */
        const uint32_t have_at = have;
        after_esc = esc_tok ? 1u : 0u;
        if (CR_UNLIKELY(is_match)) {
            const uint32_t len = sym;
            if (have + len > total || have + len > cap) { CR_V4_DRAIN(); return 0xFFFFFFFFu; }  /* corrupt stream */
            CR_V4_T(pf_m0);
            CR_V4_DRAIN();                                               /* the literals' stores are readable */
            cr_wave_sync();
            uint32_t c8, c4, c2;
            cr_lzp_learn_predict(z, pend_x, learned, have - learned, x8, c8, c4, c2);
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
            /* only the last four pushes survive in the 32-bit context; they sit in the lanes that copied
             * them (the last batch holds bytes len-1, len-2, ... in lanes (len-1)&63, ...) */
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
                /* the copied positions become pending: lane i held byte have+i; xa = the 8 bytes ending there */
                uint32_t t = mine & 0xffu;
                u64 xa = (u64)t << 56;
#pragma unroll
                for (uint32_t k = 1; k < 8u; k++) {
                    t = cr_shift_up1(t, (uint32_t)(x8 >> (8u * (8u - k))) & 0xffu);
                    xa |= (u64)t << (8u * (7u - k));
                }
                const uint32_t lo = cr_shift_up1((uint32_t)xa, (uint32_t)x8), hi = cr_shift_up1((uint32_t)(xa >> 32), (uint32_t)(x8 >> 32));
                pend_x = ((u64)hi << 32) | lo;                           /* lane 0: position have, lane j: have+j */
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
            CR_V4_DRAIN();
#ifdef CR_V4_PROF
            CR_V4_ACC(pf_match, pf_m0); pf_nmatch++;
#endif
        }
        if (CR_LIKELY(is_lit)) {                                         /* a literal byte at `have` (register bookkeeping only) */
            if (lane == have - learned) pend_x = x8;
            x8 = (x8 >> 8) | ((u64)lit << 56);
            have++;
            if (CR_UNLIKELY(have - learned == CRGPU_WAVE)) { cr_lzp_learn(z, pend_x, learned + lane); learned = have; }
        }

        /* ---- next step's loads */
        newctx = cr_uni(newctx);
        CR_V4_ISSUE(newctx);

        /* ---- model updates (cr-ppm.c:199-232), in registers */
        {
            const bool was_hit = s == 256u, was_byte = s < 256u;
            /* o2_model_update of the coded byte (+1), except after a halving escape (cr-ppm.c:160-162) */
            const bool bump = !was_hit && !esc_halved;
            const uint32_t addv = bump ? (1u << ((sym & 3u) * 8u)) : 0u;
            if (lane == (sym >> 2)) w += addv;
            const uint32_t hit1 = (f_hit + (was_hit ? 1u : 0u)) & 0xffu;
            uint32_t esc1 = x >> 8;                                      /* already counts this step's escape */
            if (was_byte && frq == 1u) esc1 = (esc1 - 1u) & 0xffu;       /* PPMX singleton rule, cr-ppm.c:136-138 */
            x = hit1 | (esc1 << 8);
            const bool halve = was_hit ? hit1 > 250u : (was_byte && (frq + 1u > 250u || (frq == 1u && esc1 > 250u)));
            if (CR_UNLIKELY(halve)) cr_lean_halve(w, x);
            /* ppm_update_o3, cr-ppm.c:69-88: hit -> min(conf + 1, 15); miss -> (c>1)+(c>2)+(c>4)+(c>8), replace at 0 */
            const u64 lut = was_hit ? 0xFFEDCBA987654321ull : 0x4444444333322100ull;
            uint32_t c = (uint32_t)(lut >> (conf * 4u)) & 15u;
            if (c == 0u) { pred = sym; c = 1u; }
            conf = c;
        }
        /* ---- the step's five stores, each by every lane (a lane whose word did not change, and every
         * lane of a step without an order-1 update, aims at a scratch word instead) */
        {
            const uint32_t no = key * CRGPU_NODE_BYTES;
            const uint32_t a_w = (w != w_was || store_all) ? vo_nodes + no : so_scr + 256u;
            const uint32_t a_x = so_nodes + no, v_x = x | (gen << 16);
            const uint32_t val3 = (pred << 8) | (g3 << 4) | conf;
            const uint32_t a_e = so_o3d + (k3 << 1);
            const uint32_t a_r = (s == 257u) ? vo_o1 + (row_idx << 8) : so_scr;
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
        /* ---- the model of the next step: its loads are older than the five stores */
        CR_V4_T(pf_w0);
        CR_V4_TAKE(5);
#ifdef CR_V4_PROF
        CR_V4_ACC(pf_take, pf_w0); pf_steps++; pf_esc += (s == 257u);
#endif
        ctx = newctx;
    }
    CR_V4_DRAIN();
#ifdef CR_V4_PROF
    if (st && lane == 0) { st[8] = __builtin_amdgcn_s_memtime() - pf_t0; st[9] = pf_take; st[10] = pf_match; st[11] = pf_nmatch; st[12] = pf_steps; st[13] = pf_esc; }
#endif
#undef CR_V4_ISSUE
#undef CR_V4_TAKE
#undef CR_V4_DRAIN
#undef CR_V4_REFILL
#undef CR_V4_CONSUME
    cr_stamp(st, 5);
    return have;
}

#endif

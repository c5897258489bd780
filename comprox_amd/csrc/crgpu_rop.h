/*
 * comprox_amd/csrc/crgpu_rop.h — comprop block codec (lzencode / lzdecode) for one wavefront.
 *
 * Reference: /root/reference/src/ropmain/cr-coder.c:119-292. Block layout (cr-coder.c:59-66,
 * sizeof == 20): [0] coded flag, [4..7] original size LE, [8] escape byte, [9..17] first nine
 * bytes, the rest zero; then the range-coder bytes. Stored form: 20 zero bytes + the raw input.
 */
#ifndef CRGPU_ROP_H
#define CRGPU_ROP_H

#include "crgpu_ppm.h"
#include "crgpu_lzp.h"

#define CR_ROP_HEADER 20u

struct CrShared {
    uint32_t hist[256];
};

/* 256-byte register window over a byte array: lane l holds bytes base+4l..base+4l+3 */
struct CrWindow {
    const uint8_t* p;
    uint32_t size, base, word;
};
CR_DEV void cr_window_fill(CrWindow& w, uint32_t at) {
    w.base = at;
    uint32_t o = at + cr_lane() * 4u, v = 0;
    if (o + 4u <= w.size) v = *reinterpret_cast<const cr_u32u*>(w.p + o);
    else for (uint32_t j = 0; j < 4; j++) if (o + j < w.size) v |= (uint32_t)w.p[o + j] << (8 * j);
    w.word = v;
    cr_drain_loads();
}
CR_DEV void cr_window_init(CrWindow& w, const uint8_t* p, uint32_t size, uint32_t at) {
    w.p = p; w.size = size;
    cr_window_fill(w, at);
}
CR_DEV uint32_t cr_window_at(CrWindow& w, uint32_t pos) {
    uint32_t rel = pos - w.base;
    if (rel >= 256u) { cr_window_fill(w, pos); rel = 0; }
    return cr_table_byte(w.word, rel);
}

/* least frequent byte value, lowest value on ties (cr-coder.c:147-156) */
CR_DEV uint32_t cr_pick_escape(const uint8_t* d, uint32_t n, uint32_t* hist) {
    const uint32_t lane = cr_lane();
    for (uint32_t i = lane; i < 256u; i += CRGPU_WAVE) hist[i] = 0;
    cr_wave_sync();
    const uint32_t body = n & ~3u;
    for (uint32_t i = lane * 4u; i < body; i += 4u * CRGPU_WAVE) {
        uint32_t v = *reinterpret_cast<const cr_u32u*>(d + i);
        atomicAdd(&hist[v & 0xffu], 1u);
        atomicAdd(&hist[(v >> 8) & 0xffu], 1u);
        atomicAdd(&hist[(v >> 16) & 0xffu], 1u);
        atomicAdd(&hist[v >> 24], 1u);
    }
    if (lane < (n & 3u)) atomicAdd(&hist[d[body + lane]], 1u);
    cr_wave_sync();
    u64 best = ~0ull;
    for (uint32_t j = 0; j < 4; j++) {
        uint32_t v = lane * 4u + j;
        u64 cand = ((u64)hist[v] << 8) | v;
        best = cand < best ? cand : best;
    }
    for (int dlt = 32; dlt; dlt >>= 1) {
        u64 o = __shfl_xor(best, dlt);
        best = o < best ? o : best;
    }
    cr_wave_sync();
    return (uint32_t)best & 0xffu;
}

CR_DEV void cr_rop_store_raw(const uint8_t* src, uint32_t n, uint8_t* dst) {   /* cr-coder.c:222-228 */
    const uint32_t lane = cr_lane();
    if (lane < CR_ROP_HEADER) dst[lane] = 0;
    for (uint32_t i = lane; i < n; i += CRGPU_WAVE) dst[CR_ROP_HEADER + i] = src[i];
}

/* lzencode, cr-coder.c:119-229. Returns the number of bytes written at dst. */
CR_DEV void cr_stamp(u64* st, int slot) { if (st && cr_lane() == 0) st[slot] = wall_clock64(); }

CR_DEV uint32_t cr_rop_encode_block(const uint8_t* src, uint32_t n, uint8_t* dst, const uint8_t* lens, uint8_t* arena,
                                    const CrArenaLayout& L, uint32_t fresh, uint32_t persist, CrShared& sh, u64* st) {
    const uint32_t lane = cr_lane();
    cr_stamp(st, 0);
    if (n < 16u) { cr_rop_store_raw(src, n, dst); return CR_ROP_HEADER + n; }      /* cr-coder.c:140-142 */

    const uint32_t esc = cr_pick_escape(src, n, sh.hist);
    cr_stamp(st, 1); cr_stamp(st, 2); cr_stamp(st, 3);
    /* `lens`: LZP agreement length at every position, produced by k_rop_lzp (cr-coder.c:95-118) */

    CrPpm m;
    cr_ppm_attach(m, arena, L, persist ? L.cap_o3 : cr_log2_ceil_pow2(2u * n, 1024u, L.cap_o3));
    if (fresh) cr_ppm_reset(m); else cr_ppm_resume(m);
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
    cr_wave_sync();

    cr_stamp(st, 4);
    CrSink out; out.dst = dst + CR_ROP_HEADER; out.n = 0;
    CrRc rc; cr_rc_init(rc);
    CrWindow win, lwin;
    cr_window_init(win, src, n, CR_LZP_SKIP);
    cr_window_init(lwin, lens, n, CR_LZP_SKIP);

    uint32_t pos = CR_LZP_SKIP, ntok = 0;
    bool stored = false;
#ifdef CRGPU_PROF
    CrProf prof; prof.last = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < 8; i++) prof.acc[i] = 0;
#endif
    CrFetch F; F.valid = 0; F.ctx = 0; F.with_row = 0;
    m.defer = 1;
    /* One ppm_encode call site per loop pass (a token is one or two passes): with several inlined
     * copies the prefetched registers would be merged by copies, and a copy waits for the load.
     * phase 0 = first symbol of the token at `pos`, 1 = second symbol (match length, or the 0 that
     * marks a literal escape byte). */
    uint32_t len = 1, c = 0, phase = 0;
    if (pos < n) {
        if (pos + CR_LZP_TAIL < n) len = cr_window_at(lwin, pos);
        c = cr_window_at(win, pos);
    }
    while (pos < n) {                                                    /* cr-coder.c:169-207 */
        const bool two = len > 1u || c == esc;           /* esc + length / esc + 0 */
        uint32_t sym, next;
        bool done;
        if (phase == 0) {
            sym = two ? esc : c;
            next = (m.ctx << 8) | sym;                   /* both cases push exactly `sym` next */
            done = !two;
        } else {
            sym = len > 1u ? len : 0u;
            next = len > 1u ? cr_uni(__builtin_bswap32(*reinterpret_cast<const cr_u32u*>(src + pos + len - 4u)))
                            : ((m.ctx << 8) | esc);
            done = true;
        }
        cr_ppm_encode(m, rc, sym, out, F, next, 1u CR_PROF_PASS);
        m.ctx = next;
        if (!done) { phase = 1; continue; }
        phase = 0;
        pos += len; ntok++;
        if (CR_ROP_HEADER + out.n >= n) { stored = true; break; }        /* cr-coder.c:204-206 */
        if (pos < n) {
            len = 1;
            if (pos + CR_LZP_TAIL < n) len = cr_window_at(lwin, pos);
            c = cr_window_at(win, pos);
        }
    }
    cr_node_writeback(m);
    if (persist) cr_ppm_suspend(m);
    cr_stamp(st, 5);
    if (st && lane == 0) { st[6] = m.nnodes; st[7] = ntok; }
#ifdef CRGPU_PROF
    if (st && lane == 0) for (int i = 0; i < 8; i++) st[8 + i] = prof.acc[i];
#endif
    if (stored) {
        cr_wave_sync();
        cr_rop_store_raw(src, n, dst);
        return CR_ROP_HEADER + n;
    }
    cr_rc_flush(rc, out);                                                /* cr-coder.c:210 */
    cr_sink_finish(out);
    if (lane < CR_ROP_HEADER) {                                          /* cr-coder.c:213-216 */
        uint32_t v = 0;
        if (lane == 0) v = 1;
        else if (lane >= 4 && lane < 8) v = (n >> (8u * (lane - 4u))) & 0xffu;
        else if (lane == 8) v = esc;
        else if (lane >= 9 && lane < 18) v = src[lane - 9u];
        dst[lane] = (uint8_t)v;
    }
    return CR_ROP_HEADER + out.n;
}

/* lzdecode, cr-coder.c:231-292. Returns the decoded size or 0xFFFFFFFF. */
CR_DEV uint32_t cr_rop_decode_block(const uint8_t* src, uint32_t n, uint8_t* dst, uint32_t cap, uint8_t* arena,
                                    const CrArenaLayout& L, uint32_t fresh, uint32_t persist, CrShared& sh, u64* st) {
    (void)sh;
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
    CrPpm m;
    cr_ppm_attach(m, arena, L, persist ? L.cap_o3 : cr_log2_ceil_pow2(2u * total, 1024u, L.cap_o3));
    if (fresh) cr_ppm_reset(m); else cr_ppm_resume(m);
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
    cr_wave_sync();

    cr_stamp(st, 4);
    CrSource in;
    cr_source_init(in, src + CR_ROP_HEADER, n - CR_ROP_HEADER);
    CrRc rc; cr_rc_dec_init(rc, in);

#ifdef CRGPU_PROF
    CrProf prof; prof.last = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < 8; i++) prof.acc[i] = 0;
#endif
    CrFetch F; F.valid = 0; F.ctx = 0; F.with_row = 1;
    uint32_t have = CR_LZP_SKIP;       /* bytes produced */
    uint32_t learned = CR_LZP_SKIP;    /* positions < learned are in the LZP tables */
    uint32_t after_esc = 0;            /* the symbol being decoded is the one that follows an escape byte */
    /* The LZP tables are only consulted at match tokens, so learning is batched up to there
     * (cr-coder.c:284-288 feeds every produced byte to matcher_update). The 8 bytes in front of the
     * write position are kept in a register, and lane j remembers them for position learned+j: a
     * match token can then insert and look up without reading the output back. */
    u64 x8 = *reinterpret_cast<const cr_u64u*>(src + 10);                 /* bytes 1..8 of the block */
    x8 = ((u64)cr_uni((uint32_t)(x8 >> 32)) << 32) | cr_uni((uint32_t)x8);
    u64 pend_x = 0;
#define CR_DEC_LITERAL(byte_) do { \
        if (lane == 0) dst[have] = (uint8_t)(byte_); \
        if (lane == have - learned) pend_x = x8; \
        x8 = (x8 >> 8) | ((u64)(byte_) << 56); \
        have++; \
        if (have - learned == CRGPU_WAVE) { cr_lzp_learn(z, pend_x, learned + lane); learned = have; } \
    } while (0)
    while (have < total) {                                               /* cr-coder.c:259-290 */
        /* one ppm_decode call site per pass (see the encoder's loop for why) */
        const uint32_t s = cr_ppm_decode(m, rc, in, F CR_PROF_PASS);
        if (!after_esc) {
            if (s != esc) {
                CR_DEC_LITERAL(s);
                cr_ppm_push(m, s);
            } else {
                cr_ppm_push(m, esc);
                after_esc = 1;
            }
            continue;
        }
        after_esc = 0;
        const uint32_t len = s;
        if (len == 0u) {
            CR_DEC_LITERAL(esc);
            cr_ppm_push(m, esc);
            continue;
        }
        if (have + len > total || have + len > cap) return 0xFFFFFFFFu;  /* corrupt stream */
        cr_wave_sync();                                                  /* the literals' stores are readable */
        uint32_t c8, c4, c2;
        cr_lzp_learn_predict(z, pend_x, learned, have - learned, x8, c8, c4, c2);
        learned = have;
        /* matcher_getpos' two context checks (cr-matcher.c:59-73) and the first 64 source bytes of
         * all three candidates in one round trip. Byte-serial copy semantics (cr-coder.c:277-279):
         * a source that overlaps the destination repeats with period have - from. */
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
        /* only the last four pushes survive in the 32-bit context; they sit in the lanes that
         * copied them (the last batch holds bytes len-1, len-2, ... in lanes (len-1)&63, ...) */
        if (len >= 4u && ((len - 1u) & 63u) >= 3u) {
            uint32_t l3 = (len - 1u) & 63u;
            m.ctx = (cr_lane_get(mine, l3 - 3u) << 24) | (cr_lane_get(mine, l3 - 2u) << 16) |
                    (cr_lane_get(mine, l3 - 1u) << 8) | cr_lane_get(mine, l3);
        } else {
            cr_wave_sync();
            uint32_t k = len < 4u ? len : 4u;
            for (uint32_t i = len - k; i < len; i++) cr_ppm_push(m, cr_uni(dst[have + i]));
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
            pend_x = ((u64)hi << 32) | lo;                               /* lane 0: position have, lane j: have+j */
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
    }
#undef CR_DEC_LITERAL
    cr_node_writeback(m);
    if (persist) cr_ppm_suspend(m);
    cr_stamp(st, 5);
#ifdef CRGPU_PROF
    if (st && lane == 0) for (int i = 0; i < 8; i++) st[8 + i] = prof.acc[i];
#endif
    return have;
}

#ifdef CRGPU_PROF2
#define CR_PROF2_MARK(slot) CR_PROF_MARK(slot)
#else
#define CR_PROF2_MARK(slot) do { } while (0)
#endif
#endif

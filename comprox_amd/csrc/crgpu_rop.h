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
/* ------------------------------------------------------------------------------------------------
 * lzdecode again, laid out for the one thing that bounds it: a block decodes one symbol after the
 * other, every step needs the model of a context that is only known when the previous symbol is,
 * so a step costs one memory round trip plus its instructions, and nothing else on the chip can
 * help that block. This version keeps the step straight-line:
 *   - the next context's loads (node, order-3 group, order-1 row) are issued the moment the symbol is
 *     known, by every lane, unconditionally;
 *   - the model is written through: every step stores the node's changed words, its flag word, the
 *     order-3 entry, an order-1 row and one output byte, each with ONE unpredicated store by all lanes
 *     (lanes or steps with nothing to write aim that store at a scratch word). A fixed number of
 *     stores behind the loads lets the wait at the top of the next step be "all but the last five",
 *     i.e. the loads only; no dirty tracking, no write-back on leaving a node;
 *   - all lanes store and all lanes load the same words, so a later load of the same address is
 *     ordered behind the store per lane; a load issued BEFORE a store of the same step is patched from
 *     registers (same node, last order-3 slot, last order-1 row).
 * Used for the batched API (fresh model per block); the persist mode keeps the function above. */
CR_DEV void cr_lean_halve(uint32_t& w, uint32_t& x) {                    /* cr-o2model.c:54-71 */
    w = (w >> 1) & 0x7f7f7f7fu;
    const uint32_t singles = 1u + cr_sum(cr_count_ones_bytes(w));
    x = (((x & 0xffu) + 1u) >> 1) | ((singles & 0xffu) << 8);
}

CR_DEV uint32_t cr_rop_decode_lean(const uint8_t* src, uint32_t n, uint8_t* dst, uint32_t cap, uint8_t* arena,
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
    CrPpm m;
    cr_ppm_attach(m, arena, L, cr_log2_ceil_pow2(2u * total, 1024u, L.cap_o3));
    cr_ppm_reset(m);
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
    cr_wave_sync();
    const uint32_t gen = cr_uni(m.gen), o3_mask = m.o3_mask;
    uint32_t* const nodes = m.nodes;
    u64* const o3 = m.o3;
    uint8_t* const o1 = m.o1;
    uint8_t* const scratch = arena + L.off_dir + 4096u;                   /* 1 KiB nobody reads */

    CrSource in;
    cr_source_init(in, src + CR_ROP_HEADER, n - CR_ROP_HEADER);
    CrRc rc; cr_rc_dec_init(rc, in);

    uint32_t ctx = 0;
    uint32_t nd_key = 0xFFFFFFFFu, nd_w = 0, nd_x = 0;                   /* the node of the previous step, as stored */
    uint32_t o3_ls = 0xFFFFFFFFu; u64 o3_lv = 0;                         /* last order-3 store */
    uint32_t lr_idx = 0xFFFFFFFFu, lr_row = 0;                           /* last order-1 row store */
    uint32_t have = CR_LZP_SKIP, learned = CR_LZP_SKIP, after_esc = 0;
    u64 x8 = *reinterpret_cast<const cr_u64u*>(src + 10);
    x8 = ((u64)cr_uni((uint32_t)(x8 >> 32)) << 32) | cr_uni((uint32_t)x8);
    u64 pend_x = 0;

    uint32_t f_w, f_x, f_row, f_h; u64 f_v;
#define CR_LEAN_ISSUE(c_) do { \
        const uint32_t* p_ = nodes + (u64)((c_) & 0xffffu) * CRGPU_NODE_WORDS; \
        f_w = p_[lane]; f_x = p_[64]; \
        f_h = cr_o3_home(m, cr_o3_key(c_)); \
        f_v = o3[(f_h + (lane & 7u)) & o3_mask]; \
        f_row = reinterpret_cast<const uint32_t*>(o1 + (((c_) & 0xffu) << 8))[lane]; \
    } while (0)
    CR_LEAN_ISSUE(ctx);
    cr_stamp(st, 4);
#ifdef CRGPU_PROF
    CrProf prof; prof.last = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < 8; i++) prof.acc[i] = 0;
#endif

    while (have < total) {                                               /* cr-coder.c:259-290 */
        CR_PROF_MARK(0);
        cr_rc_pin(rc);
        in.pos = cr_uni(in.pos); in.base = cr_uni(in.base);
        ctx = cr_uni(ctx); nd_key = cr_uni(nd_key); nd_x = cr_uni(nd_x);
        /* ---- this step's model: node, order-3 entry, order-1 row */
        const uint32_t key = ctx & 0xffffu;
        const uint32_t fx = cr_uni(f_x);
        const bool same = key == nd_key, live = (fx >> 16) == gen;       /* stale tag: node not yet used in this block (o2_model_init) */
        uint32_t w = same ? nd_w : (live ? f_w : 0u);
        uint32_t x = same ? nd_x : (live ? (fx & 0xffffu) : 0x0101u);
        const uint32_t w_was = w;                                        /* what memory holds (nothing valid for a node's first use) */
        const uint32_t k3 = cr_o3_key(ctx) | 0x80000000u;
        u64 v = f_v;
        if (((f_h + (lane & 7u)) & o3_mask) == o3_ls) v = o3_lv;
        u64 hits = cr_ballot(v == 0ull || (uint32_t)(v >> 32) == k3);
        uint32_t probe = 0;
        while (!hits) {                                                  /* rare: the home group is taken by other keys */
            probe += 8u;
            v = o3[(f_h + probe + (lane & 7u)) & o3_mask];
            if (((f_h + probe + (lane & 7u)) & o3_mask) == o3_ls) v = o3_lv;
            hits = cr_ballot(v == 0ull || (uint32_t)(v >> 32) == k3);
        }
        const uint32_t first = (uint32_t)__builtin_ctzll(hits);
        const u64 got3 = cr_lane_get64(v, first);
        const uint32_t slot = (f_h + probe + first) & o3_mask;
        uint32_t pred = (uint32_t)(got3 >> 8) & 0xffu, conf = (uint32_t)got3 & 0xfu;   /* empty slot: 0, 0 like the reference's zeroed table */
        const uint32_t row_idx = ctx & 0xffu;
        uint32_t row = (row_idx == lr_idx) ? lr_row : f_row;
#ifdef CRGPU_PROF
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#endif
        CR_PROF_MARK(1);

        /* ---- ppm_decode, cr-ppm.c:169-235 */
        const uint32_t f_hit = x & 0xffu, f_esc = (x >> 8) & 0xffu;
        uint32_t wx = w;
        if (lane == (pred >> 2)) wx &= ~(0xffu << ((pred & 3u) * 8u));
        const uint32_t mysum = cr_bytesum(wx);
        const uint32_t incl = cr_scan_incl(mysum);
        const uint32_t bytes = cr_lane_get(incl, 63);
        const uint32_t target = cr_rc_dec_target(rc, bytes + f_hit + f_esc);
        uint32_t s, lower, frq;
        if (target < bytes) {
            const uint32_t excl = incl - mysum;
            const u64 owner = cr_ballot(excl <= target && target < incl);
            const uint32_t ol = (uint32_t)__builtin_ctzll(owner);
            const uint32_t ww = cr_lane_get(wx, ol), before = cr_lane_get(excl, ol);
            const uint32_t j = cr_pick_in_word(ww & 0xffu, (ww >> 8) & 0xffu, (ww >> 16) & 0xffu, ww >> 24, before, target, lower);
            s = ol * 4u + j;
            frq = (ww >> (8u * j)) & 0xffu;
        } else if (target < bytes + f_hit) {
            s = 256u; lower = bytes; frq = f_hit;
        } else {
            s = 257u; lower = bytes + f_hit; frq = f_esc;
        }
        CR_PROF_MARK(2);
        cr_rc_dec_consume(rc, lower, frq, in);
        CR_PROF_MARK(3);
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
            const uint32_t t1 = cr_rc_dec_target(rc, all);
            const uint32_t excl1 = incl1 - mine;
            const u64 owner = cr_ballot(excl1 <= t1 && t1 < incl1);
            uint32_t got = 0, lo = 0, fo = 1;
            if (owner) {
                const uint32_t ol = (uint32_t)__builtin_ctzll(owner);
                const uint32_t rw = cr_lane_get(row, ol), kp = cr_lane_get(keep, ol), before = cr_lane_get(excl1, ol);
                const uint32_t q0 = (kp & 0x000000ffu) ? ((rw & 0xffu) * 8u - 7u) : 0u;
                const uint32_t q1 = (kp & 0x0000ff00u) ? (((rw >> 8) & 0xffu) * 8u - 7u) : 0u;
                const uint32_t q2 = (kp & 0x00ff0000u) ? (((rw >> 16) & 0xffu) * 8u - 7u) : 0u;
                const uint32_t q3 = (kp & 0xff000000u) ? ((rw >> 24) * 8u - 7u) : 0u;
                const uint32_t j = cr_pick_in_word(q0, q1, q2, q3, before, t1, lo);
                got = ol * 4u + j;
                fo = ((rw >> (8u * j)) & 0xffu) * 8u - 7u;
            }
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
        CR_PROF_MARK(4);

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
                if (have + len > total || have + len > cap) return 0xFFFFFFFFu;  /* corrupt stream */
                CR_PROF2_MARK(4);
                cr_wave_sync();                                          /* the literals' stores are readable */
                CR_PROF2_MARK(5);
                uint32_t c8, c4, c2;
                cr_lzp_learn_predict(z, pend_x, learned, have - learned, x8, c8, c4, c2);
                CR_PROF2_MARK(6);
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
                CR_PROF2_MARK(7);
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
            }
        }
        CR_PROF2_MARK(0);
        if (lit_dst != scratch + 512u) {                                 /* a literal byte at `have` (register bookkeeping only) */
            if (lane == have - learned) pend_x = x8;
            x8 = (x8 >> 8) | ((u64)lit << 56);
            have++;
            if (have - learned == CRGPU_WAVE) { cr_lzp_learn(z, pend_x, learned + lane); learned = have; }
        }

        /* ---- next step's loads */
        newctx = cr_uni(newctx);
        CR_PROF_MARK(5);
        CR_LEAN_ISSUE(newctx);
        CR_PROF_MARK(6);

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
        /* ---- the step's five stores, each by every lane */
        uint32_t* np = nodes + (u64)key * CRGPU_NODE_WORDS;
        /* (a lane whose word did not change, and every lane of a step without an order-1 update,
         * aims at one scratch word instead: same instruction count, a fraction of the written lines) */
        uint32_t* wdst = (w != w_was || !(same || live)) ? np + lane : reinterpret_cast<uint32_t*>(scratch + 256u);
        *wdst = w;
        np[64] = x | (gen << 16);
        const u64 val3 = ((u64)k3 << 32) | (u64)(pred << 8) | (u64)conf;
        o3[slot] = val3;
        reinterpret_cast<uint32_t*>(row_dst)[row_dst == scratch ? 0u : lane] = row;
        *lit_dst = (uint8_t)lit;
        nd_key = key; nd_w = w; nd_x = x;
        o3_ls = slot; o3_lv = val3;
        ctx = newctx;
        CR_PROF_MARK(7);
    }
#undef CR_LEAN_ISSUE
    cr_stamp(st, 5);
#ifdef CRGPU_PROF
    if (st && lane == 0) for (int i = 0; i < 8; i++) st[8 + i] = prof.acc[i];
#endif
    return have;
}

#endif

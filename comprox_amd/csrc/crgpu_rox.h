/*
 * comprox_amd/csrc/crgpu_rox.h — comprox block codec (lzencode / lzdecode of src/roxmain) on gfx950.
 *
 * Reference: /root/reference/src/roxmain/cr-coder.c:88-114,153-318,390-526 and cr-matcher.c:34-340
 * (lazy parsing by default, the -f "flexible parsing" of cr-matcher.c:253-289 on request). Block layout (cr-coder.c:69-81, 32 bytes):
 * [0] coded flag, [1] match_min, [2] esc, [3] 0, then u32 LE original size, #spos, #pos, #len codes,
 * offsets of the spos / pos / len streams; body = main PPM stream, spos, pos, len streams.
 *
 * What is parse-independent is computed for every position up front, position-parallel:
 *   chain links   prev[p]  = largest q < p (both < n-255) of the same hash class (the reference's two
 *                            bucket passes, cr-matcher.c:89-148, produce exactly this)
 *   near links    nprev[p] = what the 65 536-entry short cache holds when p is looked up: every
 *                            earlier position has been fed to matcher_update_cache by then
 *   long match    ML[p]    = match(p, match_min, limit, 0) after the lazy-evaluation veto
 *                            (cr-matcher.c:292-310) — a pure function of the data
 *   near match    NL[p]    = agreement length with nprev[p] when it is within 256 bytes
 * Only the repeat-distance logic and the final choice (cr-matcher.c:246-251,312-338) depend on the
 * parse; they run in the sequential token loop of the coding kernel together with the PPM main
 * stream and the three u16-model side streams (models in LDS, one DPP sum per cumulative count).
 */
#ifndef CRGPU_ROX_H
#define CRGPU_ROX_H

#include "crgpu_ppm.h"
#include "crgpu_lzp.h"

#define CR_ROX_HEADER   32u
#define CR_ROX_NEAR_MIN 6u            /* match_min_near, cr-matcher.c:36 */
#define CR_ROX_MAX      255u          /* match_max */
#define CR_ROX_TAIL     1024u         /* cr-coder.c:136 */
#define CR_ROX_NONE     0xFFFFFFFFu
#define CR_ROX_LIMIT    40u           /* match_limit default, cr-matcher.c:39 */

/* side-stream models: 0 = length, 1..6 = distance digits, 7 = short distance (cr-coder.c:55-60) */
#define CR_SIDE_LEN  0
#define CR_SIDE_POS  1
#define CR_SIDE_SPOS 7

struct CrRoxShared {
    uint32_t hist[256];
    uint16_t f[8][256];
    uint32_t tot[8];
};

/* per-position results of the matching kernels */
struct CrRoxTables {
    uint32_t* prev;      /* u32[n] */
    uint32_t* nprev;     /* u32[n] */
    uint32_t* ml_pos;    /* u32[n] */
    uint8_t*  ml_len;    /* u8[n]  */
    uint8_t*  nl_len;    /* u8[n]  */
    uint8_t*  m0_len;    /* u8[n]: flexible parsing only — length of the plain match() at p, before the cut */
    uint32_t* cls_last;  /* u32[20 * classes], value = position + 1 */
    uint32_t* near_last; /* u32[65536] */
};

CR_DEV uint32_t cr_rox_mix(const uint8_t* s, uint32_t k) {            /* cr-matcher.c:45-53,203-211 */
    uint32_t h = 0;
    for (uint32_t i = 0; i < k; i++) h = (h * 123456791u) ^ s[i];
    return h;
}

/* reset_models, cr-coder.c:100-113 */
CR_DEV void cr_side_reset(CrRoxShared& sh) {
    const uint32_t lane = cr_lane();
    for (uint32_t k = lane; k < 256u; k += CRGPU_WAVE) {
        sh.f[CR_SIDE_LEN][k] = (k == 0u || k >= CR_ROX_NEAR_MIN) ? 1 : 0;
        sh.f[CR_SIDE_POS + 0][k] = (k % 8u == 0u) ? 1 : 0;
        sh.f[CR_SIDE_POS + 1][k] = 1;
        sh.f[CR_SIDE_POS + 2][k] = sh.f[CR_SIDE_POS + 3][k] = sh.f[CR_SIDE_POS + 4][k] = (k < 128u) ? 1 : 0;
        sh.f[CR_SIDE_POS + 5][k] = 1;
        sh.f[CR_SIDE_SPOS][k] = 1;
    }
    if (lane == 0) {
        sh.tot[CR_SIDE_LEN] = 251; sh.tot[CR_SIDE_POS + 0] = 32; sh.tot[CR_SIDE_POS + 1] = 256;
        sh.tot[CR_SIDE_POS + 2] = sh.tot[CR_SIDE_POS + 3] = sh.tot[CR_SIDE_POS + 4] = 128;
        sh.tot[CR_SIDE_POS + 5] = 256; sh.tot[CR_SIDE_SPOS] = 256;
    }
    cr_wave_sync();
}

/* persist mode: the side models live in LDS, so they are parked in the arena between calls */
CR_DEV void cr_side_park(const CrRoxShared& sh, uint8_t* keep) {
    const uint32_t lane = cr_lane();
    uint16_t* f = reinterpret_cast<uint16_t*>(keep);
    for (uint32_t i = lane; i < 8u * 256u; i += CRGPU_WAVE) f[i] = sh.f[i >> 8][i & 255u];
    if (lane < 8u) reinterpret_cast<uint32_t*>(keep + 4096)[lane] = sh.tot[lane];
}
CR_DEV void cr_side_unpark(CrRoxShared& sh, const uint8_t* keep) {
    const uint32_t lane = cr_lane();
    const uint16_t* f = reinterpret_cast<const uint16_t*>(keep);
    for (uint32_t i = lane; i < 8u * 256u; i += CRGPU_WAVE) sh.f[i >> 8][i & 255u] = f[i];
    if (lane < 8u) sh.tot[lane] = reinterpret_cast<const uint32_t*>(keep + 4096)[lane];
    cr_wave_sync();
}

/* model_update, cr-model.c:56-78 */
CR_DEV void cr_side_bump(CrRoxShared& sh, uint32_t m, uint32_t sym, uint32_t inc) {
    const uint32_t lane = cr_lane();
    uint32_t tot = (sh.tot[m] + inc) & 0xffffu;
    if (lane == 0) sh.f[m][sym] = (uint16_t)(sh.f[m][sym] + inc);
    cr_wave_sync();
    if (tot > 32000u) {
        uint32_t s = 0;
        for (uint32_t j = 0; j < 4u; j++) {
            uint32_t v = ((uint32_t)sh.f[m][lane * 4u + j] + 1u) >> 1;
            sh.f[m][lane * 4u + j] = (uint16_t)v;
            s += v;
        }
        tot = cr_sum(s);
    }
    if (lane == 0) sh.tot[m] = tot;
    cr_wave_sync();
}

/* M_my_enc_, cr-model.h:58-64 */
CR_DEV void cr_side_encode(CrRoxShared& sh, uint32_t m, uint32_t sym, uint32_t inc, CrRc& rc, CrSink& out) {
    const uint32_t lane = cr_lane();
    uint32_t mine = 0;
    for (uint32_t j = 0; j < 4u; j++) if (lane * 4u + j < sym) mine += sh.f[m][lane * 4u + j];
    const uint32_t below = cr_sum(mine);
    cr_rc_pin(rc);
    out.n = cr_uni(out.n);
    cr_rc_encode(rc, below, cr_uni(sh.f[m][sym]), cr_uni(sh.tot[m]), out);
    if (inc) cr_side_bump(sh, m, sym, inc);
}

/* M_my_dec_, cr-model.h:66-74 with model_get_decode_symbol, cr-model.c:98-115 */
CR_DEV uint32_t cr_side_decode(CrRoxShared& sh, uint32_t m, uint32_t inc, CrRc& rc, CrSource& in) {
    const uint32_t lane = cr_lane();
    uint32_t f0 = sh.f[m][lane * 4u], f1 = sh.f[m][lane * 4u + 1u], f2 = sh.f[m][lane * 4u + 2u], f3 = sh.f[m][lane * 4u + 3u];
    uint32_t mine = f0 + f1 + f2 + f3;
    uint32_t incl = cr_scan_incl(mine);
    cr_rc_pin(rc);
    in.pos = cr_uni(in.pos); in.base = cr_uni(in.base);
    const uint32_t target = cr_rc_dec_target(rc, cr_uni(sh.tot[m]));
    u64 owner = cr_ballot(incl - mine <= target && target < incl);
    uint32_t sym = 255, lower = 0, frq = 1;
    if (owner) {
        uint32_t ol = (uint32_t)__builtin_ctzll(owner);
        uint32_t before = cr_lane_get(incl - mine, ol);
        uint32_t j = cr_pick_in_word(cr_lane_get(f0, ol), cr_lane_get(f1, ol), cr_lane_get(f2, ol), cr_lane_get(f3, ol), before, target, lower);
        sym = ol * 4u + j;
        frq = cr_uni(sh.f[m][sym]);
    }
    cr_rc_dec_consume(rc, lower, frq, in);
    if (inc) cr_side_bump(sh, m, sym, inc);
    return sym;
}

/* ------------------------------------------------------------------ matching, position-parallel */

/* agreement length of d[a..] and d[b..] capped at 255, one lane */
CR_DEV uint32_t cr_rox_run(const uint8_t* d, uint32_t a, uint32_t b) { return cr_common_len(d, a, b); }

/* match(), cr-matcher.c:156-201, one lane. `want` is the minimum useful length, `budget` the
 * number of chain nodes to look at, `eager` the lazy-mode early return. */
CR_DEV void cr_rox_chain_search(const uint8_t* d, const uint32_t* prev, uint32_t pos, uint32_t want, uint32_t budget,
                                uint32_t eager, uint32_t& best_pos, uint32_t& best_len) {
    best_pos = 0; best_len = want - 1u;
    uint32_t at = prev[pos];
    for (uint32_t i = 0; i < budget && at != CR_ROX_NONE; i++) {
        /* the reference extends from best_len and then memcmp()s the first best_len bytes: both hold
         * exactly when the full agreement length reaches best_len */
        if (d[at + best_len] != d[pos + best_len]) { at = prev[at]; continue; }     /* agreement <= best_len: cannot win */
        uint32_t full = cr_rox_run(d, at, pos);
        uint32_t far = pos - at, cur = pos - best_pos, toll = 0;
        toll += (far >> 20) > cur ? 1u : 0u;
        toll += (far >> 12) > cur ? 1u : 0u;
        toll += (far >> 6) > cur ? 1u : 0u;
        if (full >= best_len && full > best_len + toll) {
            best_pos = at; best_len = full;
            if ((eager && eager < best_pos) || best_len == CR_ROX_MAX) return;
        }
        at = prev[at];
    }
    if (best_len < want) { best_pos = CR_ROX_NONE; best_len = 1u; }
}

/* lazy parsing part of matcher_lookup (cr-matcher.c:290-310) for position p, one lane */
CR_DEV void cr_rox_long_match(const uint8_t* d, const uint32_t* prev, uint32_t p, uint32_t long_min, uint32_t lim,
                              uint32_t& mpos, uint32_t& mlen) {
    cr_rox_chain_search(d, prev, p, long_min, lim, 0u, mpos, mlen);
    if (mlen < long_min) return;
    uint32_t qp, ql;
    cr_rox_chain_search(d, prev, p + 1u, mlen + 1u, lim / 4u, 1u, qp, ql);
    bool defer = ql > mlen + (qp < mpos ? 1u : 0u);
    if (!defer) { cr_rox_chain_search(d, prev, p + 2u, mlen + 1u, lim / 8u, 1u, qp, ql); defer = ql > 1u; }
    if (!defer) { cr_rox_chain_search(d, prev, p + 3u, mlen + 2u, lim / 8u, 1u, qp, ql); defer = ql > 1u; }
    if (!defer) { cr_rox_chain_search(d, prev, p + 4u, mlen + 2u, lim / 8u, 1u, qp, ql); defer = ql > 1u; }
    if (!defer) { cr_rox_chain_search(d, prev, p + 5u, mlen + 2u, lim / 8u, 1u, qp, ql); defer = ql > 1u; }
    if (!defer) { cr_rox_chain_search(d, prev, p + 6u, mlen + 3u, lim / 8u, 1u, qp, ql); defer = ql > 1u; }
    if (defer) { mpos = CR_ROX_NONE; mlen = 1u; }
}

/* one wave: chain links for every position with p + 255 < n (cr-matcher.c:89-148) */
CR_DEV void cr_rox_sweep_chains(const uint8_t* d, uint32_t n, uint32_t long_min, const CrRoxTables& T) {
    const uint32_t lane = cr_lane();
    const uint32_t classes = 20u + n / 25u;
    const uint32_t lim = n > CR_ROX_MAX ? n - CR_ROX_MAX : 0u;
    for (uint32_t p0 = 0; p0 < lim; p0 += CRGPU_WAVE) {
        const uint32_t p = p0 + lane;
        const bool act = p < lim;
        uint32_t cls = 0;
        if (act) cls = (((uint32_t)d[p] + d[p + 1]) % 20u) * classes + cr_rox_mix(d + p, long_min) % classes;
        /* the wave owns the class heads: the first lane of a class reads its head, the last one writes it (no atomics) */
        const u64 same = cr_same_key_mask<24>(cls, act), lower = same & ((1ull << lane) - 1ull);
        if (act) {
            uint32_t c = CR_ROX_NONE;
            if (lower) c = p0 + 63u - (uint32_t)__builtin_clzll(lower);
            else { uint32_t v = cr_ld32(T.cls_last + cls); if (v) c = v - 1u; }
            T.prev[p] = c;
            if ((same >> lane) >> 1 == 0ull) cr_st32(T.cls_last + cls, p + 1u);
        }
    }
}

/* one wave: short-cache content seen by every position (cr-matcher.c:213-216,319-331) */
CR_DEV void cr_rox_sweep_near(const uint8_t* d, uint32_t n, const CrRoxTables& T) {
    const uint32_t lane = cr_lane();
    const uint32_t lim = n > CR_ROX_TAIL ? n - CR_ROX_TAIL : 0u;
    for (uint32_t p0 = 0; p0 < lim; p0 += CRGPU_WAVE) {
        const uint32_t p = p0 + lane;
        const bool act = p < lim;
        uint32_t key = 0;
        if (act) key = cr_rox_mix(d + p, CR_ROX_NEAR_MIN) & 0xffffu;
        const u64 same = cr_same_key_mask<16>(key, act), lower = same & ((1ull << lane) - 1ull);
        if (act) {
            T.nprev[p] = lower ? p0 + 63u - (uint32_t)__builtin_clzll(lower) : cr_ld32(T.near_last + key);   /* an untouched slot reads 0 */
            if ((same >> lane) >> 1 == 0ull) cr_st32(T.near_last + key, p);
        }
    }
}

/* fast_log2, cr-matcher.c:218-235 */
CR_DEV uint32_t cr_rox_ilog2(uint32_t x) { return 31u - (uint32_t)__builtin_clz(x | 1u); }
/* M_price, cr-matcher.c:269-271 (both distances are measured from the position being parsed) */
CR_DEV uint32_t cr_rox_flex_price(uint32_t pos, uint32_t from, uint32_t len, uint32_t long_min) {
    return len >= long_min ? (len - 1u) * 3u - (cr_rox_ilog2(pos - from) * 4u) / 5u : 9u;
}
/* -f, cr-matcher.c:253-289, all threads of the workgroup: pass 1 the plain match() of every position that a
 * cut can look at, pass 2 the cut — the length that leaves the best-priced pair "this match, then whatever
 * starts right behind it". ml_len[p] == 1 afterwards means "no long match here". */
CR_DEV void cr_rox_flex_all(const uint8_t* d, uint32_t n, uint32_t long_min, uint32_t chain_limit, const CrRoxTables& T) {
    const uint32_t lim = n > CR_ROX_TAIL ? n - CR_ROX_TAIL : 0u;
    const uint32_t lim0 = lim ? lim + CR_ROX_MAX + 1u : 0u;            /* < n - 255: every chain link read is valid */
    for (uint32_t p = threadIdx.x; p < lim0; p += blockDim.x) {
        uint32_t mp, ml;
        cr_rox_chain_search(d, T.prev, p, long_min, chain_limit, 0u, mp, ml);
        T.ml_pos[p] = mp;
        T.m0_len[p] = (uint8_t)ml;
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
    __syncthreads();
    for (uint32_t p = threadIdx.x; p < lim; p += blockDim.x) {
        const uint32_t mp = T.ml_pos[p], whole = T.m0_len[p];
        uint32_t keep = whole;
        if (whole >= long_min) {
            uint32_t best = cr_rox_flex_price(p, mp, whole, long_min) + cr_rox_flex_price(p, T.ml_pos[p + whole], T.m0_len[p + whole], long_min);
            for (uint32_t i = whole - 1u; i >= 1u; i--) {
                const uint32_t v = cr_rox_flex_price(p, mp, i, long_min) + cr_rox_flex_price(p, T.ml_pos[p + i], T.m0_len[p + i], long_min);
                if (best < v) { keep = i; best = v; }
            }
            if (keep < long_min) keep = 1u;
        }
        T.ml_len[p] = (uint8_t)keep;
        uint32_t q = T.nprev[p], nl = 0;
        if (q < p && q + 256u > p) nl = cr_rox_run(d, q, p);
        T.nl_len[p] = (uint8_t)nl;
    }
}

/* all threads of the workgroup: long and near matches for every position that can start a token */
CR_DEV void cr_rox_match_all(const uint8_t* d, uint32_t n, uint32_t long_min, uint32_t chain_limit, const CrRoxTables& T) {
    const uint32_t lim = n > CR_ROX_TAIL ? n - CR_ROX_TAIL : 0u;
    for (uint32_t p = threadIdx.x; p < lim; p += blockDim.x) {
        uint32_t mp, ml;
        cr_rox_long_match(d, T.prev, p, long_min, chain_limit, mp, ml);
        T.ml_pos[p] = mp;
        T.ml_len[p] = (uint8_t)ml;
        uint32_t q = T.nprev[p], nl = 0;
        if (q < p && q + 256u > p) nl = cr_rox_run(d, q, p);
        T.nl_len[p] = (uint8_t)nl;
    }
}

/* ------------------------------------------------------------------ wave-parallel helpers of the coder */

/* agreement length of d[a..] and d[b..] (a < b), capped at 255, computed by the whole wave */
CR_DEV uint32_t cr_rox_run_wave(const uint8_t* d, uint32_t a, uint32_t b) {
    const uint32_t lane = cr_lane();
    uint32_t x = *reinterpret_cast<const cr_u32u*>(d + a + lane * 4u) ^ *reinterpret_cast<const cr_u32u*>(d + b + lane * 4u);
    u64 diff = cr_ballot(x != 0u);
    if (!diff) return CR_ROX_MAX;
    uint32_t l = (uint32_t)__builtin_ctzll(diff);
    uint32_t xx = cr_lane_get(x, l);
    uint32_t len = l * 4u + ((uint32_t)__builtin_ctz(xx) >> 3);
    return len < CR_ROX_MAX ? len : CR_ROX_MAX;
}

CR_DEV void cr_rox_store_raw(const uint8_t* src, uint32_t n, uint8_t* dst) {      /* cr-coder.c:309-313 */
    const uint32_t lane = cr_lane();
    if (lane < CR_ROX_HEADER) dst[lane] = 0;
    for (uint32_t i = lane; i < n; i += CRGPU_WAVE) dst[CR_ROX_HEADER + i] = src[i];
}

/* lzencode, cr-coder.c:153-318 — token loop + coding; the tables T were filled by k_rox_match */
CR_DEV uint32_t cr_rox_encode_block(const uint8_t* src, uint32_t n, uint8_t* dst, const CrRoxTables& T, uint8_t* side,
                                    u64 side_stride, uint8_t* arena, const CrArenaLayout& L, uint32_t fresh, uint32_t persist,
                                    CrRoxShared& sh) {
    const uint32_t lane = cr_lane();
    const uint32_t long_min = 10u + (n > 16777216u ? 1u : 0u);           /* cr-coder.c:192 */
    const uint32_t esc = cr_pick_escape(src, n, sh.hist);
    CrPpm m;
    cr_ppm_attach(m, arena, L, persist ? L.cap_o3 : cr_log2_ceil_pow2(2u * n, 1024u, L.cap_o3));
    if (fresh) { cr_side_reset(sh); cr_ppm_reset(m); }
    else { cr_side_unpark(sh, arena + L.off_keep); cr_ppm_resume(m); }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
    cr_wave_sync();

    CrSink s_main, s_spos, s_pos, s_len;
    s_main.dst = dst + CR_ROX_HEADER; s_main.n = 0;
    s_spos.dst = side; s_spos.n = 0;
    s_pos.dst = side + side_stride; s_pos.n = 0;
    s_len.dst = side + 2u * side_stride; s_len.n = 0;
    CrRc rc_main, rc_spos, rc_pos, rc_len;
    cr_rc_init(rc_main); cr_rc_init(rc_spos); cr_rc_init(rc_pos); cr_rc_init(rc_len);
    CrFetch F; F.valid = 0; F.ctx = 0; F.with_row = 0;
    m.defer = 1;
#ifdef CRGPU_PROF
    CrProf prof; prof.last = 0;
    for (int i = 0; i < 8; i++) prof.acc[i] = 0;
#endif
    CrWindow win;
    cr_window_init(win, src, n, 0);

    uint32_t pos = 0, repeat = 0, prev_dist = 0, n_spos = 0, n_pos = 0, n_len = 0;
    bool stored = false;
    while (pos < n) {                                                    /* cr-coder.c:213-276 */
        uint32_t from = CR_ROX_NONE, len = 1;
        if (pos + CR_ROX_TAIL < n) {                                     /* matcher_lookup, cr-matcher.c:237-340 */
            uint32_t mp = cr_uni(T.ml_pos[pos]), ml = cr_uni(T.ml_len[pos]);
            if (ml < 2u) mp = CR_ROX_NONE;                               /* (flexible parsing keeps the uncut position in ml_pos) */
            if (mp != CR_ROX_NONE) {
                uint32_t rp = pos - repeat, rl = 0;                      /* the previous distance again (:246-251) */
                if (rp < pos) rl = cr_rox_run_wave(src, rp, pos);
                if (ml < rl + 3u + (mp + 64u < pos ? 1u : 0u) + (mp + 4096u < pos ? 1u : 0u) + (mp + 1048576u < pos ? 1u : 0u)) { mp = rp; ml = rl; }
            }
            if (ml < CR_ROX_NEAR_MIN) { mp = cr_uni(T.nprev[pos]); ml = cr_uni(T.nl_len[pos]); }      /* (:319-331) */
            if (!(ml < CR_ROX_NEAR_MIN || (ml < long_min && mp + 256u <= pos))) {                     /* (:333-338) */
                from = mp; len = ml; repeat = pos - mp;
            }
        }
        if (from != CR_ROX_NONE) {
            /* context after the match bytes (the escape byte itself is not pushed in this codec) */
            uint32_t after = m.ctx;
            if (len >= 4u) after = cr_uni(__builtin_bswap32(*reinterpret_cast<const cr_u32u*>(src + pos + len - 4u)));
            else for (uint32_t i = 0; i < len; i++) after = (after << 8) | cr_window_at(win, pos + i);
            cr_ppm_encode(m, rc_main, esc, s_main, F, after, 0u CR_PROF_PASS);
            uint32_t dist = pos - from;
            if (dist == prev_dist) dist = 0;                             /* cr-coder.c:232-234 */
            cr_side_encode(sh, CR_SIDE_LEN, len, 30u, rc_len, s_len); n_len++;
            if (len < long_min) {
                cr_side_encode(sh, CR_SIDE_SPOS, dist, 1u, rc_spos, s_spos); n_spos++;
            } else {                                                     /* cr-coder.c:243-258 */
                uint32_t j = dist * 8u, i = 0;
                while (j >= 128u && i < 2u) { cr_side_encode(sh, CR_SIDE_POS + i, j % 128u + 128u, 1u << (2u * i), rc_pos, s_pos); i++; j /= 128u; }
                if (i >= 2u) while (j >= 64u && i < 5u) { cr_side_encode(sh, CR_SIDE_POS + i, j % 64u + 64u, 1u << (2u * i), rc_pos, s_pos); i++; j /= 64u; }
                cr_side_encode(sh, CR_SIDE_POS + i, j, 1u << (2u * i), rc_pos, s_pos);
                n_pos++;
            }
            prev_dist = dist;
            m.ctx = after;
        } else {
            const uint32_t c = cr_window_at(win, pos);
            cr_ppm_encode(m, rc_main, c, s_main, F, (m.ctx << 8) | c, 0u CR_PROF_PASS);
            if (c == esc) { cr_side_encode(sh, CR_SIDE_LEN, 0u, 30u, rc_len, s_len); n_len++; }
            cr_ppm_push(m, c);
        }
        pos += len;
        if (CR_ROX_HEADER + s_main.n >= n) { stored = true; break; }     /* cr-coder.c:273-275 */
    }
    cr_node_writeback(m);
    if (persist) { cr_ppm_suspend(m); cr_side_park(sh, arena + L.off_keep); }
    if (stored) {
        cr_wave_sync();
        cr_rox_store_raw(src, n, dst);
        return CR_ROX_HEADER + n;
    }
    cr_rc_pin(rc_main); cr_rc_flush(rc_main, s_main);
    cr_rc_pin(rc_spos); cr_rc_flush(rc_spos, s_spos);
    cr_rc_pin(rc_pos); cr_rc_flush(rc_pos, s_pos);
    cr_rc_pin(rc_len); cr_rc_flush(rc_len, s_len);
    cr_wave_sync();
    const uint32_t o_spos = CR_ROX_HEADER + s_main.n, o_pos = o_spos + s_spos.n, o_len = o_pos + s_pos.n;
    for (uint32_t i = lane; i < s_spos.n; i += CRGPU_WAVE) dst[o_spos + i] = s_spos.dst[i];
    for (uint32_t i = lane; i < s_pos.n; i += CRGPU_WAVE) dst[o_pos + i] = s_pos.dst[i];
    for (uint32_t i = lane; i < s_len.n; i += CRGPU_WAVE) dst[o_len + i] = s_len.dst[i];
    if (lane < CR_ROX_HEADER) {                                          /* cr-coder.c:289-297 */
        uint32_t word = lane >> 2, v = 0;
        const uint32_t fields[8] = {1u | (long_min << 8) | (esc << 16), n, n_spos, n_pos, n_len, o_spos, o_pos, o_len};
        v = fields[word];
        dst[lane] = (uint8_t)(v >> (8u * (lane & 3u)));
    }
    return o_len + s_len.n;
}

/* lzdecode, cr-coder.c:390-526 */
CR_DEV uint32_t cr_rox_decode_block(const uint8_t* src, uint32_t n, uint8_t* dst, uint32_t cap, uint8_t* arena,
                                    const CrArenaLayout& L, uint32_t fresh, uint32_t persist, CrRoxShared& sh) {
    const uint32_t lane = cr_lane();
    if (n < CR_ROX_HEADER) return 0xFFFFFFFFu;
    if (src[0] == 0) {
        uint32_t raw = n - CR_ROX_HEADER;
        if (raw > cap) return 0xFFFFFFFFu;
        for (uint32_t i = lane; i < raw; i += CRGPU_WAVE) dst[i] = src[CR_ROX_HEADER + i];
        return raw;
    }
    const uint32_t long_min = src[1], esc = src[2];
    uint32_t hw[7];
    for (int k = 0; k < 7; k++) hw[k] = (uint32_t)src[4 + 4 * k] | ((uint32_t)src[5 + 4 * k] << 8) | ((uint32_t)src[6 + 4 * k] << 16) | ((uint32_t)src[7 + 4 * k] << 24);
    const uint32_t total = hw[0], o_spos = hw[4], o_pos = hw[5], o_len = hw[6];
    if (total > cap || total > L.max_block || o_spos < CR_ROX_HEADER || o_spos > o_pos || o_pos > o_len || o_len > n) return 0xFFFFFFFFu;
    CrPpm m;
    cr_ppm_attach(m, arena, L, persist ? L.cap_o3 : cr_log2_ceil_pow2(2u * total, 1024u, L.cap_o3));
    if (fresh) { cr_side_reset(sh); cr_ppm_reset(m); }
    else { cr_side_unpark(sh, arena + L.off_keep); cr_ppm_resume(m); }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
    cr_wave_sync();
    CrSource in_main, in_spos, in_pos, in_len;
    cr_source_init(in_main, src + CR_ROX_HEADER, n - CR_ROX_HEADER);
    cr_source_init(in_spos, src + o_spos, n - o_spos);
    cr_source_init(in_pos, src + o_pos, n - o_pos);
    cr_source_init(in_len, src + o_len, n - o_len);
    CrRc rc_main, rc_spos, rc_pos, rc_len;
    cr_rc_dec_init(rc_main, in_main); cr_rc_dec_init(rc_spos, in_spos);
    cr_rc_dec_init(rc_pos, in_pos); cr_rc_dec_init(rc_len, in_len);
    CrFetch F; F.valid = 0; F.ctx = 0; F.with_row = 1;
#ifdef CRGPU_PROF
    CrProf prof; prof.last = 0;
    for (int i = 0; i < 8; i++) prof.acc[i] = 0;
#endif
    uint32_t have = 0, prev_dist = 0;
    while (have < total) {                                               /* cr-coder.c:459-523 */
        const uint32_t s = cr_ppm_decode(m, rc_main, in_main, F CR_PROF_PASS);
        uint32_t lit = s, len = 1, dist = 0;
        if (s == esc) {
            len = cr_side_decode(sh, CR_SIDE_LEN, 30u, rc_len, in_len);
            if (len == 0u) {
                len = 1; lit = esc;
            } else if (len < long_min) {
                dist = cr_side_decode(sh, CR_SIDE_SPOS, 1u, rc_spos, in_spos);
            } else {                                                     /* cr-coder.c:347-368 */
                uint32_t v = 0, sym = 0, j = 0;
                while (j < 2u && (sym = cr_side_decode(sh, CR_SIDE_POS + j, 1u << (2u * j), rc_pos, in_pos)) >= 128u) { v += (sym - 128u) << (7u * j); j++; }
                if (j < 2u) {
                    dist = (v + (sym << (7u * j))) / 8u;
                } else {
                    while (j < 5u && (sym = cr_side_decode(sh, CR_SIDE_POS + j, 1u << (2u * j), rc_pos, in_pos)) >= 64u) { v += (sym - 64u) << (6u * j + 2u); j++; }
                    dist = (v + (sym << (6u * j + 2u))) / 8u;
                }
            }
        }
        if (len > 1u) {                                                  /* cr-coder.c:503-514 */
            const uint32_t dd = dist > 0u ? dist : prev_dist;
            if (dd == 0u || dd > have || have + len > total || have + len > cap) return 0xFFFFFFFFu;
            prev_dist = dd;
            const uint32_t from = have - dd;
            cr_wave_sync();
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
                m.ctx = (cr_lane_get(mine, l3 - 3u) << 24) | (cr_lane_get(mine, l3 - 2u) << 16) | (cr_lane_get(mine, l3 - 1u) << 8) | cr_lane_get(mine, l3);
            } else {
                cr_wave_sync();
                uint32_t k = len < 4u ? len : 4u;
                for (uint32_t i = len - k; i < len; i++) cr_ppm_push(m, cr_uni(dst[have + i]));
            }
        } else {
            if (have >= cap) return 0xFFFFFFFFu;
            if (lane == 0) dst[have] = (uint8_t)lit;
            cr_ppm_push(m, lit);
        }
        have += len;
    }
    cr_node_writeback(m);
    if (persist) { cr_ppm_suspend(m); cr_side_park(sh, arena + L.off_keep); }
    return have;
}

#endif

/*
 * comprox_amd/csrc/crgpu_rolz.h — comprolz block codec (lzencode / lzdecode of src/rolzmain) on gfx950.
 *
 * Reference: /root/reference/src/rolzmain/cr-coder.c:78-97,138-258,283-379 and cr-matcher.c:37-197
 * (lazy parsing by default, the -f "flexible parsing" of cr-matcher.c:143-167 on request). Block layout (cr-coder.c:63-71, 16 bytes):
 * [0] first byte of the block, [1] coded flag, [2] esc, [3] 0, then u32 LE original size, number of
 * side-stream codes, offset of the side stream; body = main PPM stream, then the side stream (match lengths
 * and ranks through two u16 models, cr-model.c).
 *
 * ROLZ names a match by its RANK among the 64 newest positions that followed the same hashed 3-byte
 * context (or, failing that, the 16 newest behind the same byte). The reference keeps 262 144 rings of 64
 * positions, 85 MB re-initialised per block; a block of n bytes feeds every position once, so the same
 * information is two link arrays — the previous position fed to the same ring / the same row — plus the
 * newest entry of every ring (1 MB, per resident workgroup) and row (LDS).
 *
 * Encoding: what the reference looks up at a position only depends on the positions fed before it, and
 * every earlier position has been fed by then — so the parse result at every position is a pure function
 * of the data and is computed position-parallel (k_rolz_match: two waves build the links, then all threads
 * run matcher_lookup incl. its lazy-evaluation veto); the sequential part is the token walk with the PPM
 * main stream and the side stream (k_rolz_encode, one wave per block).
 * Decoding (k_rolz_decode, one wave per block): positions are fed in batches of up to 64 at match tokens,
 * the only place the rings are read (same scheme as the LZP tables of the comprop decoder).
 */
#ifndef CRGPU_ROLZ_H
#define CRGPU_ROLZ_H

#include "crgpu_rox.h"

#define CR_ROLZ_HEADER  16u
#define CR_ROLZ_BUCKETS 262144u       /* M_rolz_buckets, cr-matcher.h:37 */
#define CR_ROLZ_RING    64u           /* M_rolz_indices */
#define CR_ROLZ_ROW     16u           /* M_rolz_indices_short */
#define CR_ROLZ_MIN     5u            /* M_rolz_minlength */
#define CR_ROLZ_MAX     255u          /* M_rolz_maxlength */
#define CR_ROLZ_TAIL    1024u         /* cr-coder.c:121 */
#define CR_ROLZ_WARM    16u           /* cr-matcher.c:68,148 */
#define CR_ROLZ_NONE    0xFFFFFFFFu
#define CR_ROLZ_M_LEN   0             /* side models in CrRoxShared: 0 = length, 1 = rank */
#define CR_ROLZ_M_IDX   1

struct CrRolzTables {
    uint32_t* ring_prev;   /* u32[n]: previous position fed to the same ring */
    uint16_t* ring16;      /* encoder, blocks below 65 535 bytes: the same links as u16 (0xffff = none) — half the footprint of the
                            * array the ring searches chase through; nullptr otherwise */
    uint32_t* row_prev;    /* u32[n]: previous position fed to the same row */
    uint8_t*  rank;        /* u8[n]: parse result, 0xff = literal */
    uint8_t*  len;         /* u8[n] */
    uint16_t* raw16;       /* encoder, u16[n]: the plain lookup of every position, rank (0xff = none) | length << 8 */
    uint16_t* row16;       /* encoder, u16[n]: the row search of the positions whose plain lookup fell short (k_rolz_rings_lds64), same packing */
    uint32_t* ring_head;   /* u32[262144], newest position + 1 (0 = empty); per resident workgroup */
};

/* M_rolz_hash_ctx, cr-matcher.c:37-41: the three (four) bytes ending at x */
CR_DEV uint32_t cr_rolz_hash(uint32_t b1, uint32_t b2, uint32_t b3, uint32_t b4, bool ctx4) {
    uint32_t h = b1 * 1313131u + b2 * 13131u + b3 * 131u;
    if (ctx4) h += b4;
    return h % CR_ROLZ_BUCKETS;
}
/* ring / row a position joins when it is fed (matcher_update, cr-matcher.c:66-84): the context / byte in
 * front of it — except position 16, which joins ring 0 and row 0 because nothing moved the matcher before */
CR_DEV uint32_t cr_rolz_ring_of(const uint8_t* d, uint32_t p, bool ctx4) {
    if (p <= CR_ROLZ_WARM) return 0u;
    return cr_rolz_hash(d[p - 1], d[p - 2], d[p - 3], d[p - 4], ctx4);
}
CR_DEV uint32_t cr_rolz_row_of(const uint8_t* d, uint32_t p) { return p <= CR_ROLZ_WARM ? 0u : d[p - 1]; }

/* reset_models, cr-coder.c:78-97 */
CR_DEV void cr_rolz_side_reset(CrRoxShared& sh) {
    const uint32_t lane = cr_lane();
    for (uint32_t k = lane; k < 256u; k += CRGPU_WAVE) {
        sh.f[CR_ROLZ_M_LEN][k] = (k == 0u || k >= CR_ROLZ_MIN) ? 1 : 0;
        sh.f[CR_ROLZ_M_IDX][k] = (k < CR_ROLZ_RING + CR_ROLZ_ROW) ? 1 : 0;
    }
    if (lane == 0) { sh.tot[CR_ROLZ_M_LEN] = 252; sh.tot[CR_ROLZ_M_IDX] = CR_ROLZ_RING + CR_ROLZ_ROW; }
    cr_wave_sync();
}

/* ------------------------------------------------------------------ k_rolz_match, phase A: the links */

CR_DEV uint32_t cr_rolz_link(const CrRolzTables& T, uint32_t p) {
    if (T.ring16) { const uint32_t v = T.ring16[p]; return v == 0xffffu ? CR_ROLZ_NONE : v; }
    return T.ring_prev[p];
}
/* one wave: ring_prev[p] for p in [16, limit) */
CR_DEV void cr_rolz_sweep_rings(const uint8_t* d, uint32_t limit, bool ctx4, const CrRolzTables& T) {
    const uint32_t lane = cr_lane();
    for (uint32_t p0 = CR_ROLZ_WARM; p0 < limit; p0 += CRGPU_WAVE) {
        const uint32_t p = p0 + lane;
        const bool act = p < limit;
        const uint32_t key = act ? cr_rolz_ring_of(d, p, ctx4) : 0u;
        /* the wave owns the ring heads: the first lane of a ring reads its head, the last one writes it (no atomics) */
        const u64 same = cr_same_key_mask<18>(key, act), lower = same & ((1ull << lane) - 1ull);
        if (act) {
            uint32_t c = CR_ROLZ_NONE;
            if (lower) c = p0 + 63u - (uint32_t)__builtin_clzll(lower);
            else { const uint32_t v = cr_ld32(T.ring_head + key); if (v) c = v - 1u; }
            if (T.ring16) T.ring16[p] = (uint16_t)c; else T.ring_prev[p] = c;
            if ((same >> lane) >> 1 == 0ull) cr_st32(T.ring_head + key, p + 1u);
        }
    }
}
/* one wave: row_prev[p]; the 256 row heads live in LDS */
CR_DEV void cr_rolz_sweep_rows(const uint8_t* d, uint32_t limit, const CrRolzTables& T, uint32_t* row_head) {
    const uint32_t lane = cr_lane();
    for (uint32_t k = lane; k < 256u; k += CRGPU_WAVE) row_head[k] = 0;
    cr_wave_sync();
    for (uint32_t p0 = CR_ROLZ_WARM; p0 < limit; p0 += CRGPU_WAVE) {
        const uint32_t p = p0 + lane;
        const bool act = p < limit;
        const uint32_t key = act ? cr_rolz_row_of(d, p) : 0u;
        const int q = cr_prev_same_bits<8>(key, act);
        if (act) {
            uint32_t c = CR_ROLZ_NONE;
            if (q >= 0) c = p0 + (uint32_t)q;
            else { const uint32_t v = row_head[key]; if (v) c = v - 1u; }
            T.row_prev[p] = c;
        }
        cr_wave_sync();
        if (act) atomicMax(row_head + key, p + 1u);
        cr_wave_sync();
    }
}

/* ------------------------------------------------------------------ phase B: matcher_lookup for every position */

/* match(), cr-matcher.c:93-124: ring entries from `start` on that were fed before `floor`, newest first;
 * the first strictly longer agreement wins; the ring's hash byte is the entry's first byte */
CR_DEV void cr_rolz_ring_search(const uint8_t* d, uint32_t pos, uint32_t start, uint32_t floor, const CrRolzTables& T,
                                uint32_t& rank, uint32_t& len) {
    rank = CR_ROLZ_NONE; len = CR_ROLZ_MIN - 1u;
    uint32_t q = start;
    while (q != CR_ROLZ_NONE && q >= floor) q = cr_rolz_link(T, q);
    const uint32_t first = d[pos];
    uint32_t beyond = d[pos + len];                    /* an entry can only be strictly longer if it also agrees at offset `len` */
    for (uint32_t i = 0; i < CR_ROLZ_RING && len < CR_ROLZ_MAX && q != CR_ROLZ_NONE; i++, q = cr_rolz_link(T, q)) {
        if (d[q] != first || d[q + len] != beyond) continue;
        const uint32_t j = cr_common_len(d, q, pos);
        if (j > len) { rank = i; len = j; beyond = d[pos + len]; }
    }
    if (len < CR_ROLZ_MIN) { rank = CR_ROLZ_NONE; len = 1; }
}
CR_DEV uint32_t cr_rolz_price(uint32_t rank, uint32_t len) {                  /* M_price, cr-matcher.c:150-152 */
    return len >= CR_ROLZ_MIN ? (len - 1u) * 3u * CR_ROLZ_RING - 3u * rank : 9u * CR_ROLZ_RING;
}
/* matcher_lookup, cr-matcher.c:126-197, one position per thread */
/* The look-ahead searches of lazy evaluation and of -f ask about position p + i with the ring as of before p
 * (cr-matcher.c:143-167,188-196 call match() without feeding). The ring links run from newer to older positions, so
 * unless the newest entry of p + i's ring lies in [p, p + i) the answer is the plain lookup of p + i. Pass 1 therefore
 * does the plain lookup of every position once (raw16), pass 2 reuses it wherever that test allows. */
CR_DEV void cr_rolz_ahead(const uint8_t* d, uint32_t at, uint32_t floor, const CrRolzTables& T, const uint16_t* raw16,
                          uint32_t& rank, uint32_t& len) {
    const uint32_t newest = cr_rolz_link(T, at);
    if (newest == CR_ROLZ_NONE || newest < floor) {
        const uint32_t v = raw16[at];
        rank = (v & 0xffu) == 0xffu ? CR_ROLZ_NONE : (v & 0xffu);
        len = v >> 8;
    } else {
        cr_rolz_ring_search(d, at, newest, floor, T, rank, len);
    }
}

CR_DEV void cr_rolz_find_all(const uint8_t* d, uint32_t n, uint32_t link_limit, bool ctx4, bool flexible, const CrRolzTables& T,
                             uint16_t* raw16 /* plain lookup of every position: rank (0xff = none) | length << 8 */, bool raw_done, bool rows_done, u64* st = nullptr) {
    /* raw_done / rows_done: k_rolz_rings_lds64 has left the plain lookups in raw16 / the row searches in T.row16 */
    const uint32_t limit = n - CR_ROLZ_TAIL;              /* positions with p + 1024 < n */
    (void)ctx4;
    if (!raw_done) for (uint32_t p = CR_ROLZ_WARM + threadIdx.x; p < link_limit; p += blockDim.x) {
        uint32_t rank, len;
        cr_rolz_ring_search(d, p, cr_rolz_link(T, p), p, T, rank, len);
        raw16[p] = (uint16_t)((rank == CR_ROLZ_NONE ? 0xffu : rank) | (len << 8));
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
    __syncthreads();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    if (st && threadIdx.x == 0) st[9] = wall_clock64();
    for (uint32_t p = CR_ROLZ_WARM + threadIdx.x; p < limit; p += blockDim.x) {
        const uint32_t rv = raw16[p];
        uint32_t rank = (rv & 0xffu) == 0xffu ? CR_ROLZ_NONE : (rv & 0xffu), len = rv >> 8;
        const bool fell_short = len < CR_ROLZ_MIN;
        if (flexible && !fell_short) {                    /* -f (:143-167): cut where "this match + what follows" prices best */
            uint32_t best = 0, keep = len;
            for (uint32_t i = len; i >= 1u; i--) {
                uint32_t r2, l2;
                cr_rolz_ahead(d, p + i, p, T, raw16, r2, l2);
                const uint32_t v = cr_rolz_price(rank, i) + cr_rolz_price(r2, l2);
                if (i == len) best = v;
                else if (v > best) { keep = i; best = v; }
            }
            len = keep;
            if (len < CR_ROLZ_MIN) { rank = CR_ROLZ_NONE; len = 1; }
        }
        if (fell_short && rows_done) {
            const uint32_t wv = T.row16[p];
            rank = (wv & 0xffu) == 0xffu ? CR_ROLZ_NONE : (wv & 0xffu); len = wv >> 8;
        } else if (fell_short) {                          /* the 16 newest positions behind the same byte (:171-186) */
            len = CR_ROLZ_MIN - 1u; rank = CR_ROLZ_NONE;
            uint32_t q = T.row_prev[p];
            for (uint32_t i = 0; i < CR_ROLZ_ROW; i++) {
                const uint32_t at = q == CR_ROLZ_NONE ? 0u : q;          /* the reference's rows are zero-filled */
                const uint32_t j = cr_common_len(d, at, p);
                if (j > len) { rank = CR_ROLZ_RING + i; len = j; }
                if (q != CR_ROLZ_NONE) q = T.row_prev[q];
            }
            if (len < CR_ROLZ_MIN) { rank = CR_ROLZ_NONE; len = 1; }
        }
        if ((!flexible || fell_short) && len > 1u) {      /* lazy evaluation (:188-196): looks ahead without feeding */
            const uint32_t mine = cr_rolz_price(rank, len);
            for (uint32_t i = 1; i < CR_ROLZ_MIN; i++) {
                uint32_t r2, l2;
                cr_rolz_ahead(d, p + i, p, T, raw16, r2, l2);
                if (cr_rolz_price(r2, l2) > mine + i * CR_ROLZ_RING) { rank = CR_ROLZ_NONE; len = 1; break; }
            }
        }
        T.rank[p] = (uint8_t)(rank == CR_ROLZ_NONE ? 0xffu : rank);
        T.len[p] = (uint8_t)len;
    }
}

/* ------------------------------------------------------------------ lzencode, cr-coder.c:138-258 */

CR_DEV void cr_rolz_store_raw(const uint8_t* src, uint32_t n, uint8_t* dst) {    /* cr-coder.c:247-257 */
    const uint32_t lane = cr_lane();
    if (lane < CR_ROLZ_HEADER) dst[lane] = 0;
    for (uint32_t i = lane; i < n; i += CRGPU_WAVE) dst[CR_ROLZ_HEADER + i] = src[i];
}

CR_DEV uint32_t cr_rolz_encode_block(const uint8_t* src, uint32_t n, uint8_t* dst, const CrRolzTables& T, uint8_t* side,
                                     uint8_t* arena, const CrArenaLayout& L, uint32_t fresh, uint32_t persist, CrRoxShared& sh) {
    const uint32_t lane = cr_lane();
    if (n == 0) { if (lane < CR_ROLZ_HEADER) dst[lane] = 0; return CR_ROLZ_HEADER; }   /* not defined by the reference: an empty stored block */
    const uint32_t esc = cr_pick_escape(src, n, sh.hist);
    CrPpm m;
    cr_ppm_attach(m, arena, L, persist ? L.cap_o3 : cr_log2_ceil_pow2(2u * n, 1024u, L.cap_o3));
    if (fresh) { cr_rolz_side_reset(sh); cr_ppm_reset(m); }
    else { cr_side_unpark(sh, arena + L.off_keep); cr_ppm_resume(m); }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
    cr_wave_sync();

    CrSink s_main, s_side;
    s_main.dst = dst + CR_ROLZ_HEADER; s_main.n = 0;
    s_side.dst = side; s_side.n = 0;
    CrRc rc_main, rc_side;
    cr_rc_init(rc_main); cr_rc_init(rc_side);
    CrFetch F; F.valid = 0; F.ctx = 0; F.with_row = 0;
    m.defer = 1;
#ifdef CRGPU_PROF
    CrProf prof; prof.last = 0;
    for (int i = 0; i < 8; i++) prof.acc[i] = 0;
#endif
    CrWindow win;
    cr_window_init(win, src, n, 0);

    uint32_t pos = 1, codes = 0;                                         /* the first byte travels in the header */
    bool stored = false;
    while (pos < n) {                                                    /* cr-coder.c:196-236 */
        uint32_t rank = 0xffu, len = 1;
        if (pos >= CR_ROLZ_WARM && pos + CR_ROLZ_TAIL < n) { rank = cr_uni(T.rank[pos]); len = cr_uni(T.len[pos]); }
        if (rank != 0xffu) {
            /* the escape byte is not pushed; the context afterwards is the match's last four bytes */
            const uint32_t after = cr_uni(__builtin_bswap32(*reinterpret_cast<const cr_u32u*>(src + pos + len - 4u)));
            cr_ppm_encode(m, rc_main, esc, s_main, F, after, 0u CR_PROF_PASS);
            cr_side_encode(sh, CR_ROLZ_M_LEN, len, 4u, rc_side, s_side);
            cr_side_encode(sh, CR_ROLZ_M_IDX, rank, 4u, rc_side, s_side);
            codes++;
            m.ctx = after;
        } else {
            const uint32_t c = cr_window_at(win, pos);
            cr_ppm_encode(m, rc_main, c, s_main, F, (m.ctx << 8) | c, 0u CR_PROF_PASS);
            if (c == esc) { cr_side_encode(sh, CR_ROLZ_M_LEN, 0u, 4u, rc_side, s_side); codes++; }
            cr_ppm_push(m, c);
        }
        pos += len;
        if (CR_ROLZ_HEADER + s_main.n >= n) { stored = true; break; }    /* cr-coder.c:233-235 */
    }
    cr_node_writeback(m);
    if (persist) { cr_ppm_suspend(m); cr_side_park(sh, arena + L.off_keep); }
    if (stored) {
        cr_wave_sync();
        cr_rolz_store_raw(src, n, dst);
        return CR_ROLZ_HEADER + n;
    }
    cr_rc_pin(rc_main); cr_rc_flush(rc_main, s_main);
    cr_rc_pin(rc_side); cr_rc_flush(rc_side, s_side);
    cr_wave_sync();
    const uint32_t o_side = CR_ROLZ_HEADER + s_main.n;
    for (uint32_t i = lane; i < s_side.n; i += CRGPU_WAVE) dst[o_side + i] = s_side.dst[i];
    if (lane < CR_ROLZ_HEADER) {                                         /* cr-coder.c:241-245 */
        const uint32_t fields[4] = {(uint32_t)src[0] | (1u << 8) | (esc << 16), n, codes, o_side};
        dst[lane] = (uint8_t)(fields[lane >> 2] >> (8u * (lane & 3u)));
    }
    return o_side + s_side.n;
}

/* ------------------------------------------------------------------ lzdecode, cr-coder.c:283-379 */

/* matcher_update for the positions q0 .. q0+np-1 at once (lane j holds the 8 bytes in front of q0+j, newest
 * in the top byte): links from the lanes of the batch or from the heads, heads raised to the batch's newest */
CR_DEV void cr_rolz_feed(const CrRolzTables& T, uint32_t* row_head, u64 x, uint32_t q0, uint32_t np, bool ctx4) {
    const uint32_t lane = cr_lane();
    const bool act = lane < np;
    const uint32_t p = q0 + lane;
    const uint32_t b1 = (uint32_t)(x >> 56), b2 = (uint32_t)(x >> 48) & 0xffu, b3 = (uint32_t)(x >> 40) & 0xffu, b4 = (uint32_t)(x >> 32) & 0xffu;
    const uint32_t ring = p <= CR_ROLZ_WARM ? 0u : cr_rolz_hash(b1, b2, b3, b4, ctx4);
    const uint32_t row = p <= CR_ROLZ_WARM ? 0u : b1;
    const int qr = cr_prev_same_bits<18>(ring, act), qw = cr_prev_same_bits<8>(row, act);
    if (act) {
        uint32_t c = CR_ROLZ_NONE;
        if (qr >= 0) c = q0 + (uint32_t)qr;
        else { const uint32_t v = cr_ld32(T.ring_head + ring); if (v) c = v - 1u; }
        T.ring_prev[p] = c;
        c = CR_ROLZ_NONE;
        if (qw >= 0) c = q0 + (uint32_t)qw;
        else { const uint32_t v = row_head[row]; if (v) c = v - 1u; }
        T.row_prev[p] = c;
    }
    cr_wave_sync();
    if (act) { atomicMax(T.ring_head + ring, p + 1u); atomicMax(row_head + row, p + 1u); }
    cr_wave_sync();
}

/* matcher_getpos, cr-matcher.c:86-91, for the write position `have` whose preceding 8 bytes are x8 */
/* The same with a history per position: hist[8p .. 8p+7] = the eight entries of p's ring in front of it, newest first
 * (hist[8p] is the link the plain version stores), so that a rank costs rank / 8 + 2 loads instead of rank + 1.
 * A position whose predecessor sits in the same batch takes that lane's registers (chains inside a batch are
 * resolved level by level), everybody else one 28-byte load from the predecessor's history. */
CR_DEV void cr_rolz_feed_hist(const CrRolzTables& T, uint32_t* hist, uint32_t* row_head, u64 x, uint32_t q0, uint32_t np, bool ctx4) {
    const uint32_t lane = cr_lane();
    const bool act = lane < np;
    const uint32_t p = q0 + lane;
    const uint32_t b1 = (uint32_t)(x >> 56), b2 = (uint32_t)(x >> 48) & 0xffu, b3 = (uint32_t)(x >> 40) & 0xffu, b4 = (uint32_t)(x >> 32) & 0xffu;
    const uint32_t ring = p <= CR_ROLZ_WARM ? 0u : cr_rolz_hash(b1, b2, b3, b4, ctx4);
    const uint32_t row = p <= CR_ROLZ_WARM ? 0u : b1;
    /* lanes of the batch in the same ring / row: the highest lower one is the predecessor, the highest one writes the head
     * (the wave owns these tables: plain stores, no atomics) */
    const u64 mr = cr_same_key_mask<18>(ring, act), mw = cr_same_key_mask<8>(row, act);
    const u64 below = (1ull << lane) - 1ull;
    const int qr = (mr & below) ? 63 - (int)__builtin_clzll(mr & below) : -1, qw = (mw & below) ? 63 - (int)__builtin_clzll(mw & below) : -1;
    const bool last_r = act && (mr >> lane) >> 1 == 0ull, last_w = act && (mw >> lane) >> 1 == 0ull;
    uint32_t h[8];
#pragma unroll
    for (int k = 0; k < 8; k++) h[k] = CR_ROLZ_NONE;
    bool done = true;
    if (act) {
        if (qr >= 0) { h[0] = q0 + (uint32_t)qr; done = false; }
        else {
            const uint32_t v = cr_ld32(T.ring_head + ring);
            if (v) {
                h[0] = v - 1u;
                const u64* hp = reinterpret_cast<const u64*>(hist + (u64)h[0] * 8u);      /* (agent-scope loads: written by earlier batches) */
                const u64 a = cr_ld64(hp), b = cr_ld64(hp + 1), c2 = cr_ld64(hp + 2), d2 = cr_ld64(hp + 3);
                h[1] = (uint32_t)a; h[2] = (uint32_t)(a >> 32); h[3] = (uint32_t)b; h[4] = (uint32_t)(b >> 32);
                h[5] = (uint32_t)c2; h[6] = (uint32_t)(c2 >> 32); h[7] = (uint32_t)d2;
            }
        }
        uint32_t c = CR_ROLZ_NONE;
        if (qw >= 0) c = q0 + (uint32_t)qw;
        else { const uint32_t v = row_head[row]; if (v) c = v - 1u; }
        T.row_prev[p] = c;
    }
    const int from = qr >= 0 ? qr : (int)lane;
    while (cr_ballot(!done)) {                                           /* in-batch chains: one level per round */
        const int pd = __shfl((int)done, from);
        uint32_t g[7];
#pragma unroll
        for (int k = 0; k < 7; k++) g[k] = (uint32_t)__shfl((int)h[k], from);
        if (!done && pd) {
#pragma unroll
            for (int k = 0; k < 7; k++) h[k + 1] = g[k];
            done = true;
        }
    }
    if (act) {
        *reinterpret_cast<uint4*>(hist + (u64)p * 8u) = make_uint4(h[0], h[1], h[2], h[3]);
        *reinterpret_cast<uint4*>(hist + (u64)p * 8u + 4u) = make_uint4(h[4], h[5], h[6], h[7]);
    }
    if (last_r) cr_st32(T.ring_head + ring, p + 1u);
    if (last_w) row_head[row] = p + 1u;
    cr_wave_sync();
}
CR_DEV uint32_t cr_rolz_getpos_hist(const CrRolzTables& T, const uint32_t* hist, const uint32_t* row_head, uint32_t rank, uint32_t have, u64 x8, bool ctx4) {
    if (rank < CR_ROLZ_RING) {
        const uint32_t ring = have <= CR_ROLZ_WARM ? 0u
            : cr_rolz_hash((uint32_t)(x8 >> 56), (uint32_t)(x8 >> 48) & 0xffu, (uint32_t)(x8 >> 40) & 0xffu, (uint32_t)(x8 >> 32) & 0xffu, ctx4);
        const uint32_t v = cr_uni(cr_ld32(T.ring_head + ring));
        uint32_t p = v ? v - 1u : CR_ROLZ_NONE, r = rank;
        while (r >= 8u && p != CR_ROLZ_NONE) { p = cr_uni(cr_ld32(hist + (u64)p * 8u + 7u)); r -= 8u; }
        if (r > 0u && p != CR_ROLZ_NONE) p = cr_uni(cr_ld32(hist + (u64)p * 8u + (r - 1u)));
        return p;
    }
    const uint32_t row = have <= CR_ROLZ_WARM ? 0u : (uint32_t)(x8 >> 56);
    const uint32_t v = cr_uni(row_head[row]);
    uint32_t p = v ? v - 1u : CR_ROLZ_NONE;
    for (uint32_t i = CR_ROLZ_RING; i < rank && p != CR_ROLZ_NONE; i++) p = cr_uni(cr_ld32(T.row_prev + p));
    return p == CR_ROLZ_NONE ? 0u : p;
}
CR_DEV uint32_t cr_rolz_getpos(const CrRolzTables& T, const uint32_t* row_head, uint32_t rank, uint32_t have, u64 x8, bool ctx4) {
    if (rank < CR_ROLZ_RING) {
        const uint32_t ring = have <= CR_ROLZ_WARM ? 0u
            : cr_rolz_hash((uint32_t)(x8 >> 56), (uint32_t)(x8 >> 48) & 0xffu, (uint32_t)(x8 >> 40) & 0xffu, (uint32_t)(x8 >> 32) & 0xffu, ctx4);
        const uint32_t v = cr_uni(cr_ld32(T.ring_head + ring));
        uint32_t p = v ? v - 1u : CR_ROLZ_NONE;
        for (uint32_t i = 0; i < rank && p != CR_ROLZ_NONE; i++) p = cr_uni(cr_ld32(T.ring_prev + p));
        return p;
    }
    const uint32_t row = have <= CR_ROLZ_WARM ? 0u : (uint32_t)(x8 >> 56);
    const uint32_t v = cr_uni(row_head[row]);
    uint32_t p = v ? v - 1u : CR_ROLZ_NONE;
    for (uint32_t i = CR_ROLZ_RING; i < rank && p != CR_ROLZ_NONE; i++) p = cr_uni(cr_ld32(T.row_prev + p));
    return p == CR_ROLZ_NONE ? 0u : p;
}

CR_DEV uint32_t cr_rolz_decode_block(const uint8_t* src, uint32_t n, uint8_t* dst, uint32_t cap, const CrRolzTables& T, uint32_t* row_head,
                                     uint8_t* arena, const CrArenaLayout& L, uint32_t fresh, uint32_t persist, CrRoxShared& sh) {
    const uint32_t lane = cr_lane();
    if (n < CR_ROLZ_HEADER) return 0xFFFFFFFFu;
    if (src[1] == 0) {                                                   /* cr-coder.c:303-308 */
        uint32_t raw = n - CR_ROLZ_HEADER;
        if (raw > cap) return 0xFFFFFFFFu;
        for (uint32_t i = lane; i < raw; i += CRGPU_WAVE) dst[i] = src[CR_ROLZ_HEADER + i];
        return raw;
    }
    const uint32_t esc = src[2];
    uint32_t hw[3];
    for (int k = 0; k < 3; k++) hw[k] = (uint32_t)src[4 + 4 * k] | ((uint32_t)src[5 + 4 * k] << 8) | ((uint32_t)src[6 + 4 * k] << 16) | ((uint32_t)src[7 + 4 * k] << 24);
    const uint32_t total = cr_uni(hw[0]), o_side = cr_uni(hw[2]);
    uint32_t codes = cr_uni(hw[1]);
    if (total == 0 || total > cap || total > L.max_block || o_side < CR_ROLZ_HEADER || o_side > n) return 0xFFFFFFFFu;
    const bool ctx4 = total >= 4194304u;                                 /* using_ctx4, cr-coder.c:314 */
    CrPpm m;
    cr_ppm_attach(m, arena, L, persist ? L.cap_o3 : cr_log2_ceil_pow2(2u * total, 1024u, L.cap_o3));
    if (fresh) { cr_rolz_side_reset(sh); cr_ppm_reset(m); }
    else { cr_side_unpark(sh, arena + L.off_keep); cr_ppm_resume(m); }
    cr_fill(reinterpret_cast<uint8_t*>(T.ring_head), (u64)CR_ROLZ_BUCKETS * 4u, 0u);     /* matcher_init */
    for (uint32_t k = lane; k < 256u; k += CRGPU_WAVE) row_head[k] = 0;
    if (lane == 0) dst[0] = src[0];
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
    cr_wave_sync();
    CrSource in_main, in_side;
    cr_source_init(in_main, src + CR_ROLZ_HEADER, n - CR_ROLZ_HEADER);
    cr_source_init(in_side, src + o_side, n - o_side);
    CrRc rc_main, rc_side;
    cr_rc_dec_init(rc_main, in_main); cr_rc_dec_init(rc_side, in_side);
    CrFetch F; F.valid = 0; F.ctx = 0; F.with_row = 1;
#ifdef CRGPU_PROF
    CrProf prof; prof.last = 0;
    for (int i = 0; i < 8; i++) prof.acc[i] = 0;
#endif
    uint32_t have = 1, fed = 1;                /* positions < fed are in the tables (or below 16: never fed) */
    u64 x8 = (u64)cr_uni((uint32_t)src[0]) << 56;                         /* the 8 bytes in front of the write position */
    u64 pend_x = 0;                            /* lane j: those 8 bytes for position fed + j */
#define CR_ROLZ_LITERAL(byte_) do { \
        if (lane == 0) dst[have] = (uint8_t)(byte_); \
        if (have >= CR_ROLZ_WARM) { if (lane == have - fed) pend_x = x8; } else fed = have + 1u; \
        x8 = (x8 >> 8) | ((u64)(byte_) << 56); \
        have++; \
        if (have - fed == CRGPU_WAVE) { cr_rolz_feed(T, row_head, pend_x, fed, CRGPU_WAVE, ctx4); fed = have; } \
    } while (0)
    while (have < total) {                                               /* cr-coder.c:334-375 */
        const uint32_t s = cr_ppm_decode(m, rc_main, in_main, F CR_PROF_PASS);
        if (s != esc) {
            if (have >= cap) return 0xFFFFFFFFu;
            CR_ROLZ_LITERAL(s);
            cr_ppm_push(m, s);
            continue;
        }
        uint32_t len = 0, rank = 0;
        if (codes > 0u) {                                                /* cr-coder.c:265-277 */
            codes--;
            len = cr_side_decode(sh, CR_ROLZ_M_LEN, 4u, rc_side, in_side);
            if (len > 0u) rank = cr_side_decode(sh, CR_ROLZ_M_IDX, 4u, rc_side, in_side);
        }
        if (len == 0u) {                                                 /* the escape byte itself */
            if (have >= cap) return 0xFFFFFFFFu;
            CR_ROLZ_LITERAL(esc);
            cr_ppm_push(m, esc);
            continue;
        }
        if (have + len > total || have + len > cap || have < CR_ROLZ_WARM) return 0xFFFFFFFFu;   /* corrupt stream */
        if (have > fed) cr_rolz_feed(T, row_head, pend_x, fed, have - fed, ctx4);
        fed = have;
        cr_wave_sync();                                                  /* the literals' stores and the links are readable */
        const uint32_t from = cr_rolz_getpos(T, row_head, rank, have, x8, ctx4);
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
                m.ctx = (cr_lane_get(mine, l3 - 3u) << 24) | (cr_lane_get(mine, l3 - 2u) << 16) | (cr_lane_get(mine, l3 - 1u) << 8) | cr_lane_get(mine, l3);
            } else {
                cr_wave_sync();
                for (uint32_t i = len - 4u; i < len; i++) cr_ppm_push(m, cr_uni(dst[have + i]));
            }
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
                const uint32_t q = q0 + lane, np = have + len - q0 < CRGPU_WAVE ? have + len - q0 : CRGPU_WAVE;
                u64 xq = 0;
                if (q < have + len) xq = *reinterpret_cast<const cr_u64u*>(dst + q - 8);
                cr_rolz_feed(T, row_head, xq, q0, np, ctx4);
            }
            have += len;
            fed = have;
            x8 = *reinterpret_cast<const cr_u64u*>(dst + have - 8);
            x8 = ((u64)cr_uni((uint32_t)(x8 >> 32)) << 32) | cr_uni((uint32_t)x8);
        }
    }
#undef CR_ROLZ_LITERAL
    cr_node_writeback(m);
    if (persist) { cr_ppm_suspend(m); cr_side_park(sh, arena + L.off_keep); }
    return have;
}

#endif

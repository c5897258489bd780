/*
 * comprox_amd/csrc/crgpu_lzp.h — LZP predictor tables of the comprop codec on gfx950.
 *
 * Reference: /root/reference/src/ropmain/cr-matcher.c:31-96. The reference keeps three dense
 * "last position with this hashed context" tables (2^24, 2^20 and 2^16 u32 entries, 68 MB,
 * re-initialised per block). A block of n bytes can only ever populate n entries of each, so
 * here lzp8 / lzp4 are exact-keyed open-addressing tables of >= 2n slots (the stored key is the
 * reference's full 24 / 20-bit hash value, so two contexts collide here exactly when they
 * collide there) and lzp2 is the reference's own dense 65536-entry table.
 *
 * "Last position" is a maximum (positions only grow), so concurrent lanes may learn positions in
 * any order: claim the slot with a CAS, then atomicMax the packed {key+1, position}.
 */
#ifndef CRGPU_LZP_H
#define CRGPU_LZP_H

#include "crgpu_wave.h"

#define CR_LZP_MIN   4u      /* cr-matcher.h:36 */
#define CR_LZP_MAX   255u    /* cr-matcher.h:37 */
#define CR_LZP_TAIL  1024u   /* ropmain/cr-coder.c:103 */
#define CR_LZP_SKIP  9u      /* ropmain/cr-coder.c:143-145 */

struct CrLzp {
    u64*      t8;
    u64*      t4;
    uint32_t* t2;
    uint32_t  mask;      /* capacity - 1 of t8 / t4 */
    uint32_t  shift;     /* 32 - log2(capacity) */
};

CR_DEV void cr_lzp_attach(CrLzp& z, uint8_t* arena, const CrArenaLayout& L, uint32_t cap) {
    z.t8 = reinterpret_cast<u64*>(arena + L.off_lz8);
    z.t4 = reinterpret_cast<u64*>(arena + L.off_lz4);
    z.t2 = reinterpret_cast<uint32_t*>(arena + L.off_lz2);
    z.mask = cap - 1u;
    z.shift = 32u - (uint32_t)__builtin_ctz(cap);
}

/* workgroup-wide fill (all threads of the block), bytes a multiple of 16 */
CR_DEV void cr_fill_wg(uint8_t* dst, u64 bytes, uint32_t pattern) {
    uint4 v = make_uint4(pattern, pattern, pattern, pattern);
    for (u64 i = (u64)threadIdx.x * 16u; i < bytes; i += 16u * blockDim.x)
        *reinterpret_cast<uint4*>(dst + i) = v;
}
CR_DEV void cr_lzp_reset_wg(CrLzp& z) {
    cr_fill_wg(reinterpret_cast<uint8_t*>(z.t8), (u64)(z.mask + 1u) * 8u, 0u);
    cr_fill_wg(reinterpret_cast<uint8_t*>(z.t4), (u64)(z.mask + 1u) * 8u, 0u);
    cr_fill_wg(reinterpret_cast<uint8_t*>(z.t2), 65536u * 4u, 2u);
}

/* matcher_init, cr-matcher.c:35-50: empty entries answer 8 / 4 / 2 */
CR_DEV void cr_lzp_reset(CrLzp& z) {
    cr_fill(reinterpret_cast<uint8_t*>(z.t8), (u64)(z.mask + 1u) * 8u, 0u);
    cr_fill(reinterpret_cast<uint8_t*>(z.t4), (u64)(z.mask + 1u) * 8u, 0u);
    cr_fill(reinterpret_cast<uint8_t*>(z.t2), 65536u * 4u, 2u);
}

/* cr-matcher.c:31-33 on the little-endian 8 bytes in front of a position */
CR_DEV uint32_t cr_key8(u64 x) { return (uint32_t)((x ^ (x >> 20) ^ (x >> 40)) & 0xffffffull); }
CR_DEV uint32_t cr_key4(u64 x) { uint32_t y = (uint32_t)(x >> 32); return (y ^ (y >> 6) ^ (y >> 12)) & 0xfffffu; }
CR_DEV uint32_t cr_key2(u64 x) { return (uint32_t)(x >> 48); }

CR_DEV uint32_t cr_hslot(const CrLzp& z, uint32_t key) { return (key * 2654435761u) >> z.shift; }

/* per-lane lookup; returns the stored position or `dflt` */
CR_DEV uint32_t cr_htab_get(const CrLzp& z, const u64* t, uint32_t key, uint32_t dflt) {
    uint32_t h = cr_hslot(z, key);
    for (;;) {
        u64 v = cr_ld64(t + h);
        if (v == 0ull) return dflt;
        if ((uint32_t)(v >> 32) == key + 1u) return (uint32_t)v;
        h = (h + 1u) & z.mask;
    }
}
/* per-lane insert-or-raise */
CR_DEV void cr_htab_learn(const CrLzp& z, u64* t, uint32_t key, uint32_t pos) {
    uint32_t h = cr_hslot(z, key);
    const u64 val = ((u64)(key + 1u) << 32) | pos;
    for (;;) {
        u64 v = atomicCAS(t + h, 0ull, val);            /* claim an empty slot ... */
        if (v == 0ull) return;
        if ((uint32_t)(v >> 32) == key + 1u) { atomicMax(t + h, val); return; }   /* ... or raise ours */
        h = (h + 1u) & z.mask;
    }
}

/* continue a lookup / insert from slot h (after the home slot turned out to hold another key) */
CR_DEV uint32_t cr_htab_get_from(const CrLzp& z, const u64* t, uint32_t key, uint32_t dflt, uint32_t h) {
    for (;;) {
        u64 v = cr_ld64(t + h);
        if (v == 0ull) return dflt;
        if ((uint32_t)(v >> 32) == key + 1u) return (uint32_t)v;
        h = (h + 1u) & z.mask;
    }
}
CR_DEV void cr_htab_learn_from(const CrLzp& z, u64* t, uint32_t key, uint32_t pos, uint32_t h) {
    const u64 val = ((u64)(key + 1u) << 32) | pos;
    for (;;) {
        u64 v = atomicCAS(t + h, 0ull, val);
        if (v == 0ull) return;
        if ((uint32_t)(v >> 32) == key + 1u) { atomicMax(t + h, val); return; }
        h = (h + 1u) & z.mask;
    }
}

/* matcher_update for one position per active lane, cr-matcher.c:91-96; x = 8 bytes before pos.
 * The three tables' first atomics go out together (one memory round trip); only a home slot
 * taken by another key falls into the probing loop. */
CR_DEV void cr_lzp_learn(const CrLzp& z, u64 x, uint32_t pos) {
    const uint32_t k8 = cr_key8(x), k4 = cr_key4(x), k2 = cr_key2(x);
    const uint32_t h8 = cr_hslot(z, k8), h4 = cr_hslot(z, k4);
    const u64 val8 = ((u64)(k8 + 1u) << 32) | pos, val4 = ((u64)(k4 + 1u) << 32) | pos;
    u64 v8 = atomicCAS(z.t8 + h8, 0ull, val8);
    u64 v4 = atomicCAS(z.t4 + h4, 0ull, val4);
    atomicMax(z.t2 + k2, pos);
    if (v8 != 0ull) {
        if ((uint32_t)(v8 >> 32) == k8 + 1u) atomicMax(z.t8 + h8, val8);
        else cr_htab_learn_from(z, z.t8, k8, pos, (h8 + 1u) & z.mask);
    }
    if (v4 != 0ull) {
        if ((uint32_t)(v4 >> 32) == k4 + 1u) atomicMax(z.t4 + h4, val4);
        else cr_htab_learn_from(z, z.t4, k4, pos, (h4 + 1u) & z.mask);
    }
}

/* matcher_getpos (cr-matcher.c:59-73) for one uniform position whose preceding 8 bytes are x:
 * three table reads in flight together, then both context checks together. */
CR_DEV uint32_t cr_lzp_predict(const CrLzp& z, const uint8_t* d, u64 x) {
    const uint32_t k8 = cr_key8(x), k4 = cr_key4(x), k2 = cr_key2(x);
    const uint32_t h8 = cr_hslot(z, k8), h4 = cr_hslot(z, k4);
    u64 e8 = cr_ld64(z.t8 + h8);
    u64 e4 = cr_ld64(z.t4 + h4);
    uint32_t c2 = cr_ld32(z.t2 + k2);
    uint32_t c8 = 8u, c4 = 4u;
    if (e8 != 0ull) c8 = ((uint32_t)(e8 >> 32) == k8 + 1u) ? (uint32_t)e8 : cr_htab_get_from(z, z.t8, k8, 8u, (h8 + 1u) & z.mask);
    if (e4 != 0ull) c4 = ((uint32_t)(e4 >> 32) == k4 + 1u) ? (uint32_t)e4 : cr_htab_get_from(z, z.t4, k4, 4u, (h4 + 1u) & z.mask);
    u64 v8 = *reinterpret_cast<const cr_u64u*>(d + c8 - 8);
    uint32_t v4 = *reinterpret_cast<const cr_u32u*>(d + c4 - 4);
    uint32_t from = c2;
    if (v8 == x) from = c8;
    else if (v4 == (uint32_t)(x >> 32)) from = c4;
    return from;
}

/* Decoder, at a match token: matcher_update for the positions q0 .. q0+np-1 (lane j holds the
 * 8 bytes in front of q0+j) and matcher_getpos for the position that follows them (xh = its 8
 * bytes), in ONE memory round trip: the inserts and the three lookups go out together. A lookup
 * can race with this batch's inserts, which is harmless: if a batch position has the looked-up key
 * it wins by construction (it is the latest), and inserts of other keys never hide an older entry
 * of ours (open addressing without deletion). Returns the three candidates, wave-uniform. */
CR_DEV void cr_lzp_learn_predict(const CrLzp& z, u64 x, uint32_t q0, uint32_t np, u64 xh,
                                 uint32_t& c8, uint32_t& c4, uint32_t& c2) {
    const uint32_t lane = cr_lane();
    const bool act = lane < np;
    const uint32_t q = q0 + lane;
    const uint32_t k8 = cr_key8(x), k4 = cr_key4(x), k2 = cr_key2(x);
    const uint32_t h8 = cr_hslot(z, k8), h4 = cr_hslot(z, k4);
    const uint32_t K8 = cr_key8(xh), K4 = cr_key4(xh), K2 = cr_key2(xh);
    const uint32_t H8 = cr_hslot(z, K8), H4 = cr_hslot(z, K4);
    const u64 val8 = ((u64)(k8 + 1u) << 32) | q, val4 = ((u64)(k4 + 1u) << 32) | q;
    u64 v8 = 0, v4 = 0;
    if (act) {
        v8 = atomicCAS(z.t8 + h8, 0ull, val8);
        v4 = atomicCAS(z.t4 + h4, 0ull, val4);
        atomicMax(z.t2 + k2, q);
    }
    const u64 e8 = cr_ld64(z.t8 + H8);
    const u64 e4 = cr_ld64(z.t4 + H4);
    const uint32_t t2v = cr_ld32(z.t2 + K2);
    if (act && v8 != 0ull) {
        if ((uint32_t)(v8 >> 32) == k8 + 1u) atomicMax(z.t8 + h8, val8);
        else cr_htab_learn_from(z, z.t8, k8, q, (h8 + 1u) & z.mask);
    }
    if (act && v4 != 0ull) {
        if ((uint32_t)(v4 >> 32) == k4 + 1u) atomicMax(z.t4 + h4, val4);
        else cr_htab_learn_from(z, z.t4, k4, q, (h4 + 1u) & z.mask);
    }
    const uint32_t e8k = cr_uni((uint32_t)(e8 >> 32)), e8p = cr_uni((uint32_t)e8);
    const uint32_t e4k = cr_uni((uint32_t)(e4 >> 32)), e4p = cr_uni((uint32_t)e4);
    c8 = 8u; c4 = 4u; c2 = cr_uni(t2v);
    const u64 m8 = cr_ballot(act && k8 == K8), m4 = cr_ballot(act && k4 == K4), m2 = cr_ballot(act && k2 == K2);
    if (m8) c8 = q0 + 63u - (uint32_t)__builtin_clzll(m8);
    else if (e8k != 0u) c8 = (e8k == K8 + 1u) ? e8p : cr_uni(cr_htab_get_from(z, z.t8, K8, 8u, (H8 + 1u) & z.mask));
    if (m4) c4 = q0 + 63u - (uint32_t)__builtin_clzll(m4);
    else if (e4k != 0u) c4 = (e4k == K4 + 1u) ? e4p : cr_uni(cr_htab_get_from(z, z.t4, K4, 4u, (H4 + 1u) & z.mask));
    if (m2) c2 = q0 + 63u - (uint32_t)__builtin_clzll(m2);
}

/* For every active lane: the highest lower active lane holding the same key, or -1. */
CR_DEV int cr_prev_same(uint32_t key, bool active) {
    int prev = -1;
    const uint32_t lane = cr_lane();
    u64 todo = cr_ballot(active);
    while (todo) {
        uint32_t leader = (uint32_t)__builtin_ctzll(todo);
        uint32_t k = cr_lane_get(key, leader);
        u64 same = cr_ballot(active && key == k);
        if (active && key == k) {
            u64 lower = same & ((1ull << lane) - 1ull);
            prev = lower ? 63 - (int)__builtin_clzll(lower) : -1;
        }
        todo &= ~same;
    }
    return prev;
}

/* number of equal leading bytes of d[a..] and d[b..], capped at CR_LZP_MAX (cr-matcher.c:80-84);
 * the caller guarantees 263 readable bytes behind both */
CR_DEV uint32_t cr_common_len(const uint8_t* d, uint32_t a, uint32_t b) {
    uint32_t len = 0;
    while (len < CR_LZP_MAX) {
        u64 x = *reinterpret_cast<const cr_u64u*>(d + a + len) ^ *reinterpret_cast<const cr_u64u*>(d + b + len);
        if (x) { len += (uint32_t)__builtin_ctzll(x) >> 3; break; }
        len += 8;
    }
    return len < CR_LZP_MAX ? len : CR_LZP_MAX;
}

/* For every active lane: the highest lower active lane holding the same key, or -1.
 * 63 DPP wave_shr:1 steps: after d steps lane l looks at lane l-d's key. */
CR_DEV int cr_prev_same_shift(uint32_t key, bool active) {
    const uint32_t lane = cr_lane();
    const uint32_t k = active ? key : (0x80000000u | lane);      /* inactive lanes match nobody */
    uint32_t t = k;
    int prev = -1;
#pragma unroll
    for (int d = 1; d < 64; d++) {
        t = cr_shift_up1(t, 0xFFFFFFFFu);
        prev = (prev < 0 && t == k) ? (int)lane - d : prev;
    }
    return prev;
}

/* Same result, bit-sliced: one ballot per key bit; a lane keeps the lanes that agree with it on
 * every bit. NBITS ballots + ~4 VALU each, all independent (no DPP dependency chain). */
/* mask of the active lanes holding the same key as this lane (this lane included when active) */
template <int NBITS>
CR_DEV u64 cr_same_key_mask(uint32_t key, bool active) {
    u64 m = cr_ballot(active);
#pragma unroll
    for (int b = 0; b < NBITS; b++) {
        const bool bit = ((key >> b) & 1u) != 0u;
        const u64 ball = cr_ballot(bit);
        m &= bit ? ball : ~ball;
    }
    return active ? m : 0ull;
}
template <int NBITS>
CR_DEV int cr_prev_same_bits(uint32_t key, bool active) {
    const u64 m = cr_same_key_mask<NBITS>(key, active) & ((1ull << cr_lane()) - 1ull);
    return m ? 63 - (int)__builtin_clzll(m) : -1;
}

/*
 * Encoder side: agreement length for EVERY position p in [9, n-1024) at once (kernel k_rop_lzp,
 * 4 waves per datablock).
 * matcher_lookup(p) only depends on which positions q < p have been learned, and by the time the
 * reference asks about p it has learned every q in [9, p) (ropmain/cr-coder.c:101-109: all bytes
 * of every earlier token are fed to matcher_update). So the answer is parse-independent:
 *   candidate_k(p) = max{ q in [9,p) : key_k(q) == key_k(p) }, else the table's default.
 * Phase A: waves 0,1,2 each own ONE table and sweep the block 64 positions per step (table state
 * covers earlier steps, cr_prev_same_bits the positions inside the step), writing candidate
 * arrays. Phase B: all waves verify contexts and measure agreement lengths, position-parallel.
 */
struct CrLzpScratch {
    uint32_t* c8;
    uint32_t* c4;
    uint32_t* c2;
};

/* Loads and stores only: the wave owns its table, so nothing needs to be atomic. A step's distinct keys are looked
 * up by their first lanes (probe rounds of plain loads); keys that are new claim the empty slot their probe ended in -
 * lanes of one round ending in the SAME empty slot are told apart in registers (lowest lane wins, the others walk on) -
 * and every first lane then stores the step's last position with its key. */
template <int NBITS>
CR_DEV void cr_lzp_sweep_table(const CrLzp& z, int which, const uint8_t* d, uint32_t limit, uint32_t* cand) {
    const uint32_t lane = cr_lane();
    u64* const t = which == 0 ? z.t8 : z.t4;
    u64 xn = 0;
    if (CR_LZP_SKIP + lane < limit) xn = *reinterpret_cast<const cr_u64u*>(d + CR_LZP_SKIP + lane - 8);
    for (uint32_t p0 = CR_LZP_SKIP; p0 < limit; p0 += CRGPU_WAVE) {
        const uint32_t p = p0 + lane;
        const bool act = p < limit;
        const u64 x = xn;
        if (p + CRGPU_WAVE < limit) xn = *reinterpret_cast<const cr_u64u*>(d + p + CRGPU_WAVE - 8);   /* next step's context */
        const uint32_t key = which == 0 ? cr_key8(x) : which == 1 ? cr_key4(x) : cr_key2(x);
        const u64 same = cr_same_key_mask<NBITS>(key, act);          /* the step's lanes with this lane's key */
        const u64 lower = same & ((1ull << lane) - 1ull);
        const bool first = act && lower == 0ull;
        const uint32_t last = p0 + 63u - (uint32_t)__builtin_clzll(same | 1ull);
        uint32_t c = which == 0 ? 8u : which == 1 ? 4u : 2u;
        if (which == 2) {
            if (first) { c = cr_ld32(z.t2 + key); cr_st32(z.t2 + key, last); }
        } else {
            const u64 val = ((u64)(key + 1u) << 32) | last;
            uint32_t h = cr_hslot(z, key);
            bool todo = first;
            while (cr_ballot(todo)) {
                u64 v = 0;
                if (todo) v = cr_ld64(t + h);
                const bool hit = todo && (uint32_t)(v >> 32) == key + 1u;
                const bool empty = todo && v == 0ull;
                /* lanes that ended in the same empty slot: the lowest takes it */
                u64 rivals = 0ull;
                if (__builtin_popcountll(cr_ballot(empty)) > 1) rivals = cr_same_key_mask<26>(h, empty) & ((1ull << lane) - 1ull);
                const bool take = hit || (empty && rivals == 0ull);
                if (hit) c = (uint32_t)v;
                if (take) cr_st64(t + h, val);
                if (todo && !take) h = (h + 1u) & z.mask;
                todo = todo && !take;
            }
        }
        if (act) {
            if (lower) c = p0 + 63u - (uint32_t)__builtin_clzll(lower);   /* an earlier position of this very step */
            cand[p] = c;
        }
    }
}

/* blockDim.x == 256; every thread of the workgroup calls this with the same arguments */
CR_DEV void cr_lzp_block_parallel(const CrLzp& z, const CrLzpScratch& sc, const uint8_t* d, uint32_t n, uint8_t* lens) {
    if (n <= CR_LZP_TAIL + CR_LZP_SKIP) return;
    const uint32_t limit = n - CR_LZP_TAIL;           /* positions with p + 1024 < n */
    const uint32_t w = cr_wave_id();
    if (w == 0) cr_lzp_sweep_table<24>(z, 0, d, limit, sc.c8);
    else if (w == 1) cr_lzp_sweep_table<20>(z, 1, d, limit, sc.c4);
    else if (w == 2) cr_lzp_sweep_table<16>(z, 2, d, limit, sc.c2);
    __syncthreads();
    for (uint32_t p = CR_LZP_SKIP + threadIdx.x; p < limit; p += blockDim.x) {
        u64 x = *reinterpret_cast<const cr_u64u*>(d + p - 8);
        uint32_t c8 = sc.c8[p], c4 = sc.c4[p], c2 = sc.c2[p];
        u64 v8 = *reinterpret_cast<const cr_u64u*>(d + c8 - 8);
        uint32_t v4 = *reinterpret_cast<const cr_u32u*>(d + c4 - 4);
        /* matcher_getpos, cr-matcher.c:59-73 */
        uint32_t from = c2;
        if (v8 == x) from = c8;
        else if (v4 == (uint32_t)(x >> 32)) from = c4;
        /* matcher_lookup, cr-matcher.c:75-89 */
        uint32_t len = from ? cr_common_len(d, from, p) : 0u;
        lens[p] = (uint8_t)(len < CR_LZP_MIN ? 1u : len);
    }
}

#endif

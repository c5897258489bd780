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
        u64 v = cr_ld64(t + h);
        if (v == 0ull) {
            v = atomicCAS(t + h, 0ull, val);
            if (v == 0ull) return;
        }
        if ((uint32_t)(v >> 32) == key + 1u) { atomicMax(t + h, val); return; }
        h = (h + 1u) & z.mask;
    }
}

/* matcher_update for one position per active lane, cr-matcher.c:91-96; x = 8 bytes before pos */
CR_DEV void cr_lzp_learn(const CrLzp& z, u64 x, uint32_t pos) {
    cr_htab_learn(z, z.t8, cr_key8(x), pos);
    cr_htab_learn(z, z.t4, cr_key4(x), pos);
    atomicMax(z.t2 + cr_key2(x), pos);
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

/*
 * Encoder side: agreement length for EVERY position p in [9, n-1024) at once.
 * matcher_lookup(p) only depends on which positions q < p have been learned, and by the time the
 * reference asks about p it has learned every q in [9, p) (ropmain/cr-coder.c:101-109: all bytes
 * of every earlier token are fed to matcher_update). So the answer is parse-independent:
 *   candidate_k(p) = max{ q in [9,p) : key_k(q) == key_k(p) }, else the table's default,
 * evaluated 64 positions per step: table state covers earlier steps, cr_prev_same covers the
 * positions inside the step.
 */
CR_DEV void cr_lzp_scan_block(const CrLzp& z, const uint8_t* d, uint32_t n, uint8_t* lens) {
    if (n <= CR_LZP_TAIL + CR_LZP_SKIP) return;
    const uint32_t limit = n - CR_LZP_TAIL;           /* positions with p + 1024 < n */
    const uint32_t lane = cr_lane();
    for (uint32_t p0 = CR_LZP_SKIP; p0 < limit; p0 += CRGPU_WAVE) {
        const uint32_t p = p0 + lane;
        const bool act = p < limit;
        u64 x = 0;
        uint32_t k8 = 0, k4 = 0, k2 = 0, c8 = 8, c4 = 4, c2 = 2;
        if (act) {
            x = *reinterpret_cast<const cr_u64u*>(d + p - 8);
            k8 = cr_key8(x); k4 = cr_key4(x); k2 = cr_key2(x);
            c8 = cr_htab_get(z, z.t8, k8, 8u);
            c4 = cr_htab_get(z, z.t4, k4, 4u);
            c2 = cr_ld32(z.t2 + k2);
        }
        int q8 = cr_prev_same(k8, act), q4 = cr_prev_same(k4, act), q2 = cr_prev_same(k2, act);
        if (act) {
            if (q8 >= 0) c8 = p0 + (uint32_t)q8;
            if (q4 >= 0) c4 = p0 + (uint32_t)q4;
            if (q2 >= 0) c2 = p0 + (uint32_t)q2;
            /* matcher_getpos, cr-matcher.c:59-73 */
            uint32_t from = c2;
            if (*reinterpret_cast<const cr_u64u*>(d + c8 - 8) == x) from = c8;
            else if (*reinterpret_cast<const cr_u32u*>(d + c4 - 4) == (uint32_t)(x >> 32)) from = c4;
            /* matcher_lookup, cr-matcher.c:75-89 */
            uint32_t len = from ? cr_common_len(d, from, p) : 0u;
            lens[p] = (uint8_t)(len < CR_LZP_MIN ? 1u : len);
            cr_lzp_learn(z, x, p);
        }
        cr_wave_sync();
    }
}

#endif

/*
 * comprox_amd/csrc/crgpu_rox3.h — comprox's hash-chain and short-cache links without tables (kernel k_rox_links_lds,
 * 8 waves per datablock, blocks of up to 28 672 bytes).
 *
 * Reference: /root/reference/src/roxmain/cr-matcher.c:89-148 (matcher_init: m_next[p] = the previous position of the
 * same hash class) and :213-216,319-331 (m_short_cache: the last position with the same 6-byte hash). k_rox_match
 * (crgpu_rox.h) gets both by sweeping head tables in position order through HBM — 3.4 of its 4.6 ms on the bench shard;
 * both are "the previous position with the same key", i.e. cr_lz2_prev_same (crgpu_lzp2.h): the positions sorted by key
 * in LDS, the left neighbour of equal key is the link. The links go to the global arrays k_rox_match's search phase
 * reads; k_rox_match skips its sweeps for the blocks done here.
 */
#ifndef CRGPU_ROX3_H
#define CRGPU_ROX3_H

#include "crgpu_rox.h"
#include "crgpu_lzp2.h"

struct CrRoxClassKey {                  /* cr-matcher.c:100-140: first bucket by s[0] + s[1] mod 20, second by the hash of match_min bytes */
    const uint8_t* d;                   /* the block in LDS */
    uint32_t classes, long_min;
    CR_DEV uint32_t operator()(uint32_t p) const { return (((uint32_t)d[p] + d[p + 1u]) % 20u) * classes + cr_rox_mix(d + p, long_min) % classes; }
};
struct CrRoxNearKey {                   /* cr-matcher.c:213-216 */
    const uint8_t* d;
    CR_DEV uint32_t operator()(uint32_t p) const { return cr_rox_mix(d + p, CR_ROX_NEAR_MIN) & 0xffffu; }
};

/* CR_ROX_TAIL < n <= CR_LZ2_MAXN; every thread of the workgroup calls this with the same arguments */
CR_DEV void cr_rox_links_block_lds(const CrLz2Shared& S, const uint8_t* g, uint32_t n, uint32_t long_min, const CrRoxTables& T) {
    cr_lz2_stage_block(S, g, n);
    const uint32_t lim_c = n > CR_ROX_MAX ? n - CR_ROX_MAX : 0u;          /* positions closer to the end are never linked (:118) */
    const uint32_t lim_n = n > CR_ROX_TAIL ? n - CR_ROX_TAIL : 0u;
    uint32_t* const prev = T.prev;
    uint32_t* const nprev = T.nprev;
    for (uint32_t p = lim_c + threadIdx.x; p < n; p += blockDim.x) prev[p] = CR_ROX_NONE;
    CrRoxClassKey ck; ck.d = S.src; ck.classes = 20u + n / 25u; ck.long_min = long_min;
    /* (the links are gathered in LDS and leave as coalesced u32 stores: crgpu_lzp2.h, cr_lz2_prev_same_lds) */
    if (lim_c) {
        const uint16_t* const lk = cr_lz2_prev_same_lds(S, ck, 0u, lim_c, 16u, 0xffffu);
        for (uint32_t p = threadIdx.x; p < lim_c; p += blockDim.x) { const uint32_t q = lk[p]; prev[p] = q == 0xffffu ? CR_ROX_NONE : q; }
    }
    CrRoxNearKey nk; nk.d = S.src;
    if (lim_n) {
        const uint16_t* const lk = cr_lz2_prev_same_lds(S, nk, 0u, lim_n, 16u, 0u);          /* an untouched slot reads 0 */
        for (uint32_t p = threadIdx.x; p < lim_n; p += blockDim.x) nprev[p] = lk[p];
    }
}


/* the same for CR_LZ2_MAXN < n <= CR_LZ3_MAXN: the positions sorted in groups by key beside the staged block (crgpu_lzp2.h, round
 * 4); the links go straight to the global arrays (scattered stores: 65 537 answers do not fit beside the block). Returns false
 * when the keys do not split into groups (k_rox_match then sweeps its tables for the block). */
CR_DEV bool cr_rox_links_block_lds64(const CrLz2Shared& S, CrLz3Groups& G, const uint8_t* g, uint32_t n, uint32_t long_min, const CrRoxTables& T) {
    cr_lz2_stage_block(S, g, n);
    const uint32_t lim_c = n > CR_ROX_MAX ? n - CR_ROX_MAX : 0u;          /* positions closer to the end are never linked (:118) */
    const uint32_t lim_n = n > CR_ROX_TAIL ? n - CR_ROX_TAIL : 0u;
    uint32_t* const prev = T.prev;
    uint32_t* const nprev = T.nprev;
    for (uint32_t p = lim_c + threadIdx.x; p < n; p += blockDim.x) prev[p] = CR_ROX_NONE;
    CrRoxClassKey ck; ck.d = S.src; ck.classes = 20u + n / 25u; ck.long_min = long_min;
    /* (20 x (20 + n / 25) classes: 52 820 at 65 537 bytes, 16 bits) */
    if (lim_c && !cr_lz3_prev_same(S, G, ck, 0u, lim_c, 16u, [prev](uint32_t p, uint32_t q) { prev[p] = q == CR_LZ2_NONE ? CR_ROX_NONE : q; })) return false;
    CrRoxNearKey nk; nk.d = S.src;
    if (lim_n && !cr_lz3_prev_same(S, G, nk, 0u, lim_n, 16u, [nprev](uint32_t p, uint32_t q) { nprev[p] = q == CR_LZ2_NONE ? 0u : q; })) return false;   /* an untouched slot reads 0 */
    return true;
}

#endif

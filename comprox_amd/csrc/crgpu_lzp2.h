/*
 * comprox_amd/csrc/crgpu_lzp2.h — the LZP pre-pass without tables: "previous position with the same key" by a stable
 * radix sort of the block's positions in LDS (kernel k_rop_lzp_lds, 8 waves per datablock, blocks of up to 28 672 bytes).
 *
 * Reference: /root/reference/src/ropmain/cr-matcher.c:31-96. What matcher_lookup(p) gets out of the three "last
 * position with this hashed context" tables is, for each of them,
 *     candidate_k(p) = max{ q in [9, p) : key_k(q) == key_k(p) },  else the table's default (8 / 4 / 2)
 * (crgpu_lzp.h explains why this is parse-independent). k_rop_lzp finds it by sweeping a hash table in HBM in position
 * order: 64 positions per step, two or three dependent probe rounds per step, ~12 us per step with 1 526 blocks
 * resident — 11.8 GB of table traffic and 5 ms on the bench shard although the tables hold 24 KB of positions per block.
 * Here the positions are SORTED by key instead, stably (LSD radix sort, 8-bit digits, u16 position records ping-ponging
 * between two LDS buffers), which puts the positions of a key next to each other in ascending order: the candidate of a
 * position is its left neighbour if that one has the same key. Exact by construction — the keys are the reference's
 * own hash values, compared in full; nothing is probabilistic. No table, no HBM traffic but the block itself (read
 * once, coalesced), the candidate arrays (as before) and the lengths.
 *
 * The block itself is staged in LDS too: every pass needs the key of every record, i.e. a gather of the 8 bytes in
 * front of a position, and 64-lane gathers through the CU's one L1 / address path (~128 clocks per wave-instruction,
 * 16 waves queueing) were 2/3 of the kernel's time when the keys came from global memory (3.2 ms; timing experiments
 * with -DCR_LZ2_EXP=2: every candidate the table's default, i.e. no sort). From LDS a gather is three aligned dword reads and two v_alignbit.
 *
 * LDS: two u16[28 672] record buffers + the block (28 KB) + u32[waves][256] digit counts = 149 KB with 8 waves (157 KB with
 * the 16 of k_rolz_match_lds) -> one block per CU at a time; the 8-wave kernels leave 11 KB of the CU's 160 to others. The dictionary stage's blocks (23.8 KB on the bench corpus) take this path; larger blocks keep k_rop_lzp.
 */
#ifndef CRGPU_LZP2_H
#define CRGPU_LZP2_H

#include "crgpu_lzp.h"
#include "crgpu_rop2.h"       /* cr_lds_order, cr_wg_sync_global */

#define CR_LZ2_MAXN    28672u
#ifndef CR_LZ2_THREADS
#define CR_LZ2_THREADS 512u      /* 8 waves: 2.46 ms on the bench shard against 2.64 with 16 and 3.22 with 4 */
#endif
#define CR_LZ2_MAX_WAVES 16u                 /* the digit counts are laid out for up to 1 024 threads; a kernel may launch fewer */
#define CR_LZ2_SRC_BYTES (CR_LZ2_MAXN + 32u)
/* dynamic LDS of a kernel that runs `waves_` waves on this layout. The 8-wave kernels ask for 152 608 of the CU's 163 840
 * bytes, which leaves room for the one-wave decoder workgroups (256 bytes each) of another stream on the same CU. */
#define CR_LZ2_LDS_BYTES_FOR(waves_) (2u * CR_LZ2_MAXN * 2u + CR_LZ2_SRC_BYTES + (waves_) * 256u * 4u + 256u * 4u)
#define CR_LZ2_LDS_BYTES CR_LZ2_LDS_BYTES_FOR(CR_LZ2_THREADS / 64u)

struct CrLz2Shared {
    uint16_t* a;          /* u16[CR_LZ2_MAXN] */
    uint16_t* b;          /* u16[CR_LZ2_MAXN] */
    uint32_t* hist;       /* u32[waves][256]: per wave and digit, first a count, then the next free slot */
    uint32_t* base;       /* u32[256]: where a digit's run starts */
    uint8_t*  src;        /* u8[CR_LZ2_SRC_BYTES]: the block (16-byte aligned; 12 bytes behind any position are readable) */
};

/* the views of a kernel's dynamic LDS (`waves` = the waves it launches with, at most CR_LZ2_MAX_WAVES) */
CR_DEV CrLz2Shared cr_lz2_carve(uint8_t* lds, uint32_t waves) {
    CrLz2Shared S;
    S.a = reinterpret_cast<uint16_t*>(lds);
    S.b = S.a + CR_LZ2_MAXN;
    S.hist = reinterpret_cast<uint32_t*>(S.b + CR_LZ2_MAXN);
    S.base = S.hist + waves * 256u;
    S.src = reinterpret_cast<uint8_t*>(S.base + 256u);
    return S;
}

/* the 8 bytes s[a .. a + 8) of the block in LDS, any alignment: three aligned dwords, two funnel shifts */
CR_DEV u64 cr_lz2_read8(const uint8_t* s, uint32_t a) {
    const uint32_t* w = reinterpret_cast<const uint32_t*>(s) + (a >> 2);
    const uint32_t w0 = w[0], w1 = w[1], w2 = w[2];
    const uint32_t sh = (a & 3u) * 8u;
    const uint32_t lo = __builtin_amdgcn_alignbit(w1, w0, sh), hi = __builtin_amdgcn_alignbit(w2, w1, sh);
    return ((u64)hi << 32) | lo;
}
CR_DEV uint32_t cr_lz2_key(int which, const uint8_t* d, uint32_t p) {          /* d = the block in LDS */
    const u64 x = cr_lz2_read8(d, p - 8u);
    return which == 0 ? cr_key8(x) : which == 1 ? cr_key4(x) : cr_key2(x);
}
struct CrLzpKey {                                          /* key of a position for one of the three LZP tables */
    int which;
    const uint8_t* d;
    CR_DEV uint32_t operator()(uint32_t p) const { return cr_lz2_key(which, d, p); }
};
/* cr_common_len (crgpu_lzp.h) on the LDS copy */
CR_DEV uint32_t cr_lz2_common_len(const uint8_t* d, uint32_t a, uint32_t b) {
    uint32_t len = 0;
    while (len < CR_LZP_MAX) {
        const u64 x = cr_lz2_read8(d, a + len) ^ cr_lz2_read8(d, b + len);
        if (x) { len += (uint32_t)__builtin_ctzll(x) >> 3; break; }
        len += 8;
    }
    return len < CR_LZP_MAX ? len : CR_LZP_MAX;
}

/* One stable counting pass over `count` records on the digit (key >> shift) & 255. src == nullptr: the records are the
 * positions first, first + 1, ... in order (the first pass). Every thread of the workgroup calls this. */
template <class KeyFn>
CR_DEV void cr_lz2_pass(const CrLz2Shared& S, const KeyFn& key, uint32_t first, uint32_t count, uint32_t shift,
                        const uint16_t* src, uint16_t* dst) {
    const uint32_t lane = cr_lane(), w = cr_wave_id(), nw = blockDim.x >> 6;
    const uint32_t per = ((count + nw - 1u) / nw + 63u) & ~63u;      /* records per wave, whole chunks */
    const uint32_t lo = w * per < count ? w * per : count;
    const uint32_t hi = lo + per < count ? lo + per : count;
    uint32_t* myhist = S.hist + w * 256u;
    for (uint32_t k = lane; k < 256u; k += CRGPU_WAVE) myhist[k] = 0;
    cr_lds_order();
    /* 1: how many records of every digit this wave holds (the order inside a digit does not matter yet: LDS atomics).
     * The record and its key are fetched one chunk ahead: LDS read -> gather from the block is the long dependency. */
    {
        uint32_t dg_n = 0;
        if (lo + lane < hi) dg_n = (key(src ? (uint32_t)src[lo + lane] : first + lo + lane) >> shift) & 255u;
        for (uint32_t i0 = lo; i0 < hi; i0 += CRGPU_WAVE) {
            const uint32_t i = i0 + lane;
            const uint32_t dg = dg_n;
            if (i + CRGPU_WAVE < hi) dg_n = (key(src ? (uint32_t)src[i + CRGPU_WAVE] : first + i + CRGPU_WAVE) >> shift) & 255u;
            if (i < hi) atomicAdd(myhist + dg, 1u);
        }
    }
    __syncthreads();
    /* 2: digit-major exclusive sums: a digit's run holds wave 0's records first, then wave 1's, ... (stable) */
    if (threadIdx.x < 256u) {
        uint32_t run = 0;
        for (uint32_t v = 0; v < nw; v++) {
            const uint32_t c = S.hist[v * 256u + threadIdx.x];
            S.hist[v * 256u + threadIdx.x] = run;
            run += c;
        }
        S.base[threadIdx.x] = run;
    }
    __syncthreads();
    if (w == 0) {                                            /* exclusive scan of the 256 totals by one wave */
        uint32_t carry = 0;
        for (uint32_t k0 = 0; k0 < 256u; k0 += CRGPU_WAVE) {
            const uint32_t v = S.base[k0 + lane];
            const uint32_t incl = cr_scan_incl(v);
            S.base[k0 + lane] = carry + incl - v;
            carry += cr_lane_get(incl, 63);
        }
    }
    __syncthreads();
    if (threadIdx.x < 256u) {
        const uint32_t bs = S.base[threadIdx.x];
        for (uint32_t v = 0; v < nw; v++) S.hist[v * 256u + threadIdx.x] += bs;
    }
    __syncthreads();
    /* 3: place the records, each wave its own in order */
    uint32_t p_n = 0, dg_n = 0;
    if (lo + lane < hi) { p_n = src ? (uint32_t)src[lo + lane] : first + lo + lane; dg_n = (key(p_n) >> shift) & 255u; }
    for (uint32_t i0 = lo; i0 < hi; i0 += CRGPU_WAVE) {
        const uint32_t i = i0 + lane;
        const bool act = i < hi;
        const uint32_t p = p_n, dg = dg_n;
        if (i + CRGPU_WAVE < hi) { p_n = src ? (uint32_t)src[i + CRGPU_WAVE] : first + i + CRGPU_WAVE; dg_n = (key(p_n) >> shift) & 255u; }
        const u64 same = cr_same_key_mask<8>(dg, act);
        const u64 lower = same & ((1ull << lane) - 1ull);
        if (act) {
            const uint32_t at = myhist[dg];
            dst[at + (uint32_t)__builtin_popcountll(lower)] = (uint16_t)p;
        }
        cr_lds_order();
        if (act && (same >> lane) >> 1 == 0ull) myhist[dg] += (uint32_t)__builtin_popcountll(same);   /* the group's last lane */
        cr_lds_order();
    }
    __syncthreads();
}

/* "The previous position with the same key" for the positions first .. first + count - 1: sorts them by key (`bits`
 * key bits, stable) and calls out(p, q) for every position p with q = the largest earlier position of equal key, or
 * out(p, CR_LZ2_NONE). Returns the buffer that holds the sorted positions (the other one is free by then). */
#define CR_LZ2_NONE 0xFFFFFFFFu
template <class KeyFn, class OutFn>
CR_DEV const uint16_t* cr_lz2_prev_same(const CrLz2Shared& S, const KeyFn& key, uint32_t first, uint32_t count, uint32_t bits,
                                        uint16_t* buf0, uint16_t* buf1, const OutFn& out) {
    const uint16_t* cur = nullptr;
    uint16_t* nxt = buf0;
    for (uint32_t shift = 0; shift < bits; shift += 8u) {
        cr_lz2_pass(S, key, first, count, shift, cur, nxt);
        cur = nxt;
        nxt = cur == buf0 ? buf1 : buf0;
    }
    /* equal keys lie next to each other, positions ascending: the left neighbour is the previous position of the key */
    for (uint32_t i = threadIdx.x; i < count; i += blockDim.x) {
        const uint32_t p = cur[i];
        uint32_t q = CR_LZ2_NONE;
        if (i > 0u) {
            const uint32_t l = cur[i - 1u];
            if (key(l) == key(p)) q = l;
        }
        out(p, q);
    }
    __syncthreads();
    return cur;
}

/* cr_lz2_prev_same with its answers scattered into the record buffer the last pass left free: u16[count] indexed by
 * position - first, `none` where there is no earlier position of the key. In sorted order the answers come position by
 * position at random — as global stores that is 64 different lines per instruction; from LDS they leave coalesced.
 * The array is valid until the next sort writes the buffers. */
template <class KeyFn>
CR_DEV const uint16_t* cr_lz2_prev_same_lds(const CrLz2Shared& S, const KeyFn& key, uint32_t first, uint32_t count, uint32_t bits, uint32_t none) {
    uint16_t* const fr = (((bits + 7u) / 8u) & 1u) ? S.b : S.a;      /* passes write a, b, a, ...: the last one's other buffer */
    cr_lz2_prev_same(S, key, first, count, bits, S.a, S.b,
                     [fr, first, none](uint32_t p, uint32_t q) { fr[p - first] = (uint16_t)(q == CR_LZ2_NONE ? none : q); });
    return fr;
}

/* Candidates of one table for the positions 9 .. limit - 1, as u16[limit - 9] indexed by position - 9. The sorted order hands
 * them out position by position at random, so they are scattered into the record buffer the last pass left free (LDS), and
 * leave for `cand16` (global, optional) as one coalesced copy — as global stores they were 64 different lines per instruction,
 * 1 100 such instructions and 290 KB of partial-line writes per block. Returns the LDS array (valid until the next sort). */
CR_DEV const uint16_t* cr_lz2_table(const CrLz2Shared& S, int which, const uint8_t* d, uint32_t limit, uint16_t* cand16) {
    const uint32_t dflt = which == 0 ? 8u : which == 1 ? 4u : 2u;
    const uint32_t bits = which == 0 ? 24u : which == 1 ? 20u : 16u;
    const uint32_t count = limit - CR_LZP_SKIP;
    CrLzpKey key; key.which = which; key.d = d;
#if defined(CR_LZ2_EXP) && CR_LZ2_EXP == 2              /* timing experiment: no sort at all */
    uint16_t* const fr = S.a;
    for (uint32_t i = threadIdx.x; i < count; i += blockDim.x) fr[i] = (uint16_t)dflt;
    __syncthreads();
#else
    const uint16_t* const fr = cr_lz2_prev_same_lds(S, key, CR_LZP_SKIP, count, bits, dflt);
#endif
    if (cand16) {                                           /* 16 bytes per thread and round; both arrays are 16-byte aligned */
        for (uint32_t i = threadIdx.x * 8u; i < count; i += blockDim.x * 8u)
            *reinterpret_cast<uint4*>(cand16 + i) = *reinterpret_cast<const uint4*>(fr + i);
    }
    return fr;
}

/* the block into LDS, zero-padded by 16 bytes: 16 bytes per thread and round (the source may sit at any alignment) */
CR_DEV void cr_lz2_stage_block(const CrLz2Shared& S, const uint8_t* g, uint32_t n) {
    for (uint32_t i = threadIdx.x * 16u; i < n + 16u; i += blockDim.x * 16u) {
        uint4 v = make_uint4(0u, 0u, 0u, 0u);
        if (i + 16u <= n) __builtin_memcpy(&v, g + i, 16);
        else { uint8_t t[16] = {0}; for (uint32_t k = 0; k < 16u; k++) if (i + k < n) t[k] = g[i + k]; __builtin_memcpy(&v, t, 16); }
        *reinterpret_cast<uint4*>(S.src + i) = v;
    }
    __syncthreads();
}

/* blockDim.x == CR_LZ2_THREADS; n <= CR_LZ2_MAXN; every thread calls this with the same arguments */
CR_DEV void cr_lzp_block_lds(const CrLz2Shared& S, const CrLzpScratch& sc, const uint8_t* g, uint32_t n, uint8_t* lens) {
    if (n <= CR_LZP_TAIL + CR_LZP_SKIP) return;
    const uint32_t limit = n - CR_LZP_TAIL;           /* positions with p + 1024 < n */
    cr_lz2_stage_block(S, g, n);
    const uint8_t* d = S.src;
    /* the candidates of lzp8 and lzp4 wait in global memory (u16, indexed by position - 9) while the next table is sorted, those
     * of lzp2 stay in LDS */
    uint16_t* const g8 = reinterpret_cast<uint16_t*>((reinterpret_cast<uintptr_t>(sc.c8) + 15u) & ~(uintptr_t)15u);   /* (u32[max_block] each) */
    uint16_t* const g4 = reinterpret_cast<uint16_t*>((reinterpret_cast<uintptr_t>(sc.c4) + 15u) & ~(uintptr_t)15u);
    cr_lz2_table(S, 0, d, limit, g8);
    cr_lz2_table(S, 1, d, limit, g4);
    const uint16_t* const l2 = cr_lz2_table(S, 2, d, limit, nullptr);
    cr_wg_sync_global();
    for (uint32_t p = CR_LZP_SKIP + threadIdx.x; p < limit; p += blockDim.x) {
        const u64 x = cr_lz2_read8(d, p - 8u);
        const uint32_t c8 = g8[p - CR_LZP_SKIP], c4 = g4[p - CR_LZP_SKIP], c2 = l2[p - CR_LZP_SKIP];
        const u64 v8 = cr_lz2_read8(d, c8 - 8u);
        const uint32_t v4 = (uint32_t)(cr_lz2_read8(d, c4 - 4u));
        /* matcher_getpos, cr-matcher.c:59-73 */
        uint32_t from = c2;
        if (v8 == x) from = c8;
        else if (v4 == (uint32_t)(x >> 32)) from = c4;
        /* matcher_lookup, cr-matcher.c:75-89 */
        const uint32_t len = from ? cr_lz2_common_len(d, from, p) : 0u;
        lens[p] = (uint8_t)(len < CR_LZP_MIN ? 1u : len);
    }
}

#endif

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
#ifdef CR_LZ2_WAIT_ORDER                /* timing experiment: a counter wait between a chunk's LDS writes and the next chunk's reads */
#define CR_LZ2_ORDER() cr_lds_order()
#else
#define CR_LZ2_ORDER() cr_lds_order_sw()
#endif
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
template <int W> struct CrLzpKeyW {                        /* the same with the table fixed at compile time */
    const uint8_t* d;
    CR_DEV uint32_t operator()(uint32_t p) const {
        const u64 x = cr_lz2_read8(d, p - 8u);
        return W == 0 ? cr_key8(x) : W == 1 ? cr_key4(x) : cr_key2(x);
    }
};
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
        CR_LZ2_ORDER();
        if (act && (same >> lane) >> 1 == 0ull) myhist[dg] += (uint32_t)__builtin_popcountll(same);   /* the group's last lane */
        CR_LZ2_ORDER();
    }
    __syncthreads();
}

/* ---- round 4: passes that count the NEXT digit while they place (crgpu_lzp2.h, "fused" passes) -----------------------
 * cr_lz2_pass computes every record's key twice per pass (once to count its digit, once to place it). Here a pass only
 * places: the per-wave digit counts it needs were added up by the pass before it, which knows each record's whole key and
 * the slot the record goes to — hence the wave that will read it (slot / records-per-wave) — and adds one to that wave's
 * counter of the next digit. The first pass gets its counts from a sweep over the positions in order (sequential LDS
 * reads: conflict-free). Per record and pass: one key gather instead of two. The counters are u16 (a wave holds at most
 * 65 535 / waves records), two arrays in the space of cr_lz2_pass's one u32 array, and are bumped with 32-bit LDS atomics
 * on the half they live in. Records are positions RELATIVE to `first`. */
CR_DEV void cr_h16_add(uint16_t* h, uint32_t idx) { atomicAdd(reinterpret_cast<uint32_t*>(h) + (idx >> 1), 1u << ((idx & 1u) * 16u)); }

struct CrLz2Plan { uint32_t per, magic; };              /* records per wave (whole chunks); slot / per == umulhi(slot, magic) for slot < 65 536 */
CR_DEV CrLz2Plan cr_lz2_plan(uint32_t count) {
    const uint32_t nw = blockDim.x >> 6;
    CrLz2Plan P;
    P.per = ((count + nw - 1u) / nw + 63u) & ~63u;
    if (P.per == 0u) P.per = 64u;
    P.magic = 0xFFFFFFFFu / P.per + 1u;
    return P;
}
CR_DEV uint16_t* cr_lz2_hist16(const CrLz2Shared& S, uint32_t which) { return reinterpret_cast<uint16_t*>(S.hist) + which * (blockDim.x >> 6) * 256u; }

/* per-wave counts in `h` -> the slot where each wave's first record of each digit goes (digit-major, wave 0's records first:
 * stable); `zero` (the other counter array, or nullptr) is cleared for the counts of the next digit */
CR_DEV void cr_lz2_scan16(const CrLz2Shared& S, uint16_t* h, uint16_t* zero) {
    const uint32_t nw = blockDim.x >> 6, lane = cr_lane(), w = cr_wave_id();
    if (zero) for (uint32_t i = threadIdx.x; i < nw * 128u; i += blockDim.x) reinterpret_cast<uint32_t*>(zero)[i] = 0u;
    if (threadIdx.x < 256u) {
        uint32_t run = 0;
        for (uint32_t v = 0; v < nw; v++) {
            const uint32_t c = h[v * 256u + threadIdx.x];
            h[v * 256u + threadIdx.x] = (uint16_t)run;
            run += c;
        }
        S.base[threadIdx.x] = run;
    }
    __syncthreads();
    if (w == 0) {
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
        for (uint32_t v = 0; v < nw; v++) h[v * 256u + threadIdx.x] = (uint16_t)(h[v * 256u + threadIdx.x] + bs);
    }
    __syncthreads();
}

/* counts of the first digit when the records are the positions in order: every wave counts its own range */
template <class KeyFn>
CR_DEV void cr_lz2_count_first(const KeyFn& key, uint32_t first, uint32_t count, const CrLz2Plan& P, uint32_t mask, uint16_t* cur) {
    const uint32_t lane = cr_lane(), w = cr_wave_id();
    const uint32_t lo = w * P.per < count ? w * P.per : count;
    const uint32_t hi = lo + P.per < count ? lo + P.per : count;
    for (uint32_t k = lane; k < 128u; k += CRGPU_WAVE) reinterpret_cast<uint32_t*>(cur + w * 256u)[k] = 0u;
    cr_lds_order_sw();
    for (uint32_t i = lo + lane; i < hi; i += CRGPU_WAVE) cr_h16_add(cur, w * 256u + (key(first + i) & mask));
    __syncthreads();
}

/* one stable placing pass on the digit (key >> shift) & (2^NB - 1); `cur` holds the slots (cr_lz2_scan16), `nxt` (when nmask
 * != 0) receives the counts of the digit (key >> nshift) & nmask per reading wave. src == nullptr: the positions in order. */
template <int NB, class KeyFn>
CR_DEV void cr_lz2_place(const KeyFn& key, uint32_t first, uint32_t count, const CrLz2Plan& P, uint32_t shift, uint32_t nshift, uint32_t nmask,
                         const uint16_t* src, uint16_t* dst, uint16_t* cur, uint16_t* nxt) {
    const uint32_t lane = cr_lane(), w = cr_wave_id();
    const uint32_t lo = w * P.per < count ? w * P.per : count;
    const uint32_t hi = lo + P.per < count ? lo + P.per : count;
    uint16_t* const my = cur + w * 256u;
    uint32_t r_n = 0, k_n = 0;
    if (lo + lane < hi) { r_n = src ? (uint32_t)src[lo + lane] : lo + lane; k_n = key(first + r_n); }
    for (uint32_t i0 = lo; i0 < hi; i0 += CRGPU_WAVE) {
        const uint32_t i = i0 + lane;
        const bool act = i < hi;
        const uint32_t r = r_n, k = k_n;
        if (i + CRGPU_WAVE < hi) { r_n = src ? (uint32_t)src[i + CRGPU_WAVE] : i + CRGPU_WAVE; k_n = key(first + r_n); }
        const uint32_t dg = (k >> shift) & ((1u << NB) - 1u);
        const u64 same = cr_same_key_mask<NB>(dg, act);
        const u64 lower = same & ((1ull << lane) - 1ull);
        uint32_t at = 0;
        if (act) {
            at = my[dg];
            const uint32_t slot = at + (uint32_t)__builtin_popcountll(lower);
            dst[slot] = (uint16_t)r;
            if (nmask) cr_h16_add(nxt, __umulhi(slot, P.magic) * 256u + ((k >> nshift) & nmask));
        }
        cr_lds_order_sw();
        if (act && (same >> lane) >> 1 == 0ull) my[dg] = (uint16_t)(at + (uint32_t)__builtin_popcountll(same));   /* the group's last lane */
        cr_lds_order_sw();
    }
    __syncthreads();
}

/* the passes of a sort of `bits` key bits, least significant digit first; the counts of the first digit are in counter array 0
 * (slots not yet taken). from == nullptr: the positions in order. Passes write to0, to1, to0, ...; returns the last one's. */
template <class KeyFn>
CR_DEV const uint16_t* cr_lz2_passes(const CrLz2Shared& S, const KeyFn& key, uint32_t first, uint32_t count, const CrLz2Plan& P, uint32_t bits,
                                     const uint16_t* from, uint16_t* to0, uint16_t* to1) {
    uint16_t* cur = cr_lz2_hist16(S, 0);
    uint16_t* nxt = cr_lz2_hist16(S, 1);
    const uint16_t* src = from;
    uint16_t* dst = to0;
    for (uint32_t shift = 0; shift < bits; shift += 8u) {
        const uint32_t left = bits - shift;                      /* key bits of this and the later passes */
        const uint32_t nleft = left > 8u ? left - 8u : 0u;
        const uint32_t nmask = nleft == 0u ? 0u : nleft >= 8u ? 255u : (1u << nleft) - 1u;
        cr_lz2_scan16(S, cur, nmask ? nxt : nullptr);
        if (left > 4u) cr_lz2_place<8>(key, first, count, P, shift, shift + 8u, nmask, src, dst, cur, nxt);
        else cr_lz2_place<4>(key, first, count, P, shift, shift + 8u, nmask, src, dst, cur, nxt);
        src = dst;
        dst = dst == to0 ? to1 : to0;
        uint16_t* t = cur; cur = nxt; nxt = t;
    }
    return src;
}

/* the sorted records -> out(p, q): q = the record to the left when its key is the same (the largest earlier position of the
 * key), else CR_LZ2_NONE. A wave walks its own range; the left neighbour's key comes from the lane below (DPP), one gather
 * per record. */
#define CR_LZ2_NONE 0xFFFFFFFFu
/* `same` (optional, u64[ceil(count / 64)] in LDS): bit i = record i has the key of record i - 1, for callers that go on to walk
 * the runs of equal keys in the sorted records (crgpu_rolz3.h) */
template <class KeyFn, class OutFn>
CR_DEV void cr_lz2_neighbours(const KeyFn& key, uint32_t first, uint32_t count, const CrLz2Plan& P, const uint16_t* cur, const OutFn& out, u64* same = nullptr) {
    const uint32_t lane = cr_lane(), w = cr_wave_id();
    const uint32_t lo = w * P.per < count ? w * P.per : count;
    const uint32_t hi = lo + P.per < count ? lo + P.per : count;
    uint32_t carry_r = 0, carry_k = 0;
    if (lo > 0u && lo < hi) { carry_r = cur[lo - 1u]; carry_k = key(first + carry_r); }
    uint32_t r_n = 0, k_n = 0;
    if (lo + lane < hi) { r_n = cur[lo + lane]; k_n = key(first + r_n); }
    for (uint32_t i0 = lo; i0 < hi; i0 += CRGPU_WAVE) {
        const uint32_t i = i0 + lane;
        const uint32_t r = r_n, k = k_n;
        if (i + CRGPU_WAVE < hi) { r_n = cur[i + CRGPU_WAVE]; k_n = key(first + r_n); }
        const uint32_t rl = cr_shift_up1(r, carry_r), kl = cr_shift_up1(k, carry_k);
        carry_r = cr_lane_get(r, 63); carry_k = cr_lane_get(k, 63);
        const bool eq = i < hi && i > 0u && kl == k;
        if (i < hi) out(first + r, eq ? first + rl : CR_LZ2_NONE);
        if (same) { const u64 em = cr_ballot(eq); if (lane == 0u) same[i0 >> 6] = em; }     /* P.per is whole chunks: i0 is a multiple of 64 */
    }
    __syncthreads();
}

/* "The previous position with the same key" for the positions first .. first + count - 1 (count <= CR_LZ2_MAXN): sorts them by
 * key (`bits` key bits, stable) and calls out(p, q) for every position p with q = the largest earlier position of equal key,
 * or out(p, CR_LZ2_NONE). Returns the buffer that holds the sorted records (the other one is free by then). */
template <class KeyFn, class OutFn>
CR_DEV const uint16_t* cr_lz2_prev_same(const CrLz2Shared& S, const KeyFn& key, uint32_t first, uint32_t count, uint32_t bits,
                                        uint16_t* buf0, uint16_t* buf1, const OutFn& out) {
    const CrLz2Plan P = cr_lz2_plan(count);
    cr_lz2_count_first(key, first, count, P, bits >= 8u ? 255u : (1u << bits) - 1u, cr_lz2_hist16(S, 0));
    const uint16_t* cur = cr_lz2_passes(S, key, first, count, P, bits, nullptr, buf0, buf1);
    cr_lz2_neighbours(key, first, count, P, cur, out);
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
        else {                                           /* the last bytes: sixteen loads at once (clamped index), not sixteen round trips behind `if` */
            uint32_t t[16];
#pragma unroll
            for (uint32_t k = 0; k < 16u; k++) t[k] = g[i + k < n ? i + k : n - 1u];
#pragma unroll
            for (uint32_t k = 0; k < 16u; k++) if (i + k >= n) t[k] = 0u;
            v.x = t[0] | t[1] << 8 | t[2] << 16 | t[3] << 24; v.y = t[4] | t[5] << 8 | t[6] << 16 | t[7] << 24;
            v.z = t[8] | t[9] << 8 | t[10] << 16 | t[11] << 24; v.w = t[12] | t[13] << 8 | t[14] << 16 | t[15] << 24;
        }
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
    for (uint32_t p0 = CR_LZP_SKIP + threadIdx.x; p0 < limit; p0 += 4u * blockDim.x) {
      /* the candidates of four positions are fetched at once (clamped index): one at a time behind the loop's `if` every round
       * was a memory round trip, 44 of them per thread */
      uint32_t c8s[4], c4s[4];
#pragma unroll
      for (uint32_t u = 0; u < 4u; u++) {
          const uint32_t p = p0 + u * blockDim.x, q = (p < limit ? p : limit - 1u) - CR_LZP_SKIP;
          c8s[u] = g8[q]; c4s[u] = g4[q];
      }
#pragma unroll
      for (uint32_t u = 0; u < 4u; u++) {
        const uint32_t p = p0 + u * blockDim.x;
        if (p >= limit) break;
        const u64 x = cr_lz2_read8(d, p - 8u);
        const uint32_t c8 = c8s[u], c4 = c4s[u], c2 = l2[p - CR_LZP_SKIP];
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
}


/* ==== round 4: blocks of up to 65 537 bytes (north_star's 64 KiB datablock, + 1 for the dictionary stage's flag byte) ==========
 * The block itself takes 64 KB of the CU's 160, so two u16 record buffers can hold 19 200 records, not 65 528. The positions
 * are therefore sorted in GROUPS BY KEY: a digit mixed from the whole key (equal keys -> equal digit) cuts them into 256 bins,
 * consecutive bins are packed into groups of at most CR_LZ3_CAP (19 200) positions, and every group is compacted (in position order)
 * and sorted in LDS against the staged block, exactly like a small block — "previous position of the same key" never leaves a
 * group. A block with a bin above the capacity (a block of one repeated byte: one key) is left to the table sweep. */
#define CR_LZ3_MAXN      65537u
#define CR_LZ3_CAP       19200u          /* (with the block: 151 616 bytes of dynamic LDS + ~1 KB static — what the 28 KiB kernels take, so that
                                          * a launch can be placed on a CU that holds six one-wave decoder workgroups: DESIGN.md 3.6) */
#define CR_LZ3_GROUPS    16u
#define CR_LZ3_SRC_BYTES (CR_LZ3_MAXN + 63u)
#define CR_LZ3_LDS_BYTES (2u * CR_LZ3_CAP * 2u + CR_LZ3_SRC_BYTES + (CR_LZ2_THREADS / 64u) * 256u * 4u + 256u * 4u)

struct CrLz3Groups {                     /* static LDS of a kernel that sorts in groups */
    uint8_t  binmap[256];                /* bin -> group */
    uint16_t woff[CR_LZ2_MAX_WAVES][CR_LZ3_GROUPS];   /* where wave w starts writing group g's positions */
    uint32_t gsize[CR_LZ3_GROUPS];
    uint32_t ngroups;                    /* 0: the block does not fit (a bin above the capacity, or too many groups) */
    uint32_t maxbin, step;               /* scratch of cr_lz3_cut */
};

CR_DEV CrLz2Shared cr_lz3_carve(uint8_t* lds, uint32_t waves) {
    CrLz2Shared S;
    S.a = reinterpret_cast<uint16_t*>(lds);
    S.b = S.a + CR_LZ3_CAP;
    S.hist = reinterpret_cast<uint32_t*>(S.b + CR_LZ3_CAP);
    S.base = S.hist + waves * 256u;
    S.src = reinterpret_cast<uint8_t*>(S.base + 256u);
    return S;
}
CR_DEV uint32_t cr_lz3_bin(uint32_t k) { return (k * 2654435761u) >> 24; }     /* the whole key decides the bin: equal keys share it, a handful of keys rarely do */

/* per-wave bin counts (u16[waves][256], each wave its own range of the positions in order) -> groups of at most `cap` and where
 * every wave starts writing in every group (G.ngroups == 0: does not fit). Every thread calls this behind a barrier. */
CR_DEV void cr_lz3_cut(const CrLz2Shared& S, CrLz3Groups& G, const uint16_t* bins, uint32_t cap) {
    const uint32_t nw = blockDim.x >> 6, lane = cr_lane(), w = cr_wave_id();
    if (threadIdx.x == 0) G.maxbin = 0u;
    if (threadIdx.x < CR_LZ3_GROUPS) G.gsize[threadIdx.x] = 0u;
    __syncthreads();
    uint32_t tot = 0;
    if (threadIdx.x < 256u) {
        for (uint32_t v = 0; v < nw; v++) tot += bins[v * 256u + threadIdx.x];
        S.base[threadIdx.x] = tot;
        atomicMax(&G.maxbin, tot);
    }
    __syncthreads();
    if (w == 0) {                                               /* S.base <- the positions in front of each bin */
        uint32_t carry = 0;
        for (uint32_t k0 = 0; k0 < 256u; k0 += CRGPU_WAVE) {
            const uint32_t v = S.base[k0 + lane];
            const uint32_t incl = cr_scan_incl(v);
            S.base[k0 + lane] = carry + incl - v;
            carry += cr_lane_get(incl, 63);
        }
        /* Bins are packed by where they start: group = start / step with step = cap - (largest bin) + 1, so that a group —
         * the bins that start inside one step — holds fewer than step + largest bin = cap + 1 positions. No serial walk. */
        if (lane == 0) {
            const uint32_t mx = G.maxbin;
            uint32_t ng = 0, step = 1;
            if (mx <= cap) { step = cap - mx + 1u; ng = carry / step + 1u; if (ng > CR_LZ3_GROUPS) ng = 0; }
            G.step = step;
            G.ngroups = ng;
        }
    }
    __syncthreads();
    if (G.ngroups == 0u) return;
    if (threadIdx.x < 256u) {
        const uint32_t g = S.base[threadIdx.x] / G.step;
        G.binmap[threadIdx.x] = (uint8_t)g;
        if (tot) atomicAdd(&G.gsize[g], tot);
    }
    __syncthreads();
    /* where every wave starts writing in every group: thread t adds its bin's per-wave counts to the bin's group */
    uint32_t* const acc = S.base;                               /* u32[waves][CR_LZ3_GROUPS] (the totals are no longer needed) */
    for (uint32_t i = threadIdx.x; i < nw * CR_LZ3_GROUPS; i += blockDim.x) acc[i] = 0u;
    __syncthreads();
    if (threadIdx.x < 256u) {
        const uint32_t g = G.binmap[threadIdx.x];
        for (uint32_t v = 0; v < nw; v++) { const uint32_t c = bins[v * 256u + threadIdx.x]; if (c) atomicAdd(acc + v * CR_LZ3_GROUPS + g, c); }
    }
    __syncthreads();
    if (threadIdx.x < CR_LZ3_GROUPS) {
        uint32_t run = 0;
        for (uint32_t v = 0; v < nw; v++) { G.woff[v][threadIdx.x] = (uint16_t)run; run += acc[v * CR_LZ3_GROUPS + threadIdx.x]; }
    }
    __syncthreads();
}

/* the bins of the positions first .. first + count - 1 -> groups */
template <class KeyFn>
CR_DEV void cr_lz3_groups(const CrLz2Shared& S, CrLz3Groups& G, const KeyFn& key, uint32_t first, uint32_t count, const CrLz2Plan& PA, uint32_t cap = CR_LZ3_CAP) {
    const uint32_t lane = cr_lane(), w = cr_wave_id();
    uint16_t* const bins = cr_lz2_hist16(S, 0);                 /* u16[waves][256]: a wave's own range, as the compaction will walk it */
    const uint32_t lo = w * PA.per < count ? w * PA.per : count;
    const uint32_t hi = lo + PA.per < count ? lo + PA.per : count;
    for (uint32_t k = lane; k < 128u; k += CRGPU_WAVE) reinterpret_cast<uint32_t*>(bins + w * 256u)[k] = 0u;
    cr_lds_order_sw();
    for (uint32_t i = lo + lane; i < hi; i += CRGPU_WAVE) cr_h16_add(bins, w * 256u + cr_lz3_bin(key(first + i)));
    __syncthreads();
    cr_lz3_cut(S, G, bins, cap);
}

/* cr_lz2_prev_same for up to CR_LZ3_MAXN positions. Returns false (nothing called) when the block does not fit the groups. */
#ifdef CR_LZ3_PROF      /* diagnostic build: 100 MHz ticks per phase, added up over tables and groups (tools/lzp64_profile.py) */
#define CR_LZ3_MARK(st_, slot_) do { if ((st_) && threadIdx.x == 0) { const u64 now_ = wall_clock64(); (st_)[slot_] += now_ - (st_)[15]; (st_)[15] = now_; } } while (0)
#else
#define CR_LZ3_MARK(st_, slot_) do { } while (0)
#endif
struct CrLz3NoGroupFn { CR_DEV void operator()(const uint16_t*, const u64*, uint16_t*, uint32_t) const {} };
/* grp(cur, same, spare, m): called by every thread once per group while its m sorted records (position - first, u16) are still
 * in LDS, with the `same` bits of cr_lz2_neighbours (they lie in the digit counters, idle between a group's last pass and the
 * next group's compaction) and the other record buffer to use — for work that wants a key's positions side by side */
template <class KeyFn, class OutFn, class GroupFn = CrLz3NoGroupFn>
CR_DEV bool cr_lz3_prev_same(const CrLz2Shared& S, CrLz3Groups& G, const KeyFn& key, uint32_t first, uint32_t count, uint32_t bits, const OutFn& out, u64* st = nullptr,
                             const GroupFn* grp = nullptr) {
    const uint32_t lane = cr_lane(), w = cr_wave_id(), nw = blockDim.x >> 6;
    const CrLz2Plan PA = cr_lz2_plan(count);
    cr_lz3_groups(S, G, key, first, count, PA);
    CR_LZ3_MARK(st, 1);
    const uint32_t ng = G.ngroups;
    if (ng == 0u) return false;
    const uint32_t lo = w * PA.per < count ? w * PA.per : count;
    const uint32_t hi = lo + PA.per < count ? lo + PA.per : count;
    uint16_t* const h0 = cr_lz2_hist16(S, 0);
    for (uint32_t g = 0; g < ng; g++) {
        const uint32_t m = G.gsize[g];
        if (m == 0u) continue;
        const CrLz2Plan P = cr_lz2_plan(m);
        for (uint32_t i = threadIdx.x; i < nw * 128u; i += blockDim.x) reinterpret_cast<uint32_t*>(h0)[i] = 0u;
        __syncthreads();
        /* the group's positions in order into buffer a, the counts of their first digit per reading wave into h0 */
        {
            uint32_t at = G.woff[w][g];
            uint32_t k_n = 0;
            if (lo + lane < hi) k_n = key(first + lo + lane);
            for (uint32_t i0 = lo; i0 < hi; i0 += CRGPU_WAVE) {
                const uint32_t i = i0 + lane;
                const uint32_t k = k_n;
                if (i + CRGPU_WAVE < hi) k_n = key(first + i + CRGPU_WAVE);
                const bool act = i < hi && G.binmap[cr_lz3_bin(k)] == g;
                const u64 am = cr_ballot(act);
                if (act) {
                    const uint32_t slot = at + (uint32_t)__builtin_popcountll(am & ((1ull << lane) - 1ull));
                    S.a[slot] = (uint16_t)i;
                    cr_h16_add(h0, __umulhi(slot, P.magic) * 256u + (k & 255u));
                }
                at += (uint32_t)__builtin_popcountll(am);
            }
        }
        __syncthreads();
        CR_LZ3_MARK(st, 2);
        const uint16_t* cur = cr_lz2_passes(S, key, first, m, P, bits, S.a, S.b, S.a);
        CR_LZ3_MARK(st, 3);
        if (grp) {
            u64* const same = reinterpret_cast<u64*>(S.hist);           /* CR_LZ3_CAP bits = 2 400 bytes of the counters' 8 KB */
            cr_lz2_neighbours(key, first, m, P, cur, out, same);
            CR_LZ3_MARK(st, 4);
            (*grp)(cur, same, cur == S.a ? S.b : S.a, m);
            __syncthreads();
            CR_LZ3_MARK(st, 7);
            continue;
        } else {
            cr_lz2_neighbours(key, first, m, P, cur, out);
        }
        CR_LZ3_MARK(st, 4);
    }
    return true;
}

/* comprop's LZP pre-pass for a block of CR_LZ2_MAXN < n <= CR_LZ3_MAXN bytes (matcher_getpos / matcher_lookup,
 * ropmain/cr-matcher.c:59-89, for every position at once: crgpu_lzp.h). The three tables are sorted one after the other; a
 * table's answer is verified where it is found (the block is in LDS) and the position's source — lzp2's candidate, replaced
 * by lzp4's where the four bytes agree, replaced by lzp8's where the eight bytes agree: the order of cr-matcher.c:66-72 —
 * is kept as u16 per position in global scratch, because 65 528 answers do not fit beside the block. blockDim.x ==
 * CR_LZ2_THREADS; returns false when the block has to go to the table sweep. */
CR_DEV bool cr_lzp_block_lds64(const CrLz2Shared& S, CrLz3Groups& G, const CrLzpScratch& sc, const uint8_t* g, uint32_t n, uint8_t* lens, u64* st = nullptr) {
    if (n <= CR_LZP_TAIL + CR_LZP_SKIP) return true;
#ifdef CR_LZ3_PROF
    if (st && threadIdx.x == 0) { for (int q = 0; q < 15; q++) st[q] = 0; st[15] = wall_clock64(); }
#endif
    const uint32_t limit = n - CR_LZP_TAIL;           /* positions with p + 1024 < n */
    const uint32_t count = limit - CR_LZP_SKIP;
    cr_lz2_stage_block(S, g, n);
    CR_LZ3_MARK(st, 0);
    const uint8_t* d = S.src;
    uint16_t* const from = reinterpret_cast<uint16_t*>((reinterpret_cast<uintptr_t>(sc.c8) + 15u) & ~(uintptr_t)15u);   /* u16[count], by position - 9 */
    CrLzpKeyW<2> key2; key2.d = d;
    CrLzpKeyW<1> key4; key4.d = d;
    CrLzpKeyW<0> key8; key8.d = d;
    if (!cr_lz3_prev_same(S, G, key2, CR_LZP_SKIP, count, 16u, [from](uint32_t p, uint32_t q) { from[p - CR_LZP_SKIP] = (uint16_t)(q == CR_LZ2_NONE ? 2u : q); }, st)) return false;
    cr_wg_sync_global();
    if (!cr_lz3_prev_same(S, G, key4, CR_LZP_SKIP, count, 20u, [from, d](uint32_t p, uint32_t q) {
            const uint32_t c = q == CR_LZ2_NONE ? 4u : q;
            if ((uint32_t)cr_lz2_read8(d, c - 4u) == (uint32_t)cr_lz2_read8(d, p - 4u)) from[p - CR_LZP_SKIP] = (uint16_t)c;
        }, st)) return false;
    cr_wg_sync_global();
    if (!cr_lz3_prev_same(S, G, key8, CR_LZP_SKIP, count, 24u, [from, d](uint32_t p, uint32_t q) {
            const uint32_t c = q == CR_LZ2_NONE ? 8u : q;
            if (cr_lz2_read8(d, c - 8u) == cr_lz2_read8(d, p - 8u)) from[p - CR_LZP_SKIP] = (uint16_t)c;
        }, st)) return false;
    cr_wg_sync_global();
    CR_LZ3_MARK(st, 5);
    for (uint32_t p0 = CR_LZP_SKIP + threadIdx.x; p0 < limit; p0 += 4u * blockDim.x) {      /* (four sources fetched at once, clamped index) */
        uint32_t srcs[4];
#pragma unroll
        for (uint32_t u = 0; u < 4u; u++) { const uint32_t p = p0 + u * blockDim.x; srcs[u] = from[(p < limit ? p : limit - 1u) - CR_LZP_SKIP]; }
#pragma unroll
        for (uint32_t u = 0; u < 4u; u++) {
            const uint32_t p = p0 + u * blockDim.x;
            if (p >= limit) break;
            const uint32_t src = srcs[u];
            const uint32_t len = src ? cr_lz2_common_len(d, src, p) : 0u;    /* matcher_lookup, cr-matcher.c:75-89 */
            lens[p] = (uint8_t)(len < CR_LZP_MIN ? 1u : len);
        }
    }
    CR_LZ3_MARK(st, 6);
    return true;
}

#endif

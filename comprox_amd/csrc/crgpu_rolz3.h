/*
 * comprox_amd/csrc/crgpu_rolz3.h — comprolz's parse result at every position with the block, its ring links and the
 * plain lookups all in LDS (kernel k_rolz_match_lds, 16 waves per datablock, blocks of up to 28 672 bytes).
 *
 * Reference: /root/reference/src/rolzmain/cr-matcher.c:43-197 (matcher_init, matcher_update, match, matcher_lookup).
 * Same answers as k_rolz_match (crgpu_rolz.h explains why the parse result at every position is a pure function of the
 * data); what changes is where the work happens. k_rolz_match keeps the newest position of each of the 262 144 rings in
 * a 1 MB table per block (swept in position order through HBM) and chases up to 64 ring links + 16 row links per
 * position through global arrays: 10.4 ms and 19.9 GB of traffic on the bench shard, 7.6 ms of it the searches. Here
 *   - ring links = "previous position with the same ring number": the positions sorted by ring number in LDS
 *     (crgpu_lzp2.h, cr_lz2_prev_same: stable radix sort of u16 position records), links kept in LDS as u16;
 *   - row links the same way with the byte in front as the key (one pass), written to the global row array;
 *   - the searches read links, the block and the plain lookups of the look-ahead positions from LDS.
 * LDS: two u16[28 672] buffers (sort ping-pong; afterwards ring links | plain lookups), the block, digit counts: 157 KB.
 */
#ifndef CRGPU_ROLZ3_H
#define CRGPU_ROLZ3_H

#include "crgpu_rolz.h"
#include "crgpu_lzp2.h"

struct CrRolzRingKey {
    const uint8_t* d;          /* the block in LDS */
    bool ctx4;
    CR_DEV uint32_t operator()(uint32_t p) const { return cr_rolz_ring_of(d, p, ctx4); }
};
struct CrRolzRowKey {
    const uint8_t* d;
    CR_DEV uint32_t operator()(uint32_t p) const { return cr_rolz_row_of(d, p); }
};

/* what the searches read: everything in LDS but the row links */
struct CrRolzLds {
    const uint8_t*  d;         /* the block */
    const uint16_t* link;      /* ring links, 0xffff = none */
    const uint32_t* row_prev;  /* global */
    uint8_t*        raw_rank;  /* plain lookup of every position (0xff = none) */
    uint8_t*        raw_len;
};
CR_DEV uint32_t cr_rolz3_link(const CrRolzLds& M, uint32_t p) { const uint32_t v = M.link[p]; return v == 0xffffu ? CR_ROLZ_NONE : v; }

/* match(), cr-matcher.c:93-124 — cr_rolz_ring_search on the LDS copies. The next link is read before the entry is
 * looked at: the chain of link reads is the critical path, the byte tests hang off it. */
CR_DEV void cr_rolz3_ring_search(const CrRolzLds& M, uint32_t pos, uint32_t start, uint32_t floor, uint32_t& rank, uint32_t& len) {
    rank = CR_ROLZ_NONE; len = CR_ROLZ_MIN - 1u;
    uint32_t q = start;
    while (q != CR_ROLZ_NONE && q >= floor) q = cr_rolz3_link(M, q);
    const uint32_t first = M.d[pos];
    uint32_t beyond = M.d[pos + len];                  /* an entry can only be strictly longer if it also agrees at offset `len` */
    for (uint32_t i = 0; i < CR_ROLZ_RING && len < CR_ROLZ_MAX && q != CR_ROLZ_NONE; i++) {
        const uint32_t qn = cr_rolz3_link(M, q);
        if (M.d[q] == first && M.d[q + len] == beyond) {
            const uint32_t j = cr_lz2_common_len(M.d, q, pos);
            if (j > len) { rank = i; len = j; beyond = M.d[pos + len]; }
        }
        q = qn;
    }
    if (len < CR_ROLZ_MIN) { rank = CR_ROLZ_NONE; len = 1; }
}
/* the same for TWO positions at once (either may be inactive: start == CR_ROLZ_NONE and pos = any valid position): the
 * searches are chains of dependent LDS reads (link -> two bytes -> next link), so a lane that walks two rings in step
 * has two of them in flight. Same result per position as cr_rolz3_ring_search with floor = pos. */
CR_DEV void cr_rolz3_ring_search2(const CrRolzLds& M, uint32_t posA, uint32_t startA, uint32_t posB, uint32_t startB,
                                  uint32_t& rankA, uint32_t& lenA, uint32_t& rankB, uint32_t& lenB) {
    rankA = CR_ROLZ_NONE; lenA = CR_ROLZ_MIN - 1u; rankB = CR_ROLZ_NONE; lenB = CR_ROLZ_MIN - 1u;
    uint32_t qa = startA, qb = startB;                                   /* (the newest entry of a position's own ring lies below it) */
    const uint32_t firstA = M.d[posA], firstB = M.d[posB];
    uint32_t beyondA = M.d[posA + lenA], beyondB = M.d[posB + lenB];
    uint32_t ia = 0, ib = 0;
    bool goA = qa != CR_ROLZ_NONE, goB = qb != CR_ROLZ_NONE;
    while (goA || goB) {
        uint32_t qna = CR_ROLZ_NONE, qnb = CR_ROLZ_NONE, ha = 0, hb = 0, ta = 0, tb = 0;
        if (goA) { qna = cr_rolz3_link(M, qa); ha = M.d[qa]; ta = M.d[qa + lenA]; }
        if (goB) { qnb = cr_rolz3_link(M, qb); hb = M.d[qb]; tb = M.d[qb + lenB]; }
        if (goA) {
            if (ha == firstA && ta == beyondA) {
                const uint32_t j = cr_lz2_common_len(M.d, qa, posA);
                if (j > lenA) { rankA = ia; lenA = j; beyondA = M.d[posA + lenA]; }
            }
            qa = qna; ia++;
            goA = ia < CR_ROLZ_RING && lenA < CR_ROLZ_MAX && qa != CR_ROLZ_NONE;
        }
        if (goB) {
            if (hb == firstB && tb == beyondB) {
                const uint32_t j = cr_lz2_common_len(M.d, qb, posB);
                if (j > lenB) { rankB = ib; lenB = j; beyondB = M.d[posB + lenB]; }
            }
            qb = qnb; ib++;
            goB = ib < CR_ROLZ_RING && lenB < CR_ROLZ_MAX && qb != CR_ROLZ_NONE;
        }
    }
    if (lenA < CR_ROLZ_MIN) { rankA = CR_ROLZ_NONE; lenA = 1; }
    if (lenB < CR_ROLZ_MIN) { rankB = CR_ROLZ_NONE; lenB = 1; }
}
CR_DEV void cr_rolz3_ahead(const CrRolzLds& M, uint32_t at, uint32_t floor, uint32_t& rank, uint32_t& len) {
    const uint32_t newest = cr_rolz3_link(M, at);
    if (newest == CR_ROLZ_NONE || newest < floor) {
        rank = M.raw_rank[at] == 0xffu ? CR_ROLZ_NONE : M.raw_rank[at];
        len = M.raw_len[at];
    } else {
        cr_rolz3_ring_search(M, at, newest, floor, rank, len);
    }
}

/* matcher_lookup for every position, cr-matcher.c:126-197 — cr_rolz_find_all on the LDS copies; T.rank / T.len out */
CR_DEV void cr_rolz3_find_all(const CrRolzLds& M, uint32_t n, uint32_t link_limit, bool flexible, const CrRolzTables& T, u64* st = nullptr) {
    const uint32_t limit = n - CR_ROLZ_TAIL;              /* positions with p + 1024 < n */
    for (uint32_t p = CR_ROLZ_WARM + threadIdx.x; p < link_limit; p += 2u * blockDim.x) {
        const uint32_t p2 = p + blockDim.x;
        const bool two = p2 < link_limit;
        uint32_t rank, len, rank2, len2;
        cr_rolz3_ring_search2(M, p, cr_rolz3_link(M, p), two ? p2 : p, two ? cr_rolz3_link(M, p2) : CR_ROLZ_NONE, rank, len, rank2, len2);
        M.raw_rank[p] = (uint8_t)(rank == CR_ROLZ_NONE ? 0xffu : rank);
        M.raw_len[p] = (uint8_t)len;
        if (two) {
            M.raw_rank[p2] = (uint8_t)(rank2 == CR_ROLZ_NONE ? 0xffu : rank2);
            M.raw_len[p2] = (uint8_t)len2;
        }
    }
    __syncthreads();
    cr_wg_stamp(st, 4);
    for (uint32_t p = CR_ROLZ_WARM + threadIdx.x; p < limit; p += blockDim.x) {
        uint32_t rank = M.raw_rank[p] == 0xffu ? CR_ROLZ_NONE : M.raw_rank[p], len = M.raw_len[p];
        const bool fell_short = len < CR_ROLZ_MIN;
        if (flexible && !fell_short) {                    /* -f (:143-167): cut where "this match + what follows" prices best */
            uint32_t best = 0, keep = len;
            for (uint32_t i = len; i >= 1u; i--) {
                uint32_t r2, l2;
                cr_rolz3_ahead(M, p + i, p, r2, l2);
                const uint32_t v = cr_rolz_price(rank, i) + cr_rolz_price(r2, l2);
                if (i == len) best = v;
                else if (v > best) { keep = i; best = v; }
            }
            len = keep;
            if (len < CR_ROLZ_MIN) { rank = CR_ROLZ_NONE; len = 1; }
        }
        if (fell_short) {                                 /* the 16 newest positions behind the same byte (:171-186) */
            len = CR_ROLZ_MIN - 1u; rank = CR_ROLZ_NONE;
            uint32_t q = M.row_prev[p];
            for (uint32_t i = 0; i < CR_ROLZ_ROW; i++) {
                const uint32_t at = q == CR_ROLZ_NONE ? 0u : q;          /* the reference's rows are zero-filled */
                const uint32_t j = cr_lz2_common_len(M.d, at, p);
                if (j > len) { rank = CR_ROLZ_RING + i; len = j; }
                if (q != CR_ROLZ_NONE) q = M.row_prev[q];
            }
            if (len < CR_ROLZ_MIN) { rank = CR_ROLZ_NONE; len = 1; }
        }
        if ((!flexible || fell_short) && len > 1u) {      /* lazy evaluation (:188-196): looks ahead without feeding */
            const uint32_t mine = cr_rolz_price(rank, len);
            for (uint32_t i = 1; i < CR_ROLZ_MIN; i++) {
                uint32_t r2, l2;
                cr_rolz3_ahead(M, p + i, p, r2, l2);
                if (cr_rolz_price(r2, l2) > mine + i * CR_ROLZ_RING) { rank = CR_ROLZ_NONE; len = 1; break; }
            }
        }
        T.rank[p] = (uint8_t)(rank == CR_ROLZ_NONE ? 0xffu : rank);
        T.len[p] = (uint8_t)len;
    }
}

/* CR_ROLZ_TAIL + CR_ROLZ_WARM < n <= CR_LZ2_MAXN; every thread calls this with the same arguments */
CR_DEV void cr_rolz_match_block_lds(const CrLz2Shared& S, const uint8_t* g, uint32_t n, bool flexible, const CrRolzTables& T, u64* st = nullptr) {
    const bool ctx4 = false;                              /* using_ctx4 needs 4 MiB blocks, cr-coder.c:158 */
    const uint32_t link_limit = n - CR_ROLZ_TAIL + (flexible ? CR_ROLZ_MAX + 1u : CR_ROLZ_MIN);
    cr_wg_stamp(st, 0);                                  /* stamps (100 MHz): staged | row links | ring links | plain lookups | parse */
    cr_lz2_stage_block(S, g, n);
    cr_wg_stamp(st, 1);
    const uint32_t count = link_limit - CR_ROLZ_WARM;
    /* row links first (one pass, byte in front as the key): gathered in the free record buffer, they leave for the global row
     * array as coalesced stores (crgpu_lzp2.h, cr_lz2_prev_same_lds) before the ring sort needs both buffers */
    CrRolzRowKey wk; wk.d = S.src;
    uint32_t* const rows = T.row_prev;
    {
        const uint16_t* const rw = cr_lz2_prev_same_lds(S, wk, CR_ROLZ_WARM, count, 8u, 0xffffu);
        for (uint32_t i = threadIdx.x; i < count; i += blockDim.x) { const uint32_t q = rw[i]; rows[CR_ROLZ_WARM + i] = q == 0xffffu ? CR_LZ2_NONE : q; }
    }
    cr_wg_stamp(st, 2);
    /* ring links -> the buffer the sort leaves free, where the searches read them */
    CrRolzRingKey rk; rk.d = S.src; rk.ctx4 = ctx4;
    uint16_t* links = nullptr;
    {
        /* the sort's last output buffer is known in advance: 18 key bits = three passes -> a, b, a */
        links = S.b;
        uint16_t* const lk = links;
        cr_lz2_prev_same(S, rk, CR_ROLZ_WARM, count, 18u, S.a, S.b, [lk](uint32_t p, uint32_t q) { lk[p] = (uint16_t)(q == CR_LZ2_NONE ? 0xffffu : q); });
    }
    cr_wg_sync_global();
    cr_wg_stamp(st, 3);
    CrRolzLds M;
    M.d = S.src; M.link = links; M.row_prev = T.row_prev;
    M.raw_rank = reinterpret_cast<uint8_t*>(S.a);
    M.raw_len = M.raw_rank + CR_LZ2_MAXN;
    cr_rolz3_find_all(M, n, link_limit, flexible, T, st);
    __syncthreads();
    cr_wg_stamp(st, 5);
}


/* match() (cr-matcher.c:93-124) for every position of a group of rings, from the group's SORTED records: a position's ring
 * entries — the up to 64 newest earlier positions of its ring — are the records in front of it, newest nearest, so the search
 * reads them side by side instead of chasing links, and `same` says in one 64-bit window how many there are. What bounds such a
 * search is the LDS pipe (byte gathers at 64 random addresses per instruction: 295 us per block of 36 800 bytes with two of
 * them per entry) and the divergence of the lanes that go on to a full comparison, so 16 bits hashed from each record's first
 * five bytes are laid out in sorted order first (`two`, the spare record buffer): an entry only counts with CR_ROLZ_MIN = 5
 * bytes in common, the entries of a position are then consecutive u16 — conflict-free reads — and hardly any but the ones that
 * count go on to the byte at `len` and the full comparison.
 * raw16[p] = rank (0xff = none) | length << 8, as cr_rolz_find_all's first pass leaves it. */
#ifndef CR_ROLZ3_BATCH
#define CR_ROLZ3_BATCH 8u      /* entries looked at per round */
#endif
CR_DEV uint32_t cr_rolz3_two(const uint8_t* d, uint32_t p) {                  /* 16 bits of the FIVE bytes at p: equal bytes, equal value */
    const u64 x = cr_lz2_read8(d, p) & 0xFFFFFFFFFFull;
    return (uint32_t)((x * 0x9E3779B97F4A7C15ull) >> 48);
}
CR_DEV void cr_rolz3_lay_two(const uint8_t* d, const uint16_t* cur, uint16_t* two, uint32_t m) {
    for (uint32_t i = threadIdx.x; i < m; i += blockDim.x) two[i] = (uint16_t)cr_rolz3_two(d, CR_ROLZ_WARM + cur[i]);
    __syncthreads();
}
CR_DEV uint32_t cr_rolz3_have(const u64* same, uint32_t i) {                 /* records in front of record i with its key, at most 64 */
    const uint32_t wi = i >> 6, bi = i & 63u;
    u64 win = same[wi] << (63u - bi);                                      /* bit 63 = record i, bit 62 = record i - 1, ... */
    if (bi != 63u && wi > 0u) win |= same[wi - 1u] >> (bi + 1u);
    return ~win ? (uint32_t)__builtin_clzll(~win) : 64u;
}
CR_DEV void cr_rolz3_group_lookups(const uint8_t* d, const uint16_t* cur, const u64* same, uint16_t* two, uint32_t m, uint16_t* raw16, u64* st = nullptr) {
    cr_rolz3_lay_two(d, cur, two, m);
    CR_LZ3_MARK(st, 5);
    for (uint32_t i = threadIdx.x; i < m; i += blockDim.x) {
        const uint32_t pos = CR_ROLZ_WARM + cur[i];
        const uint32_t have = cr_rolz3_have(same, i);                       /* at most CR_ROLZ_RING */
        uint32_t rank = CR_ROLZ_NONE, len = CR_ROLZ_MIN - 1u;
        const uint32_t mine = two[i];
        uint32_t beyond = d[pos + len];
        for (uint32_t k0 = 0; k0 < have && len < CR_ROLZ_MAX; k0 += CR_ROLZ3_BATCH) {
            uint32_t hit = 0;
#pragma unroll
            for (uint32_t u = 0; u < CR_ROLZ3_BATCH; u++) hit |= (k0 + u < have && two[i - 1u - k0 - u] == mine ? 1u : 0u) << u;
            while (hit && len < CR_ROLZ_MAX) {
                const uint32_t u = (uint32_t)__builtin_ctz(hit);
                hit &= hit - 1u;
                const uint32_t q = CR_ROLZ_WARM + cur[i - 1u - k0 - u];
                if (d[q + len] != beyond) continue;                        /* an entry can only be strictly longer if it also agrees at offset `len` */
                const uint32_t j = cr_lz2_common_len(d, q, pos);
                if (j > len) { rank = k0 + u; len = j; beyond = d[pos + len]; }
            }
        }
        if (len < CR_ROLZ_MIN) { rank = CR_ROLZ_NONE; len = 1; }
        raw16[pos] = (uint16_t)((rank == CR_ROLZ_NONE ? 0xffu : rank) | (len << 8));
    }
}

/* the row search (cr-matcher.c:171-186) of the positions whose plain lookup fell short, from a group of rows sorted by the byte
 * in front: the 16 newest earlier positions of the row are the 16 records in front; a row with fewer answers with position 0 for
 * the rest (the reference's rows are zero-filled), and as only a strictly longer agreement replaces the best, the first of those
 * stands for all of them. Same filter as the ring searches. row16[p] packed like raw16. */
CR_DEV void cr_rolz3_group_rows(const uint8_t* d, const uint16_t* cur, const u64* same, uint16_t* two, uint32_t m, uint32_t limit, const uint16_t* raw16, uint16_t* row16, u64* st = nullptr) {
    cr_rolz3_lay_two(d, cur, two, m);
    CR_LZ3_MARK(st, 5);
    for (uint32_t i0 = threadIdx.x; i0 < m; i0 += 4u * blockDim.x) {     /* raw16 is global: four positions' answers are fetched per round */
        uint32_t ps[4], rv[4];
#pragma unroll
        for (uint32_t u = 0; u < 4u; u++) {
            const uint32_t i = i0 + u * blockDim.x;
            ps[u] = i < m ? CR_ROLZ_WARM + cur[i] : limit;
            rv[u] = ps[u] < limit ? raw16[ps[u]] : 0xff00u;
        }
#pragma unroll
      for (uint32_t u = 0; u < 4u; u++) {
        const uint32_t i = i0 + u * blockDim.x, pos = ps[u];
        if (pos >= limit || (rv[u] >> 8) >= CR_ROLZ_MIN) continue;
        uint32_t have = cr_rolz3_have(same, i);
        if (have > CR_ROLZ_ROW) have = CR_ROLZ_ROW;
        uint32_t rank = CR_ROLZ_NONE, len = CR_ROLZ_MIN - 1u;
        const uint32_t mine = two[i];
        for (uint32_t k0 = 0; k0 < have; k0 += CR_ROLZ3_BATCH) {
            uint32_t hit = 0;
#pragma unroll
            for (uint32_t u = 0; u < CR_ROLZ3_BATCH; u++) hit |= (k0 + u < have && two[i - 1u - k0 - u] == mine ? 1u : 0u) << u;
            while (hit) {
                const uint32_t u = (uint32_t)__builtin_ctz(hit);
                hit &= hit - 1u;
                const uint32_t j = cr_lz2_common_len(d, CR_ROLZ_WARM + cur[i - 1u - k0 - u], pos);
                if (j > len) { rank = CR_ROLZ_RING + k0 + u; len = j; }
            }
        }
        if (have < CR_ROLZ_ROW) {
            const uint32_t j = cr_lz2_common_len(d, 0u, pos);
            if (j > len) { rank = CR_ROLZ_RING + have; len = j; }
        }
        if (len < CR_ROLZ_MIN) { rank = CR_ROLZ_NONE; len = 1; }
        row16[pos] = (uint16_t)((rank == CR_ROLZ_NONE ? 0xffu : rank) | (len << 8));
      }
    }
}

/* the ring links AND the plain lookups of a block of CR_LZ2_MAXN < n <= CR_LZ3_MAXN bytes by the same sort in groups
 * (crgpu_lzp2.h, round 4): the links go to the array k_rolz_match's look-ahead reads, and while a group's records are sorted in
 * LDS every position of the group gets its ring search from them (cr_rolz3_group_lookups -> T.raw16). k_rolz_match then skips its
 * sweep of the 1 MB head table and its first pass for the block — on the harder corpus 7.6 of its 12.5 ms. Then the rows the same
 * way (cr_rolz3_group_rows -> T.row16): k_rolz_match keeps the parse's look-ahead. Returns the block's CrBatch::pre_done mark: 2 =
 * all of that, 3 = rings and plain lookups only (one row holds more positions than a group: k_rolz_match sweeps and searches the
 * rows), 0 = the ring keys do not split into groups either (one repeated byte). */
CR_DEV uint32_t cr_rolz_rings_block_lds64(const CrLz2Shared& S, CrLz3Groups& G, const uint8_t* g, uint32_t n, bool flexible, const CrRolzTables& T, u64* st = nullptr) {
    const uint32_t link_limit = n - CR_ROLZ_TAIL + (flexible ? CR_ROLZ_MAX + 1u : CR_ROLZ_MIN);
#ifdef CR_LZ3_PROF      /* tools/rolz_match_profile.py: slots 1-5, 7 = the rows (bins + cut | compaction | passes | neighbours | filter words | searches), 8-13 the rings */
    if (st && threadIdx.x == 0) { for (int q = 0; q < 15; q++) st[q] = 0; st[15] = wall_clock64(); }
#endif
    cr_lz2_stage_block(S, g, n);
    CR_LZ3_MARK(st, 0);
    CrRolzRingKey rk; rk.d = S.src; rk.ctx4 = false;      /* using_ctx4 needs 4 MiB blocks, cr-coder.c:158 */
    uint16_t* const r16 = T.ring16;
    uint32_t* const r32 = T.ring_prev;
    const uint8_t* const d = S.src;
    uint16_t* const raw16 = T.raw16;
    const auto lookups = [d, raw16, st](const uint16_t* cur, const u64* same, uint16_t* spare, uint32_t m) { cr_rolz3_group_lookups(d, cur, same, spare, m, raw16, st); };
    if (!cr_lz3_prev_same(S, G, rk, CR_ROLZ_WARM, link_limit - CR_ROLZ_WARM, 18u, [r16, r32](uint32_t p, uint32_t q) {
        if (r16) r16[p] = (uint16_t)(q == CR_LZ2_NONE ? 0xffffu : q); else r32[p] = q == CR_LZ2_NONE ? CR_ROLZ_NONE : q;
    }, st, &lookups)) return 0u;
    cr_wg_sync_global();
#ifdef CR_LZ3_PROF
    if (st && threadIdx.x == 0) { st[8] = st[1]; st[9] = st[2]; st[10] = st[3]; st[11] = st[4]; st[12] = st[7]; st[13] = st[5]; st[1] = st[2] = st[3] = st[4] = st[5] = st[7] = 0; }
    __syncthreads();
#endif                                   /* the row searches read raw16 */
    /* rows: sorted by the byte in front (one digit), searched where the plain lookup fell short; no links are written — nothing
     * reads them once the searches are done */
    CrRolzRowKey wk; wk.d = S.src;
    const uint32_t limit = n - CR_ROLZ_TAIL;
    uint16_t* const row16 = T.row16;
    const auto rows = [d, raw16, row16, limit, st](const uint16_t* cur, const u64* same, uint16_t* spare, uint32_t m) { cr_rolz3_group_rows(d, cur, same, spare, m, limit, raw16, row16, st); };
    return cr_lz3_prev_same(S, G, wk, CR_ROLZ_WARM, link_limit - CR_ROLZ_WARM, 8u, [](uint32_t, uint32_t) {}, st, &rows) ? 2u : 3u;
}

#endif

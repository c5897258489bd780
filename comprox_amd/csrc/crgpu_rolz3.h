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


/* the ring links of a block of CR_LZ2_MAXN < n <= CR_LZ3_MAXN bytes by the same sort in groups (crgpu_lzp2.h, round 4), written
 * to the link array k_rolz_match's searches read — that kernel then skips its sweep of the 1 MB head table for the block (the
 * row links, 256 heads in LDS, stay with it). Returns false when the keys do not split into groups. */
CR_DEV bool cr_rolz_rings_block_lds64(const CrLz2Shared& S, CrLz3Groups& G, const uint8_t* g, uint32_t n, bool flexible, const CrRolzTables& T) {
    const uint32_t link_limit = n - CR_ROLZ_TAIL + (flexible ? CR_ROLZ_MAX + 1u : CR_ROLZ_MIN);
    cr_lz2_stage_block(S, g, n);
    CrRolzRingKey rk; rk.d = S.src; rk.ctx4 = false;      /* using_ctx4 needs 4 MiB blocks, cr-coder.c:158 */
    uint16_t* const r16 = T.ring16;
    uint32_t* const r32 = T.ring_prev;
    return cr_lz3_prev_same(S, G, rk, CR_ROLZ_WARM, link_limit - CR_ROLZ_WARM, 18u, [r16, r32](uint32_t p, uint32_t q) {
        if (r16) r16[p] = (uint16_t)(q == CR_LZ2_NONE ? 0xffffu : q); else r32[p] = q == CR_LZ2_NONE ? CR_ROLZ_NONE : q;
    });
}

#endif

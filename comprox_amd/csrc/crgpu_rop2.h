/*
 * comprox_amd/csrc/crgpu_rop2.h — context-partitioned comprop ENCODER (batched, reset-per-block mode).
 *
 * The reference's lzencode (src/ropmain/cr-coder.c:119-229) pushes one symbol at a time through
 * ppm_encode (src/cr-ppm.c:103-167). For an encoder the whole event list — (context, symbol) of
 * every ppm_encode call — is known once the LZP parse is, and the three model components evolve
 * independently per key:
 *   order-3 predictor  per 22-bit key      (cr-ppm.c:66-88)   needs: symbols at that key, in order
 *   order-2 node       per 16-bit context  (cr-o2model.c)     needs: symbols + predicted byte
 *   order-1 row        per last byte       (cr-ppm.c:90-98)   needs: the escapes + exclusion sets
 * Only the range coder (cr-rangecoder.c:60-70) is a serial recurrence over all events. So instead of
 * one wave walking 45 000 steps of ~300 instructions each with a memory round trip in every step:
 *   k_rop_events  token loop only: emits the event list                       (1 wave / block)
 *   k_rop_links   stable radix sorts of the events by order-2 key and by order-3 key (histograms in
 *                 LDS): every chain becomes a contiguous run, in coding order     (4 waves / block)
 *   k_rop_o3      one LANE per order-3 chain: predicted byte of every event
 *   k_rop_o2      one LANE per order-2 chain: the node lives in that lane's slice of LDS; emits
 *                 (cum, frq, tot) per event and, for escapes, the exclusion set
 *   k_rop_rc      1 wave / block: range coder over the prepared triples, order-1 step for the
 *                 escapes (wave-parallel sums), output bytes, stored-block test, header
 * Bit-exactness: every component performs the same updates in the same per-key order as the
 * sequential coder; tests/test_gpu_rop.py checks both encoders against the oracle.
 */
#ifndef CRGPU_ROP2_H
#define CRGPU_ROP2_H

#include "crgpu_rop.h"

#define CR_EV_LAST   0x8000u      /* ev_sym: last event of its token (the ob >= ib test follows it) */
#define CR_T_HIT     0u
#define CR_T_BYTE    1u
#define CR_T_ESC     2u
/* bit 60 of an escape's triple (round 5): the node held no byte when it escaped — every first event of a chain, 38 % of the bench
 * corpus's escapes, all of the Markov stream's — so its exclusion set is empty: k_rop_o2 does not store one (two scattered 16-byte
 * stores less), the compaction copies the bit into the sorted record (bit 56) and k_rop_o1 does not gather one */
#define CR_TRIP_EMPTY 60u
/* the triple of an escape out of a node that holds nothing yet (counts 0 / 1 / 1), predicted with byte 0: cum 1, total 2, frequency 1 —
 * every event of the Markov stream and every sixth of the bench corpus's. k_rop_o3 fills the triples with it in passing (in order; the
 * region is the links kernels' scratch until then) and k_rop_o2 only stores the triples that differ. */
#define CR_TRIP_FRESH (1ull | (2ull << 20) | (1ull << 40) | ((u64)CR_T_ESC << 50) | (1ull << CR_TRIP_EMPTY))

/* per-block scratch (device memory owned by the context, one slot per block of the batch) */
struct CrEvViews {
    uint32_t* ctr;       /* [0] #events, [1] #order-2 chains, [2] #order-3 chains, [3] esc | stored<<8 */
    uint32_t* ev_ctx;    /* u32[cap]: the four bytes in front of the event */
    uint16_t* ev_sym;    /* u16[cap]: symbol | CR_EV_LAST */
    u64*      trip;      /* u64[cap]: cum | tot << 20 | frq << 40 | type << 50 | predicted byte << 52 */
    uint32_t* mask;      /* u32[cap][8]: bit s set = byte s has a count in the node (escape events only);
                          * the order-1 pass replaces words 0-1 by the escape's second triple lo | all << 20 | frq << 40 */
    /* order-2 chains, laid out one after the other (slot = position in that layout) */
    uint32_t* list2;     /* u32[cap]: event number at each slot */
    uint32_t* slot2;     /* u32[cap]: slot of each event */
    uint16_t* csym2;     /* u16[cap]: symbol at each slot | 0x8000 on the last slot of a chain */
    uint8_t*  cpred;     /* u8[cap]: predicted byte at each slot (written by the order-3 pass) */
    u64*      chains2;   /* u64[cap]: first slot | end slot << 32, long chains first */
    /* order-3 chains, same idea */
    uint32_t* cslot3;    /* u32[cap]: order-2 slot of the event at each order-3 slot */
    uint16_t* csym3;     /* u16[cap] */
    uint32_t* starts3;   /* u32[cap]: first slot of every chain */
    /* sort scratch, aliased onto `mask` and `trip` (both are written later, by the order-2 pass) */
    u64*      sortA;     /* u64[cap]: key << 32 | event */
    u64*      sortB;     /* u64[cap] */
    uint32_t* starts2;   /* u32[cap] */
    uint32_t* thist;     /* u32[tiles][256]: digit counts per tile */
    /* order-1 pass scratch, aliased onto chains2 and list2+slot2 (dead once the order-2 pass is done) */
    u64*      escA;      /* u64[cap]: row << 32 | event, escapes in coding order */
    u64*      escB;      /* u64[cap]: the same, row after row */
    uint32_t  cap;
};

CR_DEV u64 cr_ev_slot_bytes(uint32_t cap) { return 64ull + (u64)cap * (8u + 32u + 4u * 5u + 8u + 2u * 3u + 2u) + ((u64)cap / 1024u + 2u) * 1024u + 512u; }

CR_DEV CrEvViews cr_ev_views(uint8_t* base, uint32_t cap) {
    CrEvViews V;
    V.cap = cap;
    V.ctr = reinterpret_cast<uint32_t*>(base);
    uint8_t* p = base + 64;
    V.trip = reinterpret_cast<u64*>(p);            p += (u64)cap * 8u;
    V.mask = reinterpret_cast<uint32_t*>(p);       p += (u64)cap * 32u;
    V.chains2 = reinterpret_cast<u64*>(p);         p += (u64)cap * 8u;
    V.list2 = reinterpret_cast<uint32_t*>(p);      p += (u64)cap * 4u;
    V.slot2 = reinterpret_cast<uint32_t*>(p);      p += (u64)cap * 4u;
    V.ev_ctx = reinterpret_cast<uint32_t*>(p);     p += (u64)cap * 4u;
    V.cslot3 = reinterpret_cast<uint32_t*>(p);     p += (u64)cap * 4u;
    V.starts3 = reinterpret_cast<uint32_t*>(p);    p += (u64)cap * 4u;
    V.ev_sym = reinterpret_cast<uint16_t*>(p);     p += (u64)cap * 2u;
    V.csym2 = reinterpret_cast<uint16_t*>(p);      p += (u64)cap * 2u;
    V.csym3 = reinterpret_cast<uint16_t*>(p);      p += (u64)cap * 2u;
    V.cpred = p;                                   p += (u64)cap;
    p += (16u - (reinterpret_cast<uintptr_t>(p) & 15u)) & 15u;
    V.thist = reinterpret_cast<uint32_t*>(p);
    V.sortA = reinterpret_cast<u64*>(V.mask);
    V.sortB = reinterpret_cast<u64*>(V.mask) + (u64)cap;
    V.starts2 = reinterpret_cast<uint32_t*>(V.trip);
    V.escA = V.chains2;
    V.escB = reinterpret_cast<u64*>(V.list2);
    return V;
}

/* ------------------------------------------------------------------ k_rop_events */

/* token loop of lzencode (cr-coder.c:169-207) without the coding, 64 positions per step.
 * Which positions start a token only depends on the match lengths (a match of length L at a token
 * start covers the next L-1 positions), so a step resolves its few matches with scalar bit
 * operations and everything else is per lane: the context of a token is the four bytes in front of
 * it. Two exceptions are patched by a (rare) ordered walk over the step's tokens: the context starts
 * at 0 at position 9 (cr-coder.c:143-145) and a literal escape byte enters the context twice
 * (cr-coder.c:186-190); either wears off after a match or a few tokens. */
#define CR_EVT_BATCH 4u
CR_DEV void cr_rop_emit_events(const uint8_t* src, uint32_t n, const uint8_t* lens, CrEvViews& V, CrShared& sh) {
    const uint32_t lane = cr_lane();
    if (n < 16u) {                                            /* cr-coder.c:140-142 */
        if (lane == 0) { V.ctr[0] = 0; V.ctr[1] = 0; V.ctr[2] = 0; V.ctr[3] = 0x100u; }
        return;
    }
    const uint32_t esc = cr_pick_escape(src, n, sh.hist);
    uint32_t nev = 0, skip_until = CR_LZP_SKIP;
    uint32_t fixn = 4, fctx = 0;                              /* tokens still to patch, and with what */
    for (uint32_t base0 = CR_LZP_SKIP; base0 < n; base0 += 64u * CR_EVT_BATCH) {
        uint32_t bc[CR_EVT_BATCH], bl[CR_EVT_BATCH], bx[CR_EVT_BATCH];
#pragma unroll
        for (uint32_t u = 0; u < CR_EVT_BATCH; u++) {      /* unconditional loads (clamped index): behind `if (p < n)` every batch member's loads were waited for on the spot */
            const uint32_t p = base0 + u * 64u + lane, q = p < n ? p : n - 1u;
            bc[u] = src[q];
            bx[u] = *reinterpret_cast<const cr_u32u*>(src + q - 4u);
            bl[u] = lens[q];
        }
#pragma unroll
        for (uint32_t u = 0; u < CR_EVT_BATCH; u++) {
            const uint32_t p = base0 + u * 64u + lane;
            const bool in = p < n;
            bc[u] = in ? bc[u] : 0u;
            bx[u] = in ? __builtin_bswap32(bx[u]) : 0u;
            bl[u] = in && p + CR_LZP_TAIL < n ? bl[u] : 1u;
        }
#pragma unroll
        for (uint32_t u = 0; u < CR_EVT_BATCH; u++) {
            const uint32_t base = base0 + u * 64u;
            if (base >= n) break;
            const uint32_t c = bc[u], len = bl[u];
            uint32_t ctx = bx[u];
            const bool is_match = len > 1u;
            /* token starts of this step */
            const u64 mm = cr_ballot(is_match);
            u64 starts = 0;
            uint32_t cur = skip_until > base ? skip_until - base : 0u;
            while (cur < 64u) {
                const u64 rest = mm >> cur << cur;
                if (!rest) { starts |= ~0ull << cur; cur = 64u; break; }
                const uint32_t l = (uint32_t)__builtin_ctzll(rest);
                starts |= (~0ull << cur) & (l == 63u ? ~0ull : ((2ull << l) - 1ull));
                cur = l + cr_lane_get(len, l);
            }
            skip_until = base + cur;
            if (n - base < 64u) starts &= (1ull << (n - base)) - 1ull;
            const bool start = (starts >> lane) & 1ull;
            const bool esc_lit = start && !is_match && c == esc;
            if (fixn || cr_ballot(esc_lit)) {                                /* rare: ordered walk */
                for (u64 todo = starts; todo; todo &= todo - 1ull) {
                    const uint32_t l = (uint32_t)__builtin_ctzll(todo);
                    if (fixn) { if (lane == l) ctx = fctx; }
                    const uint32_t cx = fixn ? fctx : cr_lane_get(ctx, l);
                    const uint32_t cl = cr_lane_get(c, l);
                    if ((mm >> l) & 1ull) fixn = 0;
                    else if (cl == esc) { fctx = (cx << 16) | (esc << 8) | esc; fixn = 3; }
                    else if (fixn) { fctx = (cx << 8) | cl; fixn--; }
                }
            }
            const bool two = start && (is_match || c == esc);
            const uint32_t cnt = start ? (two ? 2u : 1u) : 0u;
            const uint32_t incl = cr_scan_incl(cnt);
            const uint32_t e = nev + incl - cnt;
            if (start) {
                V.ev_ctx[e] = ctx;
                if (two) {
                    V.ev_sym[e] = (uint16_t)esc;
                    V.ev_ctx[e + 1u] = (ctx << 8) | esc;
                    V.ev_sym[e + 1u] = (uint16_t)((is_match ? len : 0u) | CR_EV_LAST);
                } else {
                    V.ev_sym[e] = (uint16_t)(c | CR_EV_LAST);
                }
            }
            nev += cr_lane_get(incl, 63);
        }
    }
    if (lane == 0) { V.ctr[0] = nev; V.ctr[1] = 0; V.ctr[2] = 0; V.ctr[3] = esc; }
}

/* ------------------------------------------------------------------ k_rop_links */

#define CR_SORT_THREADS 256u
#define CR_SORT_WAVES   4u
#ifndef CR_TILE
#define CR_TILE         4096u                   /* elements per tile: 16 steps of 64 for each of the 4 waves */
#endif
#define CR_TILE_STEPS   (CR_TILE / 64u / CR_SORT_WAVES)
struct CrSortShared {
    u64      buf[CR_TILE];                      /* the tile in digit order: key << 32 | event */
    uint32_t whist[CR_SORT_WAVES * 256u];       /* [wave][digit]: counts, then next free place in buf */
    uint32_t goff[256];                         /* where the tile's run of digit d goes in the output */
    uint32_t tstart[256];                       /* where it starts in buf */
    uint32_t wsum[CR_SORT_WAVES];
    uint32_t n2, n3, front, back;
};

CR_DEV void cr_wg_sync_global() {                       /* other waves' global stores become readable */
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
    __syncthreads();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
}
/* LDS accesses of one wave execute in program order; this only stops the compiler from moving them */
CR_DEV void cr_lds_order() { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); __builtin_amdgcn_wave_barrier(); }
/* the same without the wait: the LDS unit takes a wave's DS instructions in issue order, so a later read of this wave sees
 * its earlier write; only the compiler must not move them across this point */
CR_DEV void cr_lds_order_sw() { asm volatile("" ::: "memory"); __builtin_amdgcn_wave_barrier(); }

/* exclusive prefix sum over the 256 threads (two barriers inside) */
CR_DEV uint32_t cr_wg_scan_excl(CrSortShared& sh, uint32_t v) {
    const uint32_t incl = cr_scan_incl(v);
    __syncthreads();
    if (cr_lane() == 63u) sh.wsum[cr_wave_id()] = incl;
    __syncthreads();
    uint32_t run = incl - v;
    for (uint32_t ww = 0; ww < cr_wave_id(); ww++) run += sh.wsum[ww];
    return run;
}

/* one stable counting-sort pass on 8 bits of the key, whole workgroup (4 waves), `in` == nullptr:
 * the elements are made from the event contexts (key = order-2 context or order-3 key).
 * The input is cut into tiles; a tile is first put in digit order in LDS and then copied out run by
 * run, so the global stores are contiguous pieces instead of 64 different lines per instruction.
 * Inside a tile wave w owns a contiguous quarter; the 64 elements of a step are ranked among the
 * lanes with the same digit by bit-sliced ballots. Equal keys therefore keep their input order. */
CR_DEV void cr_wg_stamp(u64* st, int slot) { if (st && threadIdx.x == 0) st[slot] = wall_clock64(); }

CR_DEV uint32_t cr_sort_pass(CrSortShared& sh, uint32_t nev, uint32_t shift, int keyfn,
                         const u64* in, const uint32_t* ev_ctx, u64* out, uint32_t* thist, u64* st) {
    const uint32_t t = threadIdx.x, w = cr_wave_id(), lane = cr_lane();
    const uint32_t ntiles = (nev + CR_TILE - 1u) / CR_TILE;
    u64 el[CR_TILE_STEPS];
#define CR_TILE_LOAD(tile_) do { \
        _Pragma("unroll") for (uint32_t u = 0; u < CR_TILE_STEPS; u++) { \
            const uint32_t s_ = (tile_) * CR_TILE + (w * CR_TILE_STEPS + u) * 64u + lane; \
            el[u] = ~0ull; \
            if (s_ < nev) { \
                if (in) el[u] = in[s_]; \
                else { const uint32_t c_ = ev_ctx[s_]; el[u] = ((u64)(keyfn == 1 ? (c_ & 0xffffu) : cr_o3_key(c_)) << 32) | s_; } \
            } \
        } } while (0)
#define CR_EL_DIGIT(e_) ((uint32_t)((e_) >> (32u + shift)) & 0xffu)
    /* 1: digit counts of every tile */
    for (uint32_t tile = 0; tile < ntiles; tile++) {
        sh.goff[t] = 0;
        CR_TILE_LOAD(tile);
        __syncthreads();
#pragma unroll
        for (uint32_t u = 0; u < CR_TILE_STEPS; u++) if (el[u] != ~0ull) atomicAdd(&sh.goff[CR_EL_DIGIT(el[u])], 1u);
        __syncthreads();
        thist[tile * 256u + t] = sh.goff[t];
    }
    cr_wg_stamp(st, 8);
    /* 2: thread d: running sum of digit d over the tiles, then the digits' bases */
    uint32_t run = 0;
    for (uint32_t tile = 0; tile < ntiles; tile++) { const uint32_t v = thist[tile * 256u + t]; thist[tile * 256u + t] = run; run += v; }
    const uint32_t dbase = cr_wg_scan_excl(sh, run);
    cr_wg_stamp(st, 9);
    /* 3: tile by tile */
    for (uint32_t tile = 0; tile < ntiles; tile++) {
        const uint32_t tile_n = nev - tile * CR_TILE < CR_TILE ? nev - tile * CR_TILE : CR_TILE;
        sh.goff[t] = dbase + thist[tile * 256u + t];
        for (uint32_t ww = 0; ww < CR_SORT_WAVES; ww++) sh.whist[ww * 256u + t] = 0;
        CR_TILE_LOAD(tile);
        __syncthreads();
#pragma unroll
        for (uint32_t u = 0; u < CR_TILE_STEPS; u++) if (el[u] != ~0ull) atomicAdd(&sh.whist[w * 256u + CR_EL_DIGIT(el[u])], 1u);
        __syncthreads();
        {
            uint32_t c[CR_SORT_WAVES], cnt = 0;
            for (uint32_t ww = 0; ww < CR_SORT_WAVES; ww++) { c[ww] = sh.whist[ww * 256u + t]; cnt += c[ww]; }
            uint32_t at = cr_wg_scan_excl(sh, cnt);
            sh.tstart[t] = at;
            for (uint32_t ww = 0; ww < CR_SORT_WAVES; ww++) { sh.whist[ww * 256u + t] = at; at += c[ww]; }
        }
        __syncthreads();
#pragma unroll
        for (uint32_t u = 0; u < CR_TILE_STEPS; u++) {
            const bool act = el[u] != ~0ull;
            const uint32_t digit = CR_EL_DIGIT(el[u]);
            const u64 same = cr_same_key_mask<8>(digit, act);
            const u64 lower = same & ((1ull << lane) - 1ull);
            uint32_t base = 0;
            if (act) {
                base = sh.whist[w * 256u + digit];
                sh.buf[base + (uint32_t)__builtin_popcountll(lower)] = el[u];
            }
            cr_lds_order();
            if (act && (same >> lane) >> 1 == 0ull) sh.whist[w * 256u + digit] = base + (uint32_t)__builtin_popcountll(same);
            cr_lds_order();
        }
        __syncthreads();
        for (uint32_t q = t; q < tile_n; q += CR_SORT_THREADS) {
            const u64 e = sh.buf[q];
            const uint32_t d = CR_EL_DIGIT(e);
            out[sh.goff[d] + (q - sh.tstart[d])] = e;
        }
        __syncthreads();
    }
#undef CR_TILE_LOAD
#undef CR_EL_DIGIT
    cr_wg_stamp(st, 10);
    cr_wg_sync_global();
    cr_wg_stamp(st, 11);
    return dbase;                                            /* thread d: where digit d starts in the output */
}

/* whole workgroup: both sorts, then the per-slot views the chain passes read sequentially */
CR_DEV void cr_rop_sort_events(CrSortShared& sh, CrEvViews& V, uint32_t* last2 /* u32[65536], global */, uint32_t nev, u64* st) {
    const uint32_t t = threadIdx.x;
    cr_wg_stamp(st, 0);
    if (t == 0) { sh.n2 = 0; sh.n3 = 0; sh.front = 0; sh.back = 0; }
    /* passes 0-1: order-2 context, 16 bits; passes 2-4: order-3 key, 22 bits */
    for (uint32_t p = 0; p < 5u; p++) {
        const u64* in = (p == 0u || p == 2u) ? nullptr : (p == 3u ? V.sortA : (p == 1u ? V.sortA : V.sortB));
        u64* out = (p == 1u || p == 3u) ? V.sortB : V.sortA;
        cr_sort_pass(sh, nev, p < 2u ? p * 8u : (p - 2u) * 8u, p < 2u ? 1 : 2, in, V.ev_ctx, out, V.thist, p == 0u ? st : nullptr);
        cr_wg_stamp(st, p < 2u ? 1 + (int)p : 2 + (int)p);
        if (p == 1u) {
            const u64* S = V.sortB;
            for (uint32_t s0 = t; s0 < nev; s0 += CR_SORT_THREADS * 4u) {
                u64 be[4]; uint32_t bn[4], bp[4], bs[4];
#pragma unroll
                for (uint32_t u = 0; u < 4u; u++) {
                    const uint32_t s = s0 + u * CR_SORT_THREADS;
                    be[u] = 0; bn[u] = 0xFFFFFFFFu; bp[u] = 0xFFFFFFFFu;
                    if (s < nev) { be[u] = S[s]; if (s + 1u < nev) bn[u] = (uint32_t)(S[s + 1u] >> 32); if (s) bp[u] = (uint32_t)(S[s - 1u] >> 32); }
                }
#pragma unroll
                for (uint32_t u = 0; u < 4u; u++) bs[u] = V.ev_sym[(uint32_t)be[u]];
#pragma unroll
                for (uint32_t u = 0; u < 4u; u++) {
                    const uint32_t s = s0 + u * CR_SORT_THREADS;
                    if (s >= nev) break;
                    const uint32_t i = (uint32_t)be[u], k = (uint32_t)(be[u] >> 32);
                    const bool last = bn[u] != k, first = bp[u] != k;
                    V.list2[s] = i;
                    V.csym2[s] = (uint16_t)((bs[u] & 0x1ffu) | (last ? 0x8000u : 0u));
                    V.cpred[s] = 0;                       /* (k_rop_o3's range walker only stores the predictions that are not 0) */
                    V.slot2[i] = s;
                    if (last) last2[k] = s + 1u;
                    if (first) V.starts2[atomicAdd(&sh.n2, 1u)] = s;
                }
            }
            cr_wg_sync_global();
            /* chains of 96 events and more first: a lane that meets one late would finish long after the others */
            const uint32_t n2 = sh.n2;
            for (uint32_t c = t; c < n2; c += CR_SORT_THREADS) {
                const uint32_t s0 = V.starts2[c], e = last2[(uint32_t)(S[s0] >> 32)];
                const uint32_t at = (e - s0 >= 96u) ? atomicAdd(&sh.front, 1u) : n2 - 1u - atomicAdd(&sh.back, 1u);
                V.chains2[at] = (u64)s0 | ((u64)e << 32);
            }
            __syncthreads();                                   /* sortB is overwritten by pass 3 */
            cr_wg_stamp(st, 3);
        }
    }
    {
        const u64* S = V.sortA;
        for (uint32_t s0 = t; s0 < nev; s0 += CR_SORT_THREADS * 4u) {
            u64 be[4]; uint32_t bn[4], bp[4], bs[4], bl[4];
#pragma unroll
            for (uint32_t u = 0; u < 4u; u++) {
                const uint32_t s = s0 + u * CR_SORT_THREADS;
                be[u] = 0; bn[u] = 0xFFFFFFFFu; bp[u] = 0xFFFFFFFFu;
                if (s < nev) { be[u] = S[s]; if (s + 1u < nev) bn[u] = (uint32_t)(S[s + 1u] >> 32); if (s) bp[u] = (uint32_t)(S[s - 1u] >> 32); }
            }
#pragma unroll
            for (uint32_t u = 0; u < 4u; u++) { bs[u] = V.ev_sym[(uint32_t)be[u]]; bl[u] = V.slot2[(uint32_t)be[u]]; }
#pragma unroll
            for (uint32_t u = 0; u < 4u; u++) {
                const uint32_t s = s0 + u * CR_SORT_THREADS;
                if (s >= nev) break;
                const uint32_t k = (uint32_t)(be[u] >> 32);
                const bool last = bn[u] != k, first = bp[u] != k;
                V.csym3[s] = (uint16_t)((bs[u] & 0x1ffu) | (last ? 0x8000u : 0u));
                V.cslot3[s] = bl[u];
                if (first) V.starts3[atomicAdd(&sh.n3, 1u)] = s;
            }
        }
    }
    __syncthreads();
    cr_wg_stamp(st, 7);
    if (t == 0) { V.ctr[1] = sh.n2; V.ctr[2] = sh.n3; }
}

/* ------------------------------------------------------------------ k_rop_o3 */

/* one lane walks one order-3 chain: ppm_update_o3 (cr-ppm.c:69-88) with the table entry in registers */
CR_DEV void cr_rop_o3_chain(CrEvViews& V, uint32_t s) {
    uint32_t pred = 0, conf = 0;
    uint32_t n_sym = V.csym3[s], n_slot = V.cslot3[s];
    for (;;) {
        const uint32_t sy = n_sym, slot = n_slot;
        const bool last = (sy & 0x8000u) != 0u;
        s++;
        if (!last) { n_sym = V.csym3[s]; n_slot = V.cslot3[s]; }
        const uint32_t sym = sy & 0x1ffu;
        V.cpred[slot] = (uint8_t)pred;
        if (sym == pred) {
            conf += conf < 15u ? 1u : 0u;
        } else {
            uint32_t c = (uint32_t)(conf > 1u) + (uint32_t)(conf > 2u) + (uint32_t)(conf > 4u) + (uint32_t)(conf > 8u);
            if (c == 0u) { pred = sym; c = 1u; }
            conf = c;
        }
        if (last) break;
    }
}

/* the same chains walked as RANGES (round 4, blocks of up to CR_O2R_MAXEV events; the why and how is at cr_rop_o2_ranges below: one
 * thread per chain is a memory round trip per step — the next slot's operands are asked for when the step starts and the step is
 * ten instructions — and two at every chain start; parked 84 % of the kernel's time): every lane walks contiguous slots, fetched four at
 * a time one round ahead, predictions stored a round later. CrO2Ranges is declared with the order-2 pass. */
struct CrO2Ranges;
CR_DEV void cr_rop_o3_ranges(CrEvViews& V, CrO2Ranges& R, uint32_t nev);

/* ------------------------------------------------------------------ k_rop_o2 */

/* node of one order-2 chain in the lane's slice of LDS: 256 byte counts, the eight 32-symbol group
 * sums and the 256-bit "has a count" set; total of the byte counts and the two flag counts in registers */
#define CR_LN_G      256u
#define CR_LN_NZ     288u
#define CR_LN_STRIDE 336u      /* 84 words: 16-byte accesses of the 64 lanes spread evenly over the banks */
struct CrLaneNode {
    uint8_t* cnt;
    uint32_t bytes, fh, fe;
};

CR_DEV uint32_t cr_ln_sum4(uint32_t w) { return __builtin_amdgcn_sad_u8(w, 0u, 0u); }
CR_DEV uint32_t cr_nz4(uint32_t x) {
    return ((x & 0x000000ffu) ? 1u : 0u) | ((x & 0x0000ff00u) ? 2u : 0u) | ((x & 0x00ff0000u) ? 4u : 0u) | ((x & 0xff000000u) ? 8u : 0u);
}

CR_DEV void cr_ln_clear(CrLaneNode& nd) {                                   /* o2_model_init */
    for (uint32_t q = 0; q < 20u; q++) reinterpret_cast<uint4*>(nd.cnt)[q] = make_uint4(0u, 0u, 0u, 0u);
    nd.bytes = 0; nd.fh = 1; nd.fe = 1;
}

/* o2_model_update's halving pass (cr-o2model.c:54-71), lane-serial over the 256 counts */
CR_DEV void cr_ln_halve(CrLaneNode& nd) {
    uint32_t singles = 1, bytes = 0;
    for (uint32_t gi = 0; gi < 8u; gi++) {
        uint32_t gs = 0, bits = 0;
        uint4* p = reinterpret_cast<uint4*>(nd.cnt + gi * 32u);
        for (uint32_t h = 0; h < 2u; h++) {
            uint4 v = p[h];
            v.x = (v.x >> 1) & 0x7f7f7f7fu; v.y = (v.y >> 1) & 0x7f7f7f7fu; v.z = (v.z >> 1) & 0x7f7f7f7fu; v.w = (v.w >> 1) & 0x7f7f7f7fu;
            p[h] = v;
            gs += cr_ln_sum4(v.x) + cr_ln_sum4(v.y) + cr_ln_sum4(v.z) + cr_ln_sum4(v.w);
            singles += cr_count_ones_bytes(v.x) + cr_count_ones_bytes(v.y) + cr_count_ones_bytes(v.z) + cr_count_ones_bytes(v.w);
            bits |= (cr_nz4(v.x) | (cr_nz4(v.y) << 4) | (cr_nz4(v.z) << 8) | (cr_nz4(v.w) << 12)) << (h * 16u);
        }
        reinterpret_cast<uint32_t*>(nd.cnt + CR_LN_G)[gi] = gs;
        reinterpret_cast<uint32_t*>(nd.cnt + CR_LN_NZ)[gi] = bits;
        bytes += gs;
    }
    nd.bytes = bytes;
    nd.fh = (nd.fh + 1u) >> 1;
    nd.fe = singles & 0xffu;
}

/* sum of the counts of the symbols below `sym` (o2_model_cum, cr-o2model.c:75-84), branch-free */
CR_DEV uint32_t cr_ln_below(const CrLaneNode& nd, uint32_t sym) {
    const uint32_t gi = sym >> 5, within = sym & 31u;
    const uint4 g0 = reinterpret_cast<const uint4*>(nd.cnt + CR_LN_G)[0], g1 = reinterpret_cast<const uint4*>(nd.cnt + CR_LN_G)[1];
    uint32_t acc = (gi > 0u ? g0.x : 0u) + (gi > 1u ? g0.y : 0u) + (gi > 2u ? g0.z : 0u) + (gi > 3u ? g0.w : 0u)
                 + (gi > 4u ? g1.x : 0u) + (gi > 5u ? g1.y : 0u) + (gi > 6u ? g1.z : 0u);
    const uint4 a = reinterpret_cast<const uint4*>(nd.cnt + gi * 32u)[0], b = reinterpret_cast<const uint4*>(nd.cnt + gi * 32u)[1];
    const uint32_t ws[8] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w};
#pragma unroll
    for (uint32_t k = 0; k < 8u; k++) {
        const int take = (int)within - (int)(k * 4u);
        const uint32_t m = take >= 4 ? 0xFFFFFFFFu : take <= 0 ? 0u : ((1u << (8 * take)) - 1u);
        acc += cr_ln_sum4(ws[k] & m);
    }
    return acc;
}

CR_DEV void cr_ln_count_up(CrLaneNode& nd, uint32_t sym, uint32_t to) {
    nd.cnt[sym] = (uint8_t)to;
    atomicAdd(reinterpret_cast<uint32_t*>(nd.cnt + CR_LN_G) + (sym >> 5), 1u);
    nd.bytes += 1u;
}

/* one coding step of an order-2 chain: the order-2 part of ppm_encode (cr-ppm.c:108-146,159-162) */
CR_DEV void cr_rop_o2_event(CrEvViews& V, CrLaneNode& nd, uint32_t i, uint32_t sym, uint32_t pred) {
    const uint32_t pf = nd.cnt[pred];
    const uint32_t bytes = nd.bytes;
    const uint32_t tot = bytes + nd.fh + nd.fe - pf;
    uint32_t cum, frq, type, empty = 0;
    if (sym == pred) {                                               /* cr-ppm.c:119-126 */
        cum = bytes - pf; frq = nd.fh; type = CR_T_HIT;
        nd.fh = (nd.fh + 1u) & 0xffu;
        if (nd.fh > 250u) cr_ln_halve(nd);
    } else {
        const uint32_t fs = nd.cnt[sym];
        if (fs) {                                                    /* cr-ppm.c:129-139 */
            cum = cr_ln_below(nd, sym) - (sym > pred ? pf : 0u); frq = fs; type = CR_T_BYTE;
            cr_ln_count_up(nd, sym, fs + 1u);
            if (fs + 1u > 250u) cr_ln_halve(nd);
            else if (fs + 1u == 2u) { nd.fe = (nd.fe - 1u) & 0xffu; if (nd.fe > 250u) cr_ln_halve(nd); }
        } else {                                                     /* cr-ppm.c:141-163 */
            cum = bytes + nd.fh - pf; frq = nd.fe; type = CR_T_ESC;
            nd.fe = (nd.fe + 1u) & 0xffu;
            bool halved = false;
            if (nd.fe > 250u) { cr_ln_halve(nd); halved = true; }
            if (nd.bytes) {                                                          /* what the node knows NOW; a node without a byte says so in the triple (CR_TRIP_EMPTY) */
                uint4* mo = reinterpret_cast<uint4*>(V.mask + (u64)i * 8u);
                mo[0] = reinterpret_cast<const uint4*>(nd.cnt + CR_LN_NZ)[0];
                mo[1] = reinterpret_cast<const uint4*>(nd.cnt + CR_LN_NZ)[1];
            } else empty = 1;
            if (!halved) {
                cr_ln_count_up(nd, sym, 1u);
                atomicOr(reinterpret_cast<uint32_t*>(nd.cnt + CR_LN_NZ) + (sym >> 5), 1u << (sym & 31u));
            }
        }
    }
    const u64 tr = (u64)cum | ((u64)tot << 20) | ((u64)frq << 40) | ((u64)type << 50) | ((u64)pred << 52) | ((u64)empty << CR_TRIP_EMPTY);
    if (tr != CR_TRIP_FRESH) V.trip[i] = tr;
}

/* the same step with its results left in registers: cr_rop_o2_ranges stores them later (why: see there) */
struct CrO2Out {
    u64 trip;
    uint4 m0, m1;          /* an escape's exclusion set: what the node knew when it escaped */
    uint32_t i;            /* event number */
    uint32_t kind;         /* 0 nothing to store, 1 the triple, 2 the triple and the exclusion set */
};
CR_DEV void cr_rop_o2_event_out(CrO2Out& out, CrLaneNode& nd, uint32_t i, uint32_t sym, uint32_t pred) {
    const uint32_t pf = nd.cnt[pred];
    const uint32_t bytes = nd.bytes;
    const uint32_t tot = bytes + nd.fh + nd.fe - pf;
    uint32_t cum, frq, type, empty = 0;
    out.kind = 1u;
    if (sym == pred) {                                               /* cr-ppm.c:119-126 */
        cum = bytes - pf; frq = nd.fh; type = CR_T_HIT;
        nd.fh = (nd.fh + 1u) & 0xffu;
        if (nd.fh > 250u) cr_ln_halve(nd);
    } else {
        const uint32_t fs = nd.cnt[sym];
        if (fs) {                                                    /* cr-ppm.c:129-139 */
            cum = cr_ln_below(nd, sym) - (sym > pred ? pf : 0u); frq = fs; type = CR_T_BYTE;
            cr_ln_count_up(nd, sym, fs + 1u);
            if (fs + 1u > 250u) cr_ln_halve(nd);
            else if (fs + 1u == 2u) { nd.fe = (nd.fe - 1u) & 0xffu; if (nd.fe > 250u) cr_ln_halve(nd); }
        } else {                                                     /* cr-ppm.c:141-163 */
            cum = bytes + nd.fh - pf; frq = nd.fe; type = CR_T_ESC;
            nd.fe = (nd.fe + 1u) & 0xffu;
            bool halved = false;
            if (nd.fe > 250u) { cr_ln_halve(nd); halved = true; }
            if (nd.bytes) {                                                              /* what the node knows NOW ... */
                out.m0 = reinterpret_cast<const uint4*>(nd.cnt + CR_LN_NZ)[0];
                out.m1 = reinterpret_cast<const uint4*>(nd.cnt + CR_LN_NZ)[1];
                out.kind = 2u;
            } else empty = 1;                                                            /* ... nothing: the triple says so, no set is stored */
            if (!halved) {
                cr_ln_count_up(nd, sym, 1u);
                atomicOr(reinterpret_cast<uint32_t*>(nd.cnt + CR_LN_NZ) + (sym >> 5), 1u << (sym & 31u));
            }
        }
    }
    out.i = i;
    out.trip = (u64)cum | ((u64)tot << 20) | ((u64)frq << 40) | ((u64)type << 50) | ((u64)pred << 52) | ((u64)empty << CR_TRIP_EMPTY);
}
CR_DEV void cr_rop_o2_store(CrEvViews& V, CrO2Out& out) {
    if (out.kind) {
        if (out.kind == 2u) {
            uint4* mo = reinterpret_cast<uint4*>(V.mask + (u64)out.i * 8u);
            mo[0] = out.m0; mo[1] = out.m1;
        }
        if (out.trip != CR_TRIP_FRESH) V.trip[out.i] = out.trip;
        out.kind = 0u;
    }
}

/* ---- round 4: the order-2 pass as RANGE walkers, for blocks of up to CR_O2R_MAXEV events (what a block of 65 537 bytes can hold).
 * The chains lie one after the other in slot order (csym2's bit 15 marks a chain's last slot), so a lane can walk a contiguous
 * range of slots and meet the chains in it one after the other: the operands of the next slots are at the next addresses.
 * What the ticket walkers above wait for is memory, three times over (75 % of the kernel's time was spent parked):
 *  - a chain is 3.6 events long on average, so in every step some lane of the wave starts one, and starting one is two dependent
 *    loads (the chain's descriptor, then the operands of its first slot) the other 63 lanes wait for as well;
 *  - this target counts loads and stores in ONE counter and in issue order (vmcnt). Where a step needs the operands it fetched
 *    ahead, the compiler has to assume that none of the step's CONDITIONAL stores (results behind `if`) was issued, so it waits
 *    until everything younger than the load has drained too: with the results stored at the end of a step that is the
 *    acknowledgement of stores issued a moment ago — a memory round trip per step whatever is prefetched. (And arithmetic on a
 *    freshly loaded value, such as packing symbol and prediction into one word, is a wait on the spot.)
 *  - every lane reads its own stream: a load or store instruction is 64 transactions with the L2, and 1 526 resident waves with
 *    six of them per step saturate it (3.7 us per step against 0.77 with 64 waves on the chip, whatever is prefetched).
 * Here: ranges of whole chains that start in the same 64 slots (`start`, a table in LDS built by one sweep over the flags),
 * handed out by a counter in LDS — ranges that run on into the next 64 slots first (`order`): they hold the long chains, and the
 * kernel lasts as long as its longest lane. A lane fetches its operands four slots at a time (aligned groups: 16 + 8 + 4 bytes
 * in three loads instead of twelve), one group ahead. A round of the loop (1) stores the PREVIOUS round's results — behind
 * `if`, only what an event really has: the stores are OLDER than the loads that follow, so nothing has to count them —,
 * (2) loads the next group, (3) computes the current group's up to four events in LDS, results into registers. Whatever the
 * round after waits for was issued a whole round (four steps) earlier. A range that starts or ends inside a group leaves the
 * lane idle for the group's other positions (~3 of ~70 steps). */
#define CR_O2R_CH     64u
#define CR_O2R_MAXEV  66816u                            /* = CrBatch::ev_cap of a 65 537-byte block, rounded up */
#define CR_O2R_CHUNKS (CR_O2R_MAXEV / CR_O2R_CH)
struct CrO2Ranges {
    uint16_t start[CR_O2R_CHUNKS + 8u];   /* start[c] = first chain start at or behind slot 64 c (the number of slots if there is none), low 16 bits:
                                           * the values only grow with c, so ONE number says from which chunk on bit 16 is set (`high`). A block of 65 537
                                           * bytes holds up to CrBatch::ev_cap = 66 689 events */
    uint16_t order[CR_O2R_CHUNKS + 8u];   /* the non-empty ranges, the ones that run on first */
    uint32_t norder, next, high;
};
CR_DEV uint32_t cr_o2r_start(const CrO2Ranges& R, uint32_t c) { return (uint32_t)R.start[c] + (c >= R.high ? 65536u : 0u); }
/* every thread of the workgroup (whole waves); nev <= CR_O2R_MAXEV; csym = the chains' symbols in slot order, bit 15 on a chain's last
 * slot */
CR_DEV void cr_rop_o2_ranges_build(const uint16_t* csym, CrO2Ranges& R, uint32_t nev) {
    const uint32_t lane = cr_lane(), w = cr_wave_id(), nw = blockDim.x >> 6;
    const uint32_t nch = (nev + CR_O2R_CH - 1u) / CR_O2R_CH;
    /* chunk c's first chain start, relative to the chunk (64 = none): slot s starts a chain iff s == 0 or slot s - 1 is a last slot
     * (every lane looks at the slot in front of its own: no carry between chunks, any wave takes any chunk; eight chunks' flags are
     * fetched per round) */
    for (uint32_t c0 = w * 8u; c0 < nch; c0 += nw * 8u) {
        uint32_t fl[8];
#pragma unroll
        for (uint32_t u = 0; u < 8u; u++) {
            const uint32_t s = (c0 + u) * CR_O2R_CH + lane;
            fl[u] = csym[s == 0u ? 0u : (s - 1u < nev ? s - 1u : nev - 1u)];
        }
#pragma unroll
        for (uint32_t u = 0; u < 8u; u++) {
            const uint32_t c = c0 + u, s = c * CR_O2R_CH + lane;
            const u64 firsts = cr_ballot(s < nev && (s == 0u || (fl[u] >> 15) != 0u));
            if (lane == 0 && c < nch) R.start[c] = (uint16_t)(firsts ? (uint32_t)__builtin_ctzll(firsts) : 64u);
        }
    }
    if (threadIdx.x == 0) { R.norder = 0; R.next = 0; }
    __syncthreads();
    if (w == 0u) {
        /* chunks without a chain start take the next one's (from the end); absolute slots from here on, 17 bits in registers */
        uint32_t follow = nev, below = 0;                                          /* below: chunks whose start is < 65 536 */
        for (uint32_t c0 = (nch + 63u) & ~63u; c0 > 0u; c0 -= 64u) {
            const uint32_t c = c0 - 64u + lane;
            const uint32_t rel = c < nch ? (uint32_t)R.start[c] : 64u;
            uint32_t v = rel < 64u ? c * CR_O2R_CH + rel : 0xffffffffu;
#pragma unroll
            for (uint32_t d = 1; d < 64u; d <<= 1) {                               /* suffix minimum over the 64 lanes */
                const uint32_t o = (uint32_t)__shfl_down((int)v, d);
                if (lane + d < 64u && o < v) v = o;
            }
            if (v > follow) v = follow;
            cr_wave_sync();
            if (c < nch) R.start[c] = (uint16_t)v;
            below += (uint32_t)__builtin_popcountll(cr_ballot(c < nch && v < 65536u));
            follow = cr_lane_get(v, 0);
            cr_wave_sync();
        }
        if (lane == 0) { R.start[nch] = (uint16_t)nev; R.high = nev < 65536u ? nch + 1u : below; }
        cr_wave_sync();
        /* hand-out order */
        for (uint32_t pass = 0; pass < 2u; pass++) {
            for (uint32_t c0 = 0; c0 < nch; c0 += 64u) {
                const uint32_t c = c0 + lane;
                bool take = false;
                if (c < nch) {
                    const uint32_t a = cr_o2r_start(R, c), b = cr_o2r_start(R, c + 1u), b2 = c + 2u <= nch ? cr_o2r_start(R, c + 2u) : nev;
                    const bool runs_on = b == b2 && c + 1u < nch;                    /* the next chunk starts no chain: one of this chunk's is still running */
                    take = a < b && (pass == 0u ? runs_on : !runs_on);
                }
                const u64 tm = cr_ballot(take);
                const uint32_t base = cr_uni(R.norder);
                if (take) R.order[base + (uint32_t)__builtin_popcountll(tm & ((1ull << lane) - 1ull))] = (uint16_t)c;
                cr_wave_sync();
                if (lane == 0) R.norder = base + (uint32_t)__builtin_popcountll(tm);
                cr_wave_sync();
            }
        }
    }
    __syncthreads();
}
CR_DEV void cr_rop_o2_ranges(CrEvViews& V, uint8_t* lane_node, CrO2Ranges& R, uint32_t nev, u64* st = nullptr) {
    if (st && cr_lane() == 0) st[0] = wall_clock64();
    cr_rop_o2_ranges_build(V.csym2, R, nev);
    if (st && cr_lane() == 0) { st[1] = wall_clock64(); st[3] = R.norder; }
    uint32_t rounds = 0;
    CrLaneNode nd;
    nd.cnt = lane_node; nd.bytes = 0; nd.fh = 1; nd.fe = 1;
    const uint32_t norder = cr_uni(R.norder);
    uint32_t fat = 0, fend = 0;                           /* the range the lane is fetching from */
    bool more = true;                                     /* ranges may be left */
    uint4 n_i = make_uint4(0u, 0u, 0u, 0u); uint2 n_sym = make_uint2(0u, 0u); uint32_t n_pred = 0;   /* the group fetched ahead: untouched until its round */
    uint32_t n_lo = 0, n_hi = 0; bool n_first = false;    /* its positions [n_lo, n_hi) are the lane's; n_first: position n_lo starts a range */
    const auto fetch = [&]() __attribute__((always_inline)) {
        bool first = false;
        if (fat >= fend && more) {
            const uint32_t k = atomicAdd(&R.next, 1u);
            more = k < norder;
            if (more) { const uint32_t c = R.order[k]; fat = cr_o2r_start(R, c); fend = cr_o2r_start(R, c + 1u); first = true; }
        }
        const bool has = fat < fend;
        const uint32_t g = has ? fat >> 2 : 0u;
        const uint32_t left = fend - 4u * g;
        n_lo = has ? fat & 3u : 0u; n_hi = has ? (left < 4u ? left : 4u) : 0u; n_first = first;
        n_i = reinterpret_cast<const uint4*>(V.list2)[g];
        n_sym = reinterpret_cast<const uint2*>(V.csym2)[g];
        n_pred = reinterpret_cast<const uint32_t*>(V.cpred)[g];
        if (has) fat = 4u * g + n_hi;
    };
    fetch();
    CrO2Out r0, r1, r2, r3;
    r0.kind = r1.kind = r2.kind = r3.kind = 0u;
    r0.trip = r1.trip = r2.trip = r3.trip = 0; r0.i = r1.i = r2.i = r3.i = 0;
    r0.m0 = r0.m1 = r1.m0 = r1.m1 = r2.m0 = r2.m1 = r3.m0 = r3.m1 = make_uint4(0u, 0u, 0u, 0u);
    bool fresh = true;
    for (;;) {
        if ((rounds & 15u) == 0u) cr_take_turns<0>(rounds >> 6);          /* (two of these waves on a SIMD take turns: crgpu_wave.h) */
        const uint4 c_i = n_i; const uint2 c_sym = n_sym; const uint32_t c_pred = n_pred;
        const uint32_t c_lo = n_lo, c_hi = n_hi; const bool c_first = n_first;
        if (!__builtin_amdgcn_ballot_w64(c_lo < c_hi || (r0.kind | r1.kind | r2.kind | r3.kind) != 0u)) break;
        cr_rop_o2_store(V, r0); cr_rop_o2_store(V, r1); cr_rop_o2_store(V, r2); cr_rop_o2_store(V, r3);
        fetch();
#define CR_O2R_POS(j_, out_, i_, sym_, pred_) \
        if ((j_) >= c_lo && (j_) < c_hi) { \
            if ((c_first && (j_) == c_lo) || fresh) cr_ln_clear(nd); \
            const uint32_t sy_ = (sym_); \
            cr_rop_o2_event_out(out_, nd, (i_), sy_ & 0x1ffu, (pred_)); \
            fresh = (sy_ >> 15) != 0u; \
        }
        CR_O2R_POS(0u, r0, c_i.x, c_sym.x & 0xffffu, c_pred & 0xffu)
        CR_O2R_POS(1u, r1, c_i.y, c_sym.x >> 16, (c_pred >> 8) & 0xffu)
        CR_O2R_POS(2u, r2, c_i.z, c_sym.y & 0xffffu, (c_pred >> 16) & 0xffu)
        CR_O2R_POS(3u, r3, c_i.w, c_sym.y >> 16, c_pred >> 24)
#undef CR_O2R_POS
        rounds++;
    }
    if (st && cr_lane() == 0) { st[2] = wall_clock64(); st[4] = rounds * 4u; }
}

CR_DEV void cr_rop_o3_ranges(CrEvViews& V, CrO2Ranges& R, uint32_t nev) {
    /* (round 5, last) A chain's first event is predicted with byte 0 (cr-ppm.c:66-88: a fresh entry), and so are all events until a
     * byte has been seen twice — more than half of the bench corpus's events, all of the Markov stream's. The kernels that lay out the
     * slots (k_rop_links*) clear every slot's prediction in passing, in order, and a lane here only stores the ones that are not 0:
     * every store it skips is a request less on the CU's address path, which is what this kernel is bound by. (Cleared HERE, with a
     * fence in front of the walk, the order-3 pass was slower on the bench step: 0.42 -> 0.54 ms.) */
    cr_rop_o2_ranges_build(V.csym3, R, nev);
    const uint32_t norder = R.norder;
    uint32_t fat = 0, fend = 0;
    bool more = true;
    /* groups of EIGHT slots here (a step is ten instructions: a round is as long as its fetch takes, so fewer, larger rounds) */
    uint4 n_sa = make_uint4(0u, 0u, 0u, 0u), n_sb = n_sa, n_sym = n_sa;
    uint32_t n_lo = 0, n_hi = 0;
    const auto fetch = [&]() __attribute__((always_inline)) {
        if (fat >= fend && more) {
            const uint32_t k = atomicAdd(&R.next, 1u);
            more = k < norder;
            if (more) { const uint32_t c = R.order[k]; fat = cr_o2r_start(R, c); fend = cr_o2r_start(R, c + 1u); }
        }
        const bool has = fat < fend;
        const uint32_t g = has ? fat >> 3 : 0u;
        const uint32_t left = fend - 8u * g;
        n_lo = has ? fat & 7u : 0u; n_hi = has ? (left < 8u ? left : 8u) : 0u;
        n_sa = reinterpret_cast<const uint4*>(V.cslot3)[2u * g];
        n_sb = reinterpret_cast<const uint4*>(V.cslot3)[2u * g + 1u];
        n_sym = reinterpret_cast<const uint4*>(V.csym3)[g];
        if (has) fat = 8u * g + n_hi;
    };
    fetch();
    uint32_t pred = 0, conf = 0;
    bool fresh = true;
    uint32_t w_s0 = 0, w_s1 = 0, w_s2 = 0, w_s3 = 0, w_s4 = 0, w_s5 = 0, w_s6 = 0, w_s7 = 0;   /* the previous round's predictions, still to be stored */
    uint32_t w_pa = 0, w_pb = 0, w_mask = 0;
    for (;;) {
        const uint4 c_sa = n_sa, c_sb = n_sb, c_sym = n_sym;
        const uint32_t c_lo = n_lo, c_hi = n_hi;
        if (!__builtin_amdgcn_ballot_w64(c_lo < c_hi || w_mask != 0u)) break;
        if ((w_mask & 1u) && (w_pa & 0xffu)) V.cpred[w_s0] = (uint8_t)w_pa;
        if ((w_mask & 2u) && (w_pa & 0xff00u)) V.cpred[w_s1] = (uint8_t)(w_pa >> 8);
        if ((w_mask & 4u) && (w_pa & 0xff0000u)) V.cpred[w_s2] = (uint8_t)(w_pa >> 16);
        if ((w_mask & 8u) && (w_pa >> 24)) V.cpred[w_s3] = (uint8_t)(w_pa >> 24);
        if ((w_mask & 16u) && (w_pb & 0xffu)) V.cpred[w_s4] = (uint8_t)w_pb;
        if ((w_mask & 32u) && (w_pb & 0xff00u)) V.cpred[w_s5] = (uint8_t)(w_pb >> 8);
        if ((w_mask & 64u) && (w_pb & 0xff0000u)) V.cpred[w_s6] = (uint8_t)(w_pb >> 16);
        if ((w_mask & 128u) && (w_pb >> 24)) V.cpred[w_s7] = (uint8_t)(w_pb >> 24);
        w_mask = 0; w_pa = 0; w_pb = 0;
        fetch();
#define CR_O3R_POS(j_, slot_, sym_, wslot_, wp_) \
        if ((j_) >= c_lo && (j_) < c_hi) { \
            if (fresh) { pred = 0; conf = 0; } \
            const uint32_t sy_ = (sym_), sym1_ = sy_ & 0x1ffu; \
            wslot_ = (slot_); wp_ |= pred << (8u * ((j_) & 3u)); w_mask |= 1u << (j_); \
            if (sym1_ == pred) { \
                conf += conf < 15u ? 1u : 0u; \
            } else { \
                uint32_t c_ = (uint32_t)(conf > 1u) + (uint32_t)(conf > 2u) + (uint32_t)(conf > 4u) + (uint32_t)(conf > 8u); \
                if (c_ == 0u) { pred = sym1_; c_ = 1u; } \
                conf = c_; \
            } \
            fresh = (sy_ >> 15) != 0u; \
        }
        CR_O3R_POS(0u, c_sa.x, c_sym.x & 0xffffu, w_s0, w_pa)
        CR_O3R_POS(1u, c_sa.y, c_sym.x >> 16, w_s1, w_pa)
        CR_O3R_POS(2u, c_sa.z, c_sym.y & 0xffffu, w_s2, w_pa)
        CR_O3R_POS(3u, c_sa.w, c_sym.y >> 16, w_s3, w_pa)
        CR_O3R_POS(4u, c_sb.x, c_sym.z & 0xffffu, w_s4, w_pb)
        CR_O3R_POS(5u, c_sb.y, c_sym.z >> 16, w_s5, w_pb)
        CR_O3R_POS(6u, c_sb.z, c_sym.w & 0xffffu, w_s6, w_pb)
        CR_O3R_POS(7u, c_sb.w, c_sym.w >> 16, w_s7, w_pb)
#undef CR_O3R_POS
    }
}

/* every lane keeps pulling chains from a shared counter (chains differ in length by three orders of
 * magnitude: a lane that finishes a short one must not idle behind a long one). A chain is a strictly
 * serial run, so what bounds the kernel is the latency of one step: the node never leaves LDS and the
 * chain's events are read sequentially, one step ahead of their use. */
CR_DEV void cr_rop_o2_all(CrEvViews& V, uint8_t* lane_node, uint32_t nchains, uint32_t* next_chain) {
    CrLaneNode nd;
    nd.cnt = lane_node; nd.bytes = 0; nd.fh = 1; nd.fe = 1;
    uint32_t at = 0, end = 0;                 /* slots [at, end) of the current chain are still to do */
    uint32_t n_i = 0, n_sp = 0;               /* event number and sym | pred << 16 of slot `at` (prefetched) */
    for (;;) {
        if (at == end) {
            const uint32_t h = atomicAdd(next_chain, 1u);
            if (h >= nchains) break;
            const u64 ch = V.chains2[h];
            at = (uint32_t)ch; end = (uint32_t)(ch >> 32);
            cr_ln_clear(nd);
            n_i = V.list2[at]; n_sp = (uint32_t)V.csym2[at] | ((uint32_t)V.cpred[at] << 16);
        }
        const uint32_t i = n_i, sp = n_sp;
        at++;
        if (at < end) { n_i = V.list2[at]; n_sp = (uint32_t)V.csym2[at] | ((uint32_t)V.cpred[at] << 16); }
        cr_rop_o2_event(V, nd, i, sp & 0x1ffu, sp >> 16);
    }
}

/* ------------------------------------------------------------------ k_rop_o1 */

/* The order-1 step of the escapes (cr-ppm.c:148-157, update cr-ppm.c:90-97). A row of the order-1
 * table only changes at escapes whose previous byte selects it, so the escapes are grouped by row
 * (one counting-sort pass, coding order kept) and each row is run by one wave with the row's 256
 * counts in registers, one word per lane. Rows are taken longest first. Result per escape: the
 * second range-coder triple, stored over the first two words of its exclusion set. */
#define CR_O1_THREADS CR_SORT_THREADS
CR_DEV void cr_rop_o1_row(CrEvViews& V, uint32_t* lds_masks /* [64][8] of this wave */, uint32_t at, uint32_t end) {
    const uint32_t lane = cr_lane();
    uint32_t row = 0x01010101u;                                          /* ppm_init: every count 1 */
    /* software pipeline over batches of 64 escapes: while batch b is coded the operands of b+1 load */
    u64 e_next = 0; uint32_t sy_next = 0; u64 tr_next = 0; uint4 m0_next = make_uint4(0, 0, 0, 0), m1_next = m0_next;
    if (at + lane < end) e_next = V.escB[at + lane];
    {
        const uint32_t i = (uint32_t)e_next;
        if (at + lane < end) {
            sy_next = V.ev_sym[i]; tr_next = V.trip[i];
            const bool empty = ((uint32_t)(e_next >> 56) & 1u) != 0u;                  /* CR_TRIP_EMPTY: no set in memory */
            const uint4* mp = reinterpret_cast<const uint4*>(V.mask + (empty ? (u64)0 : (u64)i * 8u));
            m0_next = mp[0]; m1_next = mp[1];
            if (empty) { m0_next = make_uint4(0u, 0u, 0u, 0u); m1_next = m0_next; }
        }
    }
    u64 e_after = 0;
    if (at + 64u + lane < end) e_after = V.escB[at + 64u + lane];
    for (; at < end; at += 64u) {
        const uint32_t cnt = end - at < 64u ? end - at : 64u;
        const u64 e_cur = e_next; const uint32_t sy_cur = sy_next; const u64 tr_cur = tr_next;
        reinterpret_cast<uint4*>(lds_masks + lane * 8u)[0] = m0_next;
        reinterpret_cast<uint4*>(lds_masks + lane * 8u)[1] = m1_next;
        {   /* the predicted byte is excluded like a byte with a count (cr-ppm.c:150): it joins the escape's set here, once per
             * escape, instead of eight instructions in every round of the loop below (lane l staged escape l) */
            const uint32_t pr = ((uint32_t)(tr_cur >> 32) >> 20) & 0xffu;
            atomicOr(lds_masks + lane * 8u + (pr >> 5), 1u << (pr & 31u));
        }
        e_next = e_after;
        {
            const uint32_t i = (uint32_t)e_next;
            if (at + 64u + lane < end) {
                sy_next = V.ev_sym[i]; tr_next = V.trip[i];
                const bool empty = ((uint32_t)(e_next >> 56) & 1u) != 0u;
                const uint4* mp = reinterpret_cast<const uint4*>(V.mask + (empty ? (u64)0 : (u64)i * 8u));
                m0_next = mp[0]; m1_next = mp[1];
                if (empty) { m0_next = make_uint4(0u, 0u, 0u, 0u); m1_next = m0_next; }
            }
        }
        e_after = 0;
        if (at + 128u + lane < end) e_after = V.escB[at + 128u + lane];
        cr_lds_order();
        uint32_t res_lo = 0, res_hi = 0;                                 /* lane l keeps the triple of escape l */
        uint32_t m0_keep;                                                /* (the loop's v_writelane take their lane from m0: whatever the compiler keeps there is put back behind the loop) */
        asm volatile("s_mov_b32 %0, m0" : "=s"(m0_keep));
        for (uint32_t l = 0; l < cnt; l++) {
            const uint32_t sym = cr_lane_get(sy_cur, l) & 0x1ffu;
            const uint32_t mw = lds_masks[l * 8u + (lane >> 3)];      /* (read one round ahead, with the symbol: 2.46 -> 2.63 ms) */
            const uint32_t present = (mw >> ((lane & 7u) * 4u)) & 0xfu;          /* bit j: byte 4*lane+j has a count or is the predicted byte */
            /* bit k of the nibble -> bit 8k (the product puts bit k at k, k+7, k+14, k+21; 8k = k + 7k), then x 255 as a shift and a
             * subtraction (the compiler makes a 32-bit multiplication of it, a quarter-rate instruction) */
            const uint32_t ones = (present * 0x00204081u) & 0x01010101u;
            uint32_t keep;
            asm("v_lshlrev_b32 %0, 8, %1\n\tv_sub_u32 %0, %0, %1\n\tv_not_b32 %0, %0" : "=&v"(keep) : "v"(ones));
            /* one scan for both sums: the total is its last lane, the sum below the symbol is what lies in front of the
             * symbol's lane plus that lane's bytes below the symbol (every lane works the latter out for its own word) */
            const uint32_t mine = cr_o1_weight_sum(row, keep);
            const uint32_t part = cr_o1_weight_sum(row, keep & ((1u << ((sym & 3u) * 8u)) - 1u));
            const uint32_t incl = cr_scan_incl(mine);
            const uint32_t all = cr_lane_get(incl, 63);
            const uint32_t lo = cr_lane_get(incl - mine + part, sym >> 2);
            const uint32_t cur = cr_table_byte(row, sym);
            const uint32_t fo = cur * 8u - 7u;
            /* lo | all << 20 | fo << 40 into lane l: everything here is wave-uniform, so the two halves are put together on the
             * scalar unit and written into the lane (v_writelane) — as `if (lane == l) res = ...` it was a compare, a branch that
             * is never skipped and eight scalar instructions behind it. (The loop is what the kernel's time is: ~70 wave
             * instructions per escape, 16 M escapes, every SIMD busy.) */
            {
                const uint32_t w_lo = cr_uni(lo | (all << 20)), w_hi = cr_uni((all >> 12) | (fo << 8));
                asm volatile("s_mov_b32 m0, %4\n\tv_writelane_b32 %0, %2, m0\n\tv_writelane_b32 %1, %3, m0"
                             : "+v"(res_lo), "+v"(res_hi) : "s"(w_lo), "s"(w_hi), "s"(l));
            }
            /* ppm_update_o1, cr-ppm.c:90-97 */
            if (lane == (sym >> 2)) row += 1u << ((sym & 3u) * 8u);
            if (cur + 1u >= 255u) {                                      /* wave-uniform and rare: a branch, not five instructions every time */
                asm volatile("; o1_rescale");
                row -= (row >> 1) & 0x7f7f7f7fu;
            }
        }
        asm volatile("s_mov_b32 m0, %0" :: "s"(m0_keep));
        const u64 res = ((u64)res_hi << 32) | res_lo;
        if (lane < cnt) *reinterpret_cast<u64*>(V.mask + (u64)(uint32_t)e_cur * 8u) = res;
        cr_lds_order();
    }
}

/* The same row, 64 escapes per wave-step (round 5): lane k = escape k of the batch. Between two halvings a row is a pure
 * occurrence counter (cr-ppm.c:90-97), so escape k meets count[s] = base[s] + #{j < k : sym_j = s} — base = the row in front of
 * the batch — and with m_k = the 256-bit set of bytes NOT excluded for k (no count in its node, not the predicted byte)
 *     total_k = sum over m_k of (8 count - 7) = 8 <m_k, base> - 7 |m_k| + 8 #{j < k : sym_j in m_k}
 *     below_k = the same over m_k & {s < sym_k}              freq_k = 8 (base[sym_k] + #{j < k : sym_j = sym_k}) - 7.
 * <m, base> comes from BIT PLANES: the row lives in the wave as lane l = bytes {l, l + 64, l + 128, l + 192}, so a ballot of bit b
 * of byte j is, bit for bit, the plane of symbols 64 j .. 64 j + 63 in the mask's own order, and <m, base> = sum_b 2^b
 * popcount(m & plane_b): two instructions per mask word and plane, planes that are all zero (counts below 2^b) skipped.
 * The in-batch term is a loop over j (uniform): symbol j broadcast, every lane k > j tests it in its own set (the sets staged
 * in LDS, stride 9: the word index is uniform, the lanes are not). Equal symbols by bit-sliced ballots. A batch is cut behind
 * the first escape whose count reaches 255 (cr-ppm.c:94-96): the halving is applied and the rest starts the next batch.
 * ~15 wave instructions per escape instead of ~58. lds: 640 dwords per wave. */
#define CR_O1_LDS_PER_WAVE 640u
struct CrO1Ops { uint32_t ei, sym, pred; uint4 x0, x1; };
/* (unconditional loads at clamped indices: a load behind `if (valid)` makes every later wait a vmcnt(0)) */
CR_DEV void cr_o1_ops_load(CrO1Ops& o, const CrEvViews& V, u64 e) {
    o.ei = (uint32_t)e;
    o.sym = (uint32_t)(e >> 40) & 0xffu;                                 /* (the compaction put symbol and predicted byte into the record: two gathers less) */
    o.pred = (uint32_t)(e >> 48) & 0xffu;
    /* an escape out of a node without a byte has no set in memory (CR_TRIP_EMPTY): its lane reads the first slot like every other such
     * lane — one request for them all — and drops what it gets */
    const bool empty = ((uint32_t)(e >> 56) & 1u) != 0u;
    const uint4* mp = reinterpret_cast<const uint4*>(V.mask + (empty ? (u64)0 : (u64)o.ei * 8u));
    o.x0 = mp[0]; o.x1 = mp[1];
    if (empty) { o.x0 = make_uint4(0u, 0u, 0u, 0u); o.x1 = o.x0; }
}
CR_DEV void cr_rop_o1_row_batch(CrEvViews& V, uint32_t* lds /* [640] of this wave */, uint32_t at, uint32_t end) {
    const uint32_t lane = cr_lane();
    const u64 below_me = (1ull << lane) - 1ull;
    uint32_t row = 0x01010101u;                                          /* ppm_init: every count 1; lane l byte j = symbol l + 64 j */
    uint32_t* const my = lds + lane * 9u;
    const uint32_t last = end - 1u;
    /* software pipeline: while a batch is coded the operands of the next one load (and the event numbers of the one after) */
    CrO1Ops nx;
    u64 e_after;
    {
        const uint32_t i0 = at + lane < last ? at + lane : last, i1 = at + 64u + lane < last ? at + 64u + lane : last;
        const u64 e = V.escB[i0];
        e_after = V.escB[i1];
        cr_o1_ops_load(nx, V, e);
    }
    u64 res = 0; uint32_t res_ei = 0; bool res_mine = false;             /* a batch's triples are stored at the top of the next one: behind the wait for
                                                                          * that batch's operands, in front of the next loads (a store in between would be waited for too) */
    while (at < end) {
        const uint32_t cnt = end - at < 64u ? end - at : 64u;
        const bool valid = lane < cnt;
        const CrO1Ops op = nx;
        if (res_mine) *reinterpret_cast<u64*>(V.mask + (u64)res_ei * 8u) = res;
        cr_o1_ops_load(nx, V, e_after);
        {
            const uint32_t i2 = at + 128u + lane < last ? at + 128u + lane : last;
            e_after = V.escB[i2];
        }
        const uint32_t ei = op.ei, sym = valid ? (op.sym & 0xffu) : 0u;
        uint32_t m[8];
        {
            const uint32_t x[8] = {op.x0.x, op.x0.y, op.x0.z, op.x0.w, op.x1.x, op.x1.y, op.x1.z, op.x1.w};
            const uint32_t pw = op.pred >> 5, pb = 1u << (op.pred & 31u);
#pragma unroll
            for (uint32_t w = 0; w < 8u; w++) m[w] = valid ? ~(x[w] | (pw == w ? pb : 0u)) : 0u;   /* the predicted byte is excluded too (cr-ppm.c:150) */
        }
#pragma unroll
        for (uint32_t w = 0; w < 8u; w++) my[w] = m[w];
        /* the part of the set below the lane's own symbol */
        uint32_t ml[8];
#pragma unroll
        for (uint32_t w = 0; w < 8u; w++) {
            int d = (int)sym - (int)(32u * w);
            d = d < 0 ? 0 : (d > 32 ? 32 : d);
            ml[w] = m[w] & (uint32_t)((1ull << d) - 1ull);
        }
        /* <m, base> and <ml, base> plane by plane */
        uint32_t s_all = 0, s_lo = 0;
#pragma unroll
        for (uint32_t b = 0; b < 8u; b++) {
            const u64 p0 = cr_ballot((row >> b) & 1u), p1 = cr_ballot((row >> (8u + b)) & 1u);
            const u64 p2 = cr_ballot((row >> (16u + b)) & 1u), p3 = cr_ballot((row >> (24u + b)) & 1u);
            const uint32_t pl[8] = {(uint32_t)p0, (uint32_t)(p0 >> 32), (uint32_t)p1, (uint32_t)(p1 >> 32),
                                    (uint32_t)p2, (uint32_t)(p2 >> 32), (uint32_t)p3, (uint32_t)(p3 >> 32)};
            uint32_t a = 0, l = 0;
            if ((p0 | p1) != 0ull) {                                     /* (text never takes a byte above 127 past its first count: half the planes' words are zero) */
#pragma unroll
                for (uint32_t w = 0; w < 4u; w++) { a += (uint32_t)__builtin_popcount(m[w] & pl[w]); l += (uint32_t)__builtin_popcount(ml[w] & pl[w]); }
            }
            if ((p2 | p3) != 0ull) {
#pragma unroll
                for (uint32_t w = 4u; w < 8u; w++) { a += (uint32_t)__builtin_popcount(m[w] & pl[w]); l += (uint32_t)__builtin_popcount(ml[w] & pl[w]); }
            }
            s_all += a << b; s_lo += l << b;
        }
        uint32_t n_all = 0, n_lo = 0;
#pragma unroll
        for (uint32_t w = 0; w < 8u; w++) { n_all += (uint32_t)__builtin_popcount(m[w]); n_lo += (uint32_t)__builtin_popcount(ml[w]); }
        /* the escapes of the batch with the lane's symbol (same) and with a smaller one (less): a radix comparison of the eight
         * symbol bits by ballots, most significant first — less = "equal so far, and here its bit is 0 where mine is 1" */
        u64 same = cnt == 64u ? ~0ull : (1ull << cnt) - 1ull, less = 0;
#pragma unroll
        for (uint32_t b = 8u; b-- > 0u;) {
            const u64 bal = cr_ballot((sym >> b) & 1u);
            const bool mine = (sym >> b) & 1u;
            less |= mine ? (same & ~bal) : 0ull;
            same &= mine ? bal : ~bal;
        }
        const uint32_t before = (uint32_t)__builtin_popcountll(same & below_me);
        const uint32_t base_w = (uint32_t)__builtin_amdgcn_ds_bpermute((int)((sym & 63u) << 2), (int)row);
        const uint32_t cur = ((base_w >> ((sym >> 6) * 8u)) & 0xffu) + before;
        /* cut behind the first escape that takes its count to 255 */
        const u64 halving = cr_ballot(valid && cur + 1u >= 255u);
        const uint32_t take = halving ? (uint32_t)__builtin_ctzll(halving) + 1u : cnt;
        /* which of the escapes in front hold a symbol of the lane's own set: one test per DISTINCT symbol of the batch (a row of
         * text repeats its symbols: ~20-30 distinct ones in 64 escapes) — the symbol's first escape f broadcasts it, every lane
         * tests it in its set staged in LDS (the word index is uniform, the lanes are not) and takes the mask of the escapes
         * that hold it (lane f's `same`) or nothing; two symbols at a time, their LDS reads in flight together */
        uint32_t in_lo = 0, in_hi = 0;
        cr_lds_order_sw();
        const uint32_t nj = take - 1u;                                   /* escape take - 1 is in front of nobody */
        u64 firsts = cr_ballot((same & below_me) == 0ull) & ((1ull << nj) - 1ull);
        const uint32_t same_lo = (uint32_t)same, same_hi = (uint32_t)(same >> 32);
        while (firsts) {
            const uint32_t f0 = (uint32_t)__builtin_ctzll(firsts);
            firsts &= firsts - 1ull;
            const uint32_t f1 = firsts ? (uint32_t)__builtin_ctzll(firsts) : f0;
            firsts &= firsts - 1ull;                                     /* (0 & -1 = 0) */
            const uint32_t s0 = cr_lane_get(sym, f0), s1 = cr_lane_get(sym, f1);
            const uint32_t w0 = my[s0 >> 5], w1 = my[s1 >> 5];
            const uint32_t t0 = 0u - __builtin_amdgcn_ubfe(w0, s0 & 31u, 1u), t1 = 0u - __builtin_amdgcn_ubfe(w1, s1 & 31u, 1u);
            in_lo |= (t0 & cr_lane_get(same_lo, f0)) | (t1 & cr_lane_get(same_lo, f1));
            in_hi |= (t0 & cr_lane_get(same_hi, f0)) | (t1 & cr_lane_get(same_hi, f1));
        }
        const u64 in_front = (((u64)in_hi << 32) | in_lo) & below_me;      /* (a batch of fewer than 64: the lanes behind it hold symbol 0 and an empty set) */
        const uint32_t packed = (uint32_t)__builtin_popcountll(in_front) | ((uint32_t)__builtin_popcountll(in_front & less) << 16);
        const uint32_t all = 8u * (s_all + (packed & 0xffffu)) - 7u * n_all;
        const uint32_t lo = 8u * (s_lo + (packed >> 16)) - 7u * n_lo;
        const uint32_t fo = cur * 8u - 7u;
        res = (u64)(lo | (all << 20)) | ((u64)((all >> 12) | (fo << 8)) << 32); res_ei = ei; res_mine = lane < take;
        /* ppm_update_o1 for the escapes taken: every symbol's last escape adds its occurrences to the row */
        const u64 same_t = same & (take == 64u ? ~0ull : (1ull << take) - 1ull);
        cr_lds_order();
        lds[576u + lane] = 0u;
        cr_lds_order_sw();
        if (lane < take && (same_t >> lane) == 1ull)
            atomicAdd(&lds[576u + (sym & 63u)], (uint32_t)__builtin_popcountll(same_t) << ((sym >> 6) * 8u));
        cr_lds_order();
        row += lds[576u + lane];
        at += take;
        if (halving) {
            asm volatile("; o1_rescale (batch)");
            row -= (row >> 1) & 0x7f7f7f7fu;
            /* the batches behind a cut do not start where the loads in flight assumed: fetch again (a few times per long row) */
            const uint32_t i0 = at + lane < last ? at + lane : last, i1 = at + 64u + lane < last ? at + 64u + lane : last;
            const u64 e = V.escB[i0];
            e_after = V.escB[i1];
            cr_o1_ops_load(nx, V, e);
        }
        cr_lds_order();
    }
    if (res_mine) *reinterpret_cast<u64*>(V.mask + (u64)res_ei * 8u) = res;
}

CR_DEV void cr_rop_o1_all(CrSortShared& sh, CrEvViews& V, uint32_t* lds_masks, uint32_t nev, uint32_t serial, u64* st = nullptr) {
    const uint32_t t = threadIdx.x, w = cr_wave_id(), lane = cr_lane();
    cr_wg_stamp(st, 12);
    /* 1: the escapes, in coding order; wave w owns a contiguous quarter of the events */
    const uint32_t nchunks = (nev + 63u) >> 6, cpw = (nchunks + CR_SORT_WAVES - 1u) / CR_SORT_WAVES;
    const uint32_t c_lo = w * cpw < nchunks ? w * cpw : nchunks;
    const uint32_t c_hi = c_lo + cpw < nchunks ? c_lo + cpw : nchunks;
    uint32_t mine = 0;
    for (uint32_t c0 = c_lo; c0 < c_hi; c0 += 8u) {
        uint32_t ty[8];
#pragma unroll
        for (uint32_t u = 0; u < 8u; u++) {
            const uint32_t i = (c0 + u) * 64u + lane;
            ty[u] = 0;
            if (c0 + u < c_hi && i < nev) ty[u] = reinterpret_cast<const uint32_t*>(V.trip + i)[1];
        }
#pragma unroll
        for (uint32_t u = 0; u < 8u; u++) mine += (uint32_t)__builtin_popcountll(cr_ballot(((ty[u] >> 18) & 3u) == CR_T_ESC));
    }
    __syncthreads();
    if (lane == 0) sh.wsum[w] = mine;
    __syncthreads();
    uint32_t at = 0, nesc = 0;
    for (uint32_t ww = 0; ww < CR_SORT_WAVES; ww++) { if (ww < w) at += sh.wsum[ww]; nesc += sh.wsum[ww]; }
    for (uint32_t c0 = c_lo; c0 < c_hi; c0 += 8u) {
        uint32_t ty[8], cx[8];
#pragma unroll
        for (uint32_t u = 0; u < 8u; u++) {
            const uint32_t i = (c0 + u) * 64u + lane;
            ty[u] = 0; cx[u] = 0;
            if (c0 + u < c_hi && i < nev) {                              /* row | symbol << 8 of the event, its type */
                ty[u] = reinterpret_cast<const uint32_t*>(V.trip + i)[1];
                cx[u] = (V.ev_ctx[i] & 0xffu) | ((uint32_t)(V.ev_sym[i] & 0xffu) << 8) | (((ty[u] >> 20) & 0x1ffu) << 16);  /* | predicted byte << 16 | CR_TRIP_EMPTY << 24 */
            }
        }
#pragma unroll
        for (uint32_t u = 0; u < 8u; u++) {
            const bool is_esc = ((ty[u] >> 18) & 3u) == CR_T_ESC;
            const u64 em = cr_ballot(is_esc);
            /* record = event | row << 32 | symbol << 40 | predicted byte << 48 (the sort's digit is the row's byte): the rows
             * then gather nothing but an escape's exclusion set */
            if (is_esc) V.escA[at + (uint32_t)__builtin_popcountll(em & ((1ull << lane) - 1ull))] = ((u64)cx[u] << 32) | ((c0 + u) * 64u + lane);
            at += (uint32_t)__builtin_popcountll(em);
        }
    }
    cr_wg_sync_global();
    cr_wg_stamp(st, 13);
    /* 2: row after row */
    const uint32_t rstart = cr_sort_pass(sh, nesc, 0u, 0, V.escA, nullptr, V.escB, V.thist, nullptr);
    sh.goff[t] = rstart;
    __syncthreads();
    const uint32_t rlen = (t == 255u ? nesc : sh.goff[t + 1u]) - rstart;
    sh.whist[t] = rlen;
    if (t == 0) sh.front = 0;
    __syncthreads();
    uint32_t rank = 0;
    for (uint32_t u = 0; u < 256u; u++) { const uint32_t lu = sh.whist[u]; rank += (lu > rlen || (lu == rlen && u < t)) ? 1u : 0u; }
    sh.tstart[rank] = t;
    __syncthreads();
    cr_wg_stamp(st, 14);
    if (st && t == 0) { st[11] = nesc; }
    if (st && rank == 0) st[10] = rlen;                                  /* the longest row */
    /* 3: waves take rows, longest first */
    for (;;) {
        uint32_t k = 0;
        if (lane == 0) k = atomicAdd(&sh.front, 1u);
        k = cr_uni(k);
        if (k >= 256u) break;
        const uint32_t r = sh.tstart[k];
        const uint32_t s0 = sh.goff[r], len = sh.whist[r];
        if (len == 0u) break;                                             /* rows are sorted by length */
        /* rows of a dozen escapes and more in batches of 64 (a batch costs what ~8 escapes of the serial loop cost) */
        const u64 row_t0 = st ? __builtin_amdgcn_s_memrealtime() : 0;
        if (serial || len < 12u) cr_rop_o1_row(V, lds_masks + w * CR_O1_LDS_PER_WAVE, s0, s0 + len);
        else cr_rop_o1_row_batch(V, lds_masks + w * CR_O1_LDS_PER_WAVE, s0, s0 + len);
        if (st && k == 0 && lane == 0) st[9] = __builtin_amdgcn_s_memrealtime() - row_t0;   /* the longest row alone (100 MHz ticks) */
    }
    __syncthreads();
    cr_wg_stamp(st, 15);
}

/* ------------------------------------------------------------------ k_rop_rc */

/* 64-event register window over the per-event arrays */
struct CrEvWindow {
    uint32_t base;
    u64 trip, trip2;           /* trip2: the escape's order-1 triple lo | all << 20 | frq << 40 */
    uint32_t sym;
};
CR_DEV void cr_evwin_fill(CrEvWindow& w, const CrEvViews& V, uint32_t at, uint32_t nev) {
    w.base = at;
    const uint32_t i = at + cr_lane();
    w.trip = 0; w.trip2 = 0; w.sym = 0;
    if (i < nev) {
        w.trip = V.trip[i]; w.sym = V.ev_sym[i];
        if (((uint32_t)(w.trip >> 50) & 3u) == CR_T_ESC) w.trip2 = *reinterpret_cast<const u64*>(V.mask + (u64)i * 8u);
    }
    cr_drain_loads();
}

/* range coder over the prepared triples + output, one event after the other (cr-rangecoder.c:60-70) */
/* the coded triples of a block, event by event, into `body`; `header` = the bytes the block format puts in front of
 * the stream (the "output not smaller than the input" test of the token loop counts them: ropmain/cr-coder.c:204-206,
 * roxmain/cr-coder.c:273-275). Returns the stream's size or 0xFFFFFFFF when that test fired (block is stored). */
CR_DEV uint32_t cr_code_events(uint32_t n, uint8_t* body, uint32_t header, CrEvViews& V) {
    const uint32_t nev = cr_uni(V.ctr[0]);
    CrSink out; out.dst = body; out.n = 0;
    CrRc rc; cr_rc_init(rc);
    CrEvWindow w;
    cr_evwin_fill(w, V, 0, nev);
    for (uint32_t i = 0; i < nev; i++) {
        if (i - w.base >= CRGPU_WAVE) cr_evwin_fill(w, V, i, nev);
        const uint32_t l = i - w.base;
        const u64 t = cr_lane_get64(w.trip, l);
        const uint32_t cum = (uint32_t)t & 0xfffffu, tot = (uint32_t)(t >> 20) & 0xfffffu, frq = (uint32_t)(t >> 40) & 0x3ffu, type = (uint32_t)(t >> 50) & 3u;
        cr_rc_pin(rc); out.n = cr_uni(out.n);
        cr_rc_encode(rc, cum, frq, tot, out);
        if (type == CR_T_ESC) {
            const u64 t2 = cr_lane_get64(w.trip2, l);
            cr_rc_pin(rc); out.n = cr_uni(out.n);
            cr_rc_encode(rc, (uint32_t)t2 & 0xfffffu, (uint32_t)(t2 >> 40) & 0xfffu, (uint32_t)(t2 >> 20) & 0xfffffu, out);
        }
        if ((cr_lane_get(w.sym, l) & CR_EV_LAST) && header + out.n >= n) return 0xFFFFFFFFu;
    }
    cr_rc_pin(rc);
    cr_rc_flush(rc, out);
    return cr_uni(out.n);
}

CR_DEV void cr_rop_write_header(const uint8_t* src, uint32_t n, uint32_t esc, uint8_t* dst) {   /* cr-coder.c:213-216 */
    const uint32_t lane = cr_lane();
    if (lane < CR_ROP_HEADER) {
        uint32_t v = 0;
        if (lane == 0) v = 1;
        else if (lane >= 4 && lane < 8) v = (n >> (8u * (lane - 4u))) & 0xffu;
        else if (lane == 8) v = esc;
        else if (lane >= 9 && lane < 18) v = src[lane - 9u];
        dst[lane] = (uint8_t)v;
    }
}

CR_DEV uint32_t cr_rop_code_events(const uint8_t* src, uint32_t n, uint8_t* dst, CrEvViews& V) {
    const uint32_t info = cr_uni(V.ctr[3]);
    if (info & 0x100u) { cr_rop_store_raw(src, n, dst); return CR_ROP_HEADER + n; }
    const uint32_t got = cr_code_events(n, dst + CR_ROP_HEADER, CR_ROP_HEADER, V);
    if (got == 0xFFFFFFFFu) { cr_wave_sync(); cr_rop_store_raw(src, n, dst); return CR_ROP_HEADER + n; }
    cr_rop_write_header(src, n, info & 0xffu, dst);
    return CR_ROP_HEADER + got;
}

/* ---- the same, restructured: only the RANGE is a serial recurrence.
 * cr-rangecoder.c:60-70 per triple:  unit = range / tot;  low += cum * unit;  range = unit * frq;
 * then shift out a byte while range < 2^24. `low` never feeds back into `range`, and the emitted
 * stream is simply the big number  sum_k (cum_k * unit_k) * 256^-(S_k + 5)  (S_k = bytes shifted out
 * before triple k; cache / follow / carry in cr-rangecoder.c:44-58 are how a serial coder adds with
 * carry). So one scalar chain produces unit_k for 64 events at a time — division by multiplication
 * with a per-lane precomputed reciprocal, scalar ALU only — and everything else is per lane: the
 * products, the byte positions (prefix sum of the shift counts), and the additions into 64-bit
 * accumulators, one per output word, in an LDS ring; a final carry pass turns them into bytes. */
#define CR_RC_RING 1024u     /* output words in flight: eight windows of 128 triples shift out at most 8 x 96 words before they are retired */

CR_DEV uint32_t cr_rc_magic(uint32_t tot) {          /* floor(2^32 / tot), tot = 1 saturates (the step corrects by one) */
    return tot <= 1u ? 0xFFFFFFFFu : (uint32_t)(4294967296.0 / (double)tot);
}
/* one step of the chain; everything here is wave-uniform and meant for the scalar ALU */
CR_DEV uint32_t cr_rc_chain_step(uint32_t& range, uint32_t tot, uint32_t frq, uint32_t magic) {
    /* q = umulhi(range, magic); if (range - q * tot >= tot) q++; range = q * frq shifted up by whole bytes.
     * Nine scalar instructions (the compiler spends four on the conditional increment; the compare's SCC is the carry) */
    uint32_t q, t;
    asm("s_mul_hi_u32 %0, %2, %5\n\t"
        "s_mul_i32 %1, %0, %3\n\t"
        "s_sub_u32 %1, %2, %1\n\t"
        "s_cmp_ge_u32 %1, %3\n\t"
        "s_addc_u32 %0, %0, 0\n\t"
        "s_mul_i32 %1, %0, %4\n\t"
        "s_flbit_i32_b32 %2, %1\n\t"
        "s_and_b32 %2, %2, 24\n\t"
        "s_lshl_b32 %2, %1, %2"
        : "=&s"(q), "=&s"(t), "+s"(range) : "s"(tot), "s"(frq), "s"(magic) : "scc");
    return q;
}
CR_DEV uint32_t cr_rc_shifts(uint32_t raw_range) { return (uint32_t)__builtin_clz(raw_range) >> 3; }

struct CrRcWin { u64 t, t2; };
/* Straight-line on purpose: no branch around the loads and no type test between them. The compiler counts outstanding loads in
 * order; with the loads under `if (i < nev)` it cannot know how many newer ones are in flight when the chain needs the older
 * window and waits for ALL of them (s_waitcnt vmcnt(0) right behind the prefetch: a full memory round trip per window — k_rop_rc
 * was parked 36 % of its time), and an order-1 triple fetched only behind an escape is a second round trip. So: the index is
 * clamped, both words are fetched for every event, and cr_rcwin_esc sorts out what they mean when the window is used. The window
 * loop runs two windows per round on two register slots, each reloaded for the window after next as soon as its fields are taken
 * out: no register copies that would wait for the newest loads, and two chains of 64 events to hide a fetch behind. */
CR_DEV CrRcWin cr_rcwin_load(const CrEvViews& V, uint32_t at, uint32_t nev) {
    CrRcWin w;
    const uint32_t i = at + cr_lane();
    const uint32_t ic = i < nev ? i : (nev ? nev - 1u : 0u);
    w.t = V.trip[ic];
    w.t2 = *reinterpret_cast<const u64*>(V.mask + (u64)ic * 8u);
    return w;
}
CR_DEV void cr_rcwin_esc(CrRcWin& w, uint32_t at, uint32_t nev) {
    if (at + cr_lane() >= nev) w.t = (1ull << 20) | (1ull << 40);  /* padding: cum 0, tot 1, frq 1 changes nothing */
    if (((uint32_t)(w.t >> 50) & 3u) != CR_T_ESC) w.t2 = 0;
}

/* the stream into `body` (whole big-endian words: up to 3 bytes behind its end are written too); returns its size,
 * 0xFFFFFFFF when the token loop's size test (header + bytes written so far >= n, checked after every token) is certain
 * to have fired — the block is stored —, or 0 when only the event-by-event coder can tell.
 * The serial coder's count at the last token is S - f (S = bytes shifted out, f = its `follow`: the bytes 0xFF still
 * waiting for a carry, cr-rangecoder.c:44-58). It only grows, so the test fires somewhere iff it fires at the last token.
 * f is at most the run of equal 0xFF (or, had the carry come, 0x00) bytes that ends at stream position S. */
CR_DEV uint32_t cr_code_events_fast(uint32_t n, uint8_t* body, uint32_t header, CrEvViews& V, u64* ring /* LDS [CR_RC_RING] */) {
    const uint32_t lane = cr_lane();
    const uint32_t nev = cr_uni(V.ctr[0]), info = cr_uni(V.ctr[3]);
    if (info & 0x100u) return 0u;
    u64* accw = V.escA;                                  /* one 64-bit sum per output word (dead scratch of the order-1 pass) */
    for (uint32_t k = lane; k < CR_RC_RING; k += CRGPU_WAVE) ring[k] = 0;
    uint32_t range = 0xFFFFFFFFu, sbase = 0, wret = 0;
    CrRcWin slotA = cr_rcwin_load(V, 0, nev), slotB = cr_rcwin_load(V, CRGPU_WAVE, nev);
    const auto window = [&](CrRcWin& slot, const uint32_t at) __attribute__((always_inline)) {
        CrRcWin w = slot;
        cr_rcwin_esc(w, at, nev);
        const uint32_t cumA = (uint32_t)w.t & 0xfffffu, totA = (uint32_t)(w.t >> 20) & 0xfffffu, frqA = (uint32_t)(w.t >> 40) & 0x3ffu;
        const bool esc = ((uint32_t)(w.t >> 50) & 3u) == CR_T_ESC;
        const uint32_t cumB = (uint32_t)w.t2 & 0xfffffu, totB = (uint32_t)(w.t2 >> 20) & 0xfffffu, frqB = (uint32_t)(w.t2 >> 40) & 0xfffu;
        const uint32_t mA = cr_rc_magic(totA), mB = cr_rc_magic(totB);
        slot = cr_rcwin_load(V, at + 2u * CRGPU_WAVE, nev);     /* the slot's fields are out: it fetches the window after next */
        const u64 em = cr_ballot(esc);
        const uint32_t em_lo = cr_uni((uint32_t)em), em_hi = cr_uni((uint32_t)(em >> 32));
        uint32_t qA = 0, qB = 0;
        /* the operands of four events are read out of their lanes in one go, ahead of the chain: a scalar instruction that
         * consumes a v_readlane result straight away waits ~12 clocks for it */
#define CR_RC_GET(l, k) uint32_t t##k##_, f##k##_, m##k##_; \
            asm volatile("v_readlane_b32 %0, %3, " #l "\n\tv_readlane_b32 %1, %4, " #l "\n\tv_readlane_b32 %2, %5, " #l \
                         : "=s"(t##k##_), "=s"(f##k##_), "=s"(m##k##_) : "v"(totA), "v"(frqA), "v"(mA));
#define CR_RC_STEP(l, k) { \
            const uint32_t q_ = cr_rc_chain_step(range, t##k##_, f##k##_, m##k##_); \
            asm volatile("v_writelane_b32 %0, %1, " #l : "+v"(qA) : "s"(q_)); \
            if (((l) < 32 ? em_lo : em_hi) & (1u << ((l) & 31))) { \
                const uint32_t q2_ = cr_rc_chain_step(range, cr_lane_get(totB, l), cr_lane_get(frqB, l), cr_lane_get(mB, l)); \
                asm volatile("v_writelane_b32 %0, %1, " #l : "+v"(qB) : "s"(q2_)); \
            } }
#define CR_RC_STEP4(a, b, c, d) { CR_RC_GET(a, 0) CR_RC_GET(b, 1) CR_RC_GET(c, 2) CR_RC_GET(d, 3) CR_RC_STEP(a, 0) CR_RC_STEP(b, 1) CR_RC_STEP(c, 2) CR_RC_STEP(d, 3) }
#define CR_RC_STEP8(b) CR_RC_STEP4(b##0, b##1, b##2, b##3) CR_RC_STEP4(b##4, b##5, b##6, b##7)
        /* (v_readlane needs one wait state behind the vector instruction that wrote its source; the compiler does not see
         * into the statements below) */
        asm volatile("s_nop 0" :: "v"(totA), "v"(frqA), "v"(mA));
        /* lanes 0..63 written as octal literals 00..077 so that each step names its lane as an immediate */
        CR_RC_STEP8(00) CR_RC_STEP8(01) CR_RC_STEP8(02) CR_RC_STEP8(03) CR_RC_STEP8(04) CR_RC_STEP8(05) CR_RC_STEP8(06) CR_RC_STEP8(07)
#undef CR_RC_STEP8
#undef CR_RC_STEP4
#undef CR_RC_STEP
#undef CR_RC_GET
        /* per lane: where the two products land */
        const uint32_t shA = cr_rc_shifts(qA * frqA), shB = esc ? cr_rc_shifts(qB * frqB) : 0u;
        const uint32_t incl = cr_scan_incl(shA + shB);
        const uint32_t sA = sbase + incl - (shA + shB), sB = sA + shA;
        sbase += cr_lane_get(incl, 63);
        {
            const uint32_t d = cumA * qA, b = sA + 1u, o = (b & 3u) * 8u;
            if (d) {
                atomicAdd(reinterpret_cast<unsigned long long*>(ring + ((b >> 2) & (CR_RC_RING - 1u))), (unsigned long long)(d >> o));
                if (o) atomicAdd(reinterpret_cast<unsigned long long*>(ring + (((b >> 2) + 1u) & (CR_RC_RING - 1u))), (unsigned long long)(d << (32u - o)));
            }
        }
        if (esc) {
            const uint32_t d = cumB * qB, b = sB + 1u, o = (b & 3u) * 8u;
            if (d) {
                atomicAdd(reinterpret_cast<unsigned long long*>(ring + ((b >> 2) & (CR_RC_RING - 1u))), (unsigned long long)(d >> o));
                if (o) atomicAdd(reinterpret_cast<unsigned long long*>(ring + (((b >> 2) + 1u) & (CR_RC_RING - 1u))), (unsigned long long)(d << (32u - o)));
            }
        }
    };
    /* Words no later triple can reach go to memory every eight windows, not every window: on this target stores count in vmcnt
     * like loads, so a store loop of unknown length between the prefetch and its use makes the compiler wait for everything. The
     * inner loop has no stores and the same four loads in flight on every path: the waits are exact (vmcnt(2)). A window past the
     * end is all padding and changes nothing. */
    for (uint32_t at = 0; at < nev;) {
        cr_take_turns<0>(at >> 13);                         /* (every 8 192 events — 1.83 -> 1.68 ms; two of these chains on a SIMD take turns on its scalar unit) */
        for (uint32_t r = 0; r < 4u && at < nev; r++, at += 2u * CRGPU_WAVE) {
            window(slotA, at);
            window(slotB, at + CRGPU_WAVE);
        }
        cr_lds_order();
        const uint32_t wnew = (sbase + 1u) >> 2;
        for (uint32_t k = wret + lane; k < wnew; k += CRGPU_WAVE) { accw[k] = ring[k & (CR_RC_RING - 1u)]; ring[k & (CR_RC_RING - 1u)] = 0; }
        wret = wnew;
        cr_lds_order();
        __builtin_amdgcn_s_waitcnt(0x0F70);              /* vmcnt(0): the inner loop starts with nothing it cannot count */
    }
    const uint32_t stotal = sbase + 5u;                                   /* cr-rangecoder.c:72-79 */
    const bool maybe_stored = header + sbase >= n;                        /* the size test of the token loop could have fired */
    /* (such a stream may be longer than the block's output slot, which is header + n for comprop: it goes to scratch —
     * escB, dead since the order-1 pass — and only its end is looked at) */
    uint8_t* const wout = maybe_stored ? reinterpret_cast<uint8_t*>(V.escB) : body;
    const uint32_t wtotal = (stotal + 3u) >> 2;
    for (uint32_t k = wret + lane; k < wtotal; k += CRGPU_WAVE) accw[k] = ring[k & (CR_RC_RING - 1u)];
    cr_wave_sync();
    /* carries, from the last word to the first, 64 words per step */
    uint32_t c_in = 0;          /* upper half of the word to the right of this step */
    uint32_t bit_in = 0;        /* carry out of that word after its own additions */
    for (uint32_t k0 = (wtotal - 1u) & ~63u;; k0 -= 64u) {
        const uint32_t k = k0 + lane;
        const u64 v = k < wtotal ? accw[k] : 0ull;
        const uint32_t up = (uint32_t)(v >> 32);
        uint32_t right = (uint32_t)__shfl_down((int)up, 1);
        if (lane == 63u) right = c_in;
        const u64 v2 = (v & 0xffffffffull) + right;
        const uint32_t word = (uint32_t)v2;
        const u64 g = __builtin_bitreverse64(cr_ballot((v2 >> 32) != 0ull));
        const u64 pr = __builtin_bitreverse64(cr_ballot(word == 0xFFFFFFFFu));
        const u64 aa = g | pr;
        const u64 s1 = aa + g, s2 = s1 + bit_in;
        const u64 carries = __builtin_bitreverse64(s2 ^ aa ^ g);         /* bit j: a carry enters lane j from its right */
        const uint32_t done = word + (uint32_t)((carries >> lane) & 1ull);
        if (k < wtotal) *reinterpret_cast<cr_u32u*>(wout + (u64)k * 4u) = __builtin_bswap32(done);
        bit_in = ((s1 < aa) || (s2 < s1)) ? 1u : 0u;
        c_in = cr_lane_get(up, 0);
        if (k0 == 0u) break;
    }
    if (maybe_stored) {
        cr_wave_sync();
        uint32_t bt = 0x55u;                                             /* lane l: stream byte S - l */
        if (lane <= sbase) bt = wout[sbase - lane];
        const u64 ff = cr_ballot(bt == 0xffu), zz = cr_ballot(bt == 0u);
        const uint32_t run_ff = (uint32_t)__builtin_ctzll(~ff | (1ull << 63)), run_zz = (uint32_t)__builtin_ctzll(~zz | (1ull << 63));
        const uint32_t g = run_ff > run_zz ? run_ff : run_zz;            /* (63 = "63 or more": then nothing is certain) */
        if (g < 63u && g <= sbase && header + sbase - g >= n) return 0xFFFFFFFFu;
        return 0u;
    }
    return stotal;
}

CR_DEV uint32_t cr_rop_code_events_fast(const uint8_t* src, uint32_t n, uint8_t* dst, CrEvViews& V, u64* ring /* LDS [CR_RC_RING] */) {
    const uint32_t got = cr_code_events_fast(n, dst + CR_ROP_HEADER, CR_ROP_HEADER, V, ring);
    if (got == 0u) return 0u;
    if (got == 0xFFFFFFFFu) { cr_wave_sync(); cr_rop_store_raw(src, n, dst); return CR_ROP_HEADER + n; }
    cr_rop_write_header(src, n, cr_uni(V.ctr[3]) & 0xffu, dst);
    return CR_ROP_HEADER + got;
}

#endif

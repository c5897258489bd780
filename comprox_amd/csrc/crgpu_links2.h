/*
 * comprox_amd/csrc/crgpu_links2.h — the event sorts of the chain encoder in LDS (kernel k_rop_links_lds, 8 waves per
 * datablock, blocks of up to 65 536 events).
 *
 * Reference: what is being grouped are the model look-ups of /root/reference/src/cr-ppm.c:103-167 — the order-2 node
 * `o2_models[context & 0xffff]` and the order-3 entry of key cr-ppm.c:66 — per event, in coding order (crgpu_rop2.h
 * explains why every component of ppm_encode but the range coder is a per-key recurrence).
 *
 * Same views as cr_rop_sort_events (crgpu_rop2.h: list2 / csym2 / slot2 / chains2, csym3 / cslot3 / starts3). k_rop_links sorts
 * {key, event} pairs of 8 bytes through global memory, five passes of 8 bits, tile by tile through a 32 KB LDS buffer. Blocks of
 * up to 28 672 events (rounds 2 and 3): u16 event numbers ping-pong between two LDS buffers and a pass reads ONE BYTE of every
 * event's key, laid out per event in LDS from a coalesced read of the contexts in front of the pass (cr_rop_sort_events_lds).
 * Blocks of up to 65 536 events (round 4): records that carry their key, sorted in groups by key (cr_rop_sort_events_lk4).
 */
#ifndef CRGPU_LINKS2_H
#define CRGPU_LINKS2_H

#include "crgpu_lzp2.h"

struct CrLinks2Shared { uint32_t n2, n3, front, back; };

struct CrEvDigitKey {                         /* "key" of a record for cr_lz2_pass: the digit laid out for this pass */
    const uint8_t* d;
    CR_DEV uint32_t operator()(uint32_t i) const { return d[i]; }
};


/* digit `which` of every event's key into S.src: 0 / 1 = order-2 context bits 0-7 / 8-15, 2 / 3 / 4 = order-3 key bits
 * 0-7 / 8-15 / 16-21 */
CR_DEV void cr_links2_digits(const CrLz2Shared& S, const uint32_t* ev_ctx, uint32_t nev, int which) {
    /* four events per 16-byte load, two loads in flight per thread: the sweep is bound by the latency of global memory (the
     * contexts are 16-byte aligned and padded to the slot's capacity, S.src to 32 bytes behind CR_LZ2_MAXN) */
    const uint32_t sh = which < 2 ? (uint32_t)which * 8u : (uint32_t)(which - 2) * 8u;
    /* (round 5, last: four loads in flight, all unconditional — a group past the end reads the thread's first group again and
     * stores nothing: six round trips per sweep become three) */
    const uint32_t stride = blockDim.x * 4u;
    for (uint32_t i = threadIdx.x * 4u; i < nev; i += stride * 4u) {
        const uint32_t j1 = i + stride, j2 = i + 2u * stride, j3 = i + 3u * stride;
        const uint4 c0 = *reinterpret_cast<const uint4*>(ev_ctx + i);
        const uint4 c1 = *reinterpret_cast<const uint4*>(ev_ctx + (j1 < nev ? j1 : i));
        const uint4 c2 = *reinterpret_cast<const uint4*>(ev_ctx + (j2 < nev ? j2 : i));
        const uint4 c3 = *reinterpret_cast<const uint4*>(ev_ctx + (j3 < nev ? j3 : i));
#define CR_DG(c_) ((which < 2 ? ((c_) & 0xffffu) : cr_o3_key(c_)) >> sh & 0xffu)
#define CR_DG4(c_) (CR_DG(c_.x) | CR_DG(c_.y) << 8 | CR_DG(c_.z) << 16 | CR_DG(c_.w) << 24)
        *reinterpret_cast<uint32_t*>(S.src + i) = CR_DG4(c0);
        if (j1 < nev) *reinterpret_cast<uint32_t*>(S.src + j1) = CR_DG4(c1);
        if (j2 < nev) *reinterpret_cast<uint32_t*>(S.src + j2) = CR_DG4(c2);
        if (j3 < nev) *reinterpret_cast<uint32_t*>(S.src + j3) = CR_DG4(c3);
#undef CR_DG4
#undef CR_DG
    }
    __syncthreads();
}

/* every thread of the workgroup (CR_LZ2_THREADS); 0 < nev <= CR_LZ2_MAXN */
CR_DEV void cr_rop_sort_events_lds(const CrLz2Shared& S, CrLinks2Shared& sh, CrEvViews& V, uint32_t* last2 /* u32[65536], global */, uint32_t nev) {
    const uint32_t t = threadIdx.x;
    if (t == 0) { sh.n2 = 0; sh.n3 = 0; sh.front = 0; sh.back = 0; }
    CrEvDigitKey key; key.d = S.src;
    /* ---- order-2 context, 16 bits: a <- identity by digit 0, b <- a by digit 1 */
    cr_links2_digits(S, V.ev_ctx, nev, 0);
    cr_lz2_pass(S, key, 0u, nev, 0u, nullptr, S.a);
    cr_links2_digits(S, V.ev_ctx, nev, 1);
    cr_lz2_pass(S, key, 0u, nev, 0u, S.a, S.b);
    {
        /* the low byte of every event's key into the free buffer: key of event i = src[i] << 8 | lo[i] */
        uint8_t* lo = reinterpret_cast<uint8_t*>(S.a);
        for (uint32_t i = t; i < nev; i += blockDim.x) lo[i] = (uint8_t)V.ev_ctx[i];
        __syncthreads();
        const uint16_t* L = S.b;
        for (uint32_t s = t; s < nev; s += blockDim.x) {
            const uint32_t i = L[s];
            const uint32_t k = ((uint32_t)S.src[i] << 8) | lo[i];
            uint32_t kn = 0xFFFFFFFFu, kp = 0xFFFFFFFFu;
            if (s + 1u < nev) { const uint32_t j = L[s + 1u]; kn = ((uint32_t)S.src[j] << 8) | lo[j]; }
            if (s) { const uint32_t j = L[s - 1u]; kp = ((uint32_t)S.src[j] << 8) | lo[j]; }
            const uint32_t sy = V.ev_sym[i];
            const bool last = kn != k, first = kp != k;
            V.list2[s] = i;
            V.csym2[s] = (uint16_t)((sy & 0x1ffu) | (last ? 0x8000u : 0u));
            V.cpred[s] = 0;                       /* (k_rop_o3's range walker only stores the predictions that are not 0) */
            V.slot2[i] = s;
            if (last) last2[k] = s + 1u;
            if (first) V.starts2[atomicAdd(&sh.n2, 1u)] = s;
        }
        cr_wg_sync_global();
        /* chains of 96 events and more first: a lane that meets one late would finish long after the others */
        const uint32_t n2 = sh.n2;
        for (uint32_t c = t; c < n2; c += blockDim.x) {
            const uint32_t s0 = V.starts2[c];
            const uint32_t i0 = L[s0];
            const uint32_t e = last2[((uint32_t)S.src[i0] << 8) | lo[i0]];
            const uint32_t at = (e - s0 >= 96u) ? atomicAdd(&sh.front, 1u) : n2 - 1u - atomicAdd(&sh.back, 1u);
            V.chains2[at] = (u64)s0 | ((u64)e << 32);
        }
        __syncthreads();
    }
    /* ---- order-3 key, 22 bits: a <- identity by digit 0, b <- a by digit 1, a <- b by digit 2 */
    cr_links2_digits(S, V.ev_ctx, nev, 2);
    cr_lz2_pass(S, key, 0u, nev, 0u, nullptr, S.a);
    cr_links2_digits(S, V.ev_ctx, nev, 3);
    cr_lz2_pass(S, key, 0u, nev, 0u, S.a, S.b);
    cr_links2_digits(S, V.ev_ctx, nev, 4);
    cr_lz2_pass(S, key, 0u, nev, 0u, S.b, S.a);
    {
        uint16_t* lo = S.b;                                    /* key of event i = src[i] << 16 | lo[i] */
        for (uint32_t i = t; i < nev; i += blockDim.x) lo[i] = (uint16_t)cr_o3_key(V.ev_ctx[i]);
        __syncthreads();
        const uint16_t* L = S.a;
        for (uint32_t s = t; s < nev; s += blockDim.x) {
            const uint32_t i = L[s];
            const uint32_t k = ((uint32_t)S.src[i] << 16) | lo[i];
            uint32_t kn = 0xFFFFFFFFu, kp = 0xFFFFFFFFu;
            if (s + 1u < nev) { const uint32_t j = L[s + 1u]; kn = ((uint32_t)S.src[j] << 16) | lo[j]; }
            if (s) { const uint32_t j = L[s - 1u]; kp = ((uint32_t)S.src[j] << 16) | lo[j]; }
            const uint32_t sy = V.ev_sym[i], sl = V.slot2[i];
            const bool last = kn != k, first = kp != k;
            V.csym3[s] = (uint16_t)((sy & 0x1ffu) | (last ? 0x8000u : 0u));
            V.cslot3[s] = sl;
            if (first) V.starts3[atomicAdd(&sh.n3, 1u)] = s;
        }
    }
    __syncthreads();
    if (t == 0) { V.ctr[1] = sh.n2; V.ctr[2] = sh.n3; }
}


/* ==== round 4: up to 65 536 events, records that carry their key ===============================================================
 * Rounds 2 and 3 sorted u16 event numbers and read ONE BYTE of an event's key per pass out of a plane that had to be laid out
 * again — from global memory, a latency-bound sweep — in front of every pass; the passes gathered that byte twice per record, and
 * the chain boundaries took the rest of the key, the symbol and the order-2 slot from more planes and global gathers. Here a
 * record is 64 bits: the key in bits 42-63 and everything the views need from the event in bits 0-41 (order-2 sort: event number
 * and symbol; order-3 sort: the event's order-2 slot and symbol). A pass reads its records in order, takes the digit out of the
 * record, and counts the next digit for the wave that will read the record next (the fused passes of crgpu_lzp2.h): no plane,
 * no gather, and global memory is only read in coding order (coalesced, four chunks in flight). Two record buffers of 70 KB hold
 * 8 960 records, so the events are sorted in GROUPS BY KEY like the positions of k_rop_lzp_lds64 (cr_lz3_groups: 256 bins by a
 * digit mixed from the whole key, consecutive bins packed into groups): a group's events are compacted in coding order, sorted,
 * and handed to the view writer. Chains never leave a group, and the order of the chains among themselves does not matter to
 * the chain kernels, so a group's slots simply follow the previous group's. Event counts from 1 to 65 536; a block whose keys do
 * not split (one context for a seventh of its events) goes to k_rop_links. */
#ifndef CR_LK4_THREADS
#define CR_LK4_THREADS   512u            /* 8 waves (16: 5.0 -> 6.1 ms on 64 KiB text: more barriers and counters than the sweeps gain) */
#endif
#define CR_LK4_BUF_BYTES (CR_LK4_THREADS == 1024u ? 67584u : 71680u)          /* (152 576 bytes of dynamic LDS in all: see CR_LZ3_CAP) */
#define CR_LK4_CAP       (CR_LK4_BUF_BYTES / 8u)
#define CR_LK4_LDS_BYTES (2u * CR_LK4_BUF_BYTES + (CR_LK4_THREADS / 64u) * 256u * 4u + 256u * 4u)
#define CR_LK4_KEY_SHIFT 42u

CR_DEV CrLz2Shared cr_lk4_carve(uint8_t* lds, uint32_t waves) {
    CrLz2Shared S;
    S.a = reinterpret_cast<uint16_t*>(lds);
    S.b = reinterpret_cast<uint16_t*>(lds + CR_LK4_BUF_BYTES);
    S.hist = reinterpret_cast<uint32_t*>(lds + 2u * CR_LK4_BUF_BYTES);
    S.base = S.hist + waves * 256u;
    S.src = nullptr;
    return S;
}

CR_DEV uint32_t cr_lk4_key(u64 r) { return (uint32_t)(r >> CR_LK4_KEY_SHIFT); }

/* one placing pass over records that carry their key (cr_lz2_place without the gather) */
template <int NB>
CR_DEV void cr_lk4_place(uint32_t count, const CrLz2Plan& P, uint32_t shift, uint32_t nmask, const u64* src, u64* dst, uint16_t* cur, uint16_t* nxt) {
    const uint32_t lane = cr_lane(), w = cr_wave_id();
    const uint32_t lo = w * P.per < count ? w * P.per : count;
    const uint32_t hi = lo + P.per < count ? lo + P.per : count;
    uint16_t* const my = cur + w * 256u;
    u64 r_n = 0;
    if (lo + lane < hi) r_n = src[lo + lane];
    for (uint32_t i0 = lo; i0 < hi; i0 += CRGPU_WAVE) {
        const uint32_t i = i0 + lane;
        const bool act = i < hi;
        const u64 r = r_n;
        if (i + CRGPU_WAVE < hi) r_n = src[i + CRGPU_WAVE];
        const uint32_t k = cr_lk4_key(r);
        const uint32_t dg = (k >> shift) & ((1u << NB) - 1u);
        const u64 same = cr_same_key_mask<NB>(dg, act);
        const u64 lower = same & ((1ull << lane) - 1ull);
        uint32_t at = 0;
        if (act) {
            at = my[dg];
            const uint32_t slot = at + (uint32_t)__builtin_popcountll(lower);
            dst[slot] = r;
            if (nmask) cr_h16_add(nxt, __umulhi(slot, P.magic) * 256u + ((k >> (shift + 8u)) & nmask));
        }
        cr_lds_order_sw();
        if (act && (same >> lane) >> 1 == 0ull) my[dg] = (uint16_t)(at + (uint32_t)__builtin_popcountll(same));   /* the group's last lane */
        cr_lds_order_sw();
    }
    __syncthreads();
}

#ifdef CR_LK4_PROF     /* diagnostic build: per block, the 100 MHz ticks spent in each phase, added up over the groups (tools/links_profile.py) */
#define CR_LK4_MARK(st_, slot_) do { if ((st_) && threadIdx.x == 0) { const u64 now_ = wall_clock64(); (st_)[slot_] += now_ - (st_)[15]; (st_)[15] = now_; } } while (0)
#else
#define CR_LK4_MARK(st_, slot_) do { } while (0)
#endif
/* per-wave bins of two keys at once (the sweep over the events is bound by the latency of global memory, so the order-3 sort's
 * bins are counted during the order-2 sort's sweep): counter array 0 / 1 <- bins of key2(i) / key3(i), eight chunks in flight */
template <class K2, class K3>
CR_DEV void cr_lk4_bins(const CrLz2Shared& S, const K2& key2, const K3& key3, uint32_t nev) {
    const uint32_t lane = cr_lane(), w = cr_wave_id();
    const CrLz2Plan PA = cr_lz2_plan(nev);
    const uint32_t lo = w * PA.per < nev ? w * PA.per : nev;
    const uint32_t hi = lo + PA.per < nev ? lo + PA.per : nev;
    uint16_t* const b2 = cr_lz2_hist16(S, 0);
    uint16_t* const b3 = cr_lz2_hist16(S, 1);
    for (uint32_t k = lane; k < 128u; k += CRGPU_WAVE) { reinterpret_cast<uint32_t*>(b2 + w * 256u)[k] = 0u; reinterpret_cast<uint32_t*>(b3 + w * 256u)[k] = 0u; }
    cr_lds_order_sw();
    for (uint32_t i0 = lo; i0 < hi; i0 += 8u * CRGPU_WAVE) {
        uint32_t k2[8], k3[8];
#pragma unroll
        for (uint32_t u = 0; u < 8u; u++) { const uint32_t i = i0 + u * CRGPU_WAVE + lane, ic = i < hi ? i : hi - 1u; k2[u] = key2(ic); k3[u] = key3(ic); }   /* (clamped, not `i < hi ? load : 0`: behind a branch every one of the eight loads is waited for on the spot) */
#pragma unroll
        for (uint32_t u = 0; u < 8u; u++) if (i0 + u * CRGPU_WAVE + lane < hi) { cr_h16_add(b2, w * 256u + cr_lz3_bin(k2[u])); cr_h16_add(b3, w * 256u + cr_lz3_bin(k3[u])); }
    }
    __syncthreads();
}

/* The events 0 .. nev - 1 grouped by key (stable), G = the groups cut from the key's bins (cr_lk4_bins, cr_lz3_cut). rec(i) =
 * the event's record (key in bits 42-63), read in coding order, ONE sweep: every record goes to its group's run in `scratch`
 * (global, u64[nev]; a wave's records of a group lie together, in order), then group after group is loaded into LDS, sorted,
 * and handed to fn(gbase, m, sorted): called by every thread, the records are valid until fn returns. */
template <class RecFn, class Fn>
CR_DEV void cr_lk4_sort(const CrLz2Shared& S, const CrLz3Groups& G, const RecFn& rec, uint32_t nev, uint32_t bits, u64* scratch, const Fn& fn, u64* st = nullptr, int st0 = 0) {
    const uint32_t lane = cr_lane(), w = cr_wave_id(), nw = blockDim.x >> 6;
    const CrLz2Plan PA = cr_lz2_plan(nev);
    const uint32_t lo = w * PA.per < nev ? w * PA.per : nev;
    const uint32_t hi = lo + PA.per < nev ? lo + PA.per : nev;
    const uint32_t ng = G.ngroups;
    {   /* distribute: lane g of `off` = where this wave's next record of group g goes */
        uint32_t off = 0, gs = 0;
        for (uint32_t g = 0; g < ng; g++) { if (lane == g) off = gs + G.woff[w][g]; gs += G.gsize[g]; }
        for (uint32_t i0 = lo; i0 < hi; i0 += 4u * CRGPU_WAVE) {
            u64 r[4];
#pragma unroll
            for (uint32_t u = 0; u < 4u; u++) { const uint32_t i = i0 + u * CRGPU_WAVE + lane; r[u] = rec(i < hi ? i : hi - 1u); }
#pragma unroll
            for (uint32_t u = 0; u < 4u; u++) {
                const bool act = i0 + u * CRGPU_WAVE + lane < hi;
                const uint32_t mg = act ? (uint32_t)G.binmap[cr_lz3_bin(cr_lk4_key(r[u]))] : 0xffu;
                uint32_t dest = 0;
                for (uint32_t g = 0; g < ng; g++) {
                    const u64 am = cr_ballot(mg == g);
                    if (am == 0ull) continue;
                    const uint32_t base = cr_lane_get(off, g);
                    if (mg == g) dest = base + (uint32_t)__builtin_popcountll(am & ((1ull << lane) - 1ull));
                    if (lane == g) off += (uint32_t)__builtin_popcountll(am);
                }
                if (act) scratch[dest] = r[u];
            }
        }
    }
    cr_wg_sync_global();
    CR_LK4_MARK(st, st0 + 1);
    u64* const A = reinterpret_cast<u64*>(S.a);
    u64* const Bf = reinterpret_cast<u64*>(S.b);
    uint32_t gbase = 0;
    for (uint32_t g = 0; g < ng; g++) {
        const uint32_t m = G.gsize[g];
        if (m == 0u) continue;
        const CrLz2Plan P = cr_lz2_plan(m);
        uint16_t* cur = cr_lz2_hist16(S, 0);
        uint16_t* nxt = cr_lz2_hist16(S, 1);
        {   /* the group's run -> A, every wave the range it will read in the first pass, counting that pass's digit as it goes */
            const uint32_t mlo = w * P.per < m ? w * P.per : m;
            const uint32_t mhi = mlo + P.per < m ? mlo + P.per : m;
            for (uint32_t k = lane; k < 128u; k += CRGPU_WAVE) reinterpret_cast<uint32_t*>(cur + w * 256u)[k] = 0u;
            cr_lds_order_sw();
            const u64* const run = scratch + gbase;
            for (uint32_t j0 = mlo; j0 < mhi; j0 += 8u * CRGPU_WAVE) {
                u64 r[8];
#pragma unroll
                for (uint32_t u = 0; u < 8u; u++) { const uint32_t j = j0 + u * CRGPU_WAVE + lane; r[u] = run[j < mhi ? j : mhi - 1u]; }
#pragma unroll
                for (uint32_t u = 0; u < 8u; u++) {
                    const uint32_t j = j0 + u * CRGPU_WAVE + lane;
                    if (j < mhi) { A[j] = r[u]; cr_h16_add(cur, w * 256u + (cr_lk4_key(r[u]) & 255u)); }
                }
            }
        }
        __syncthreads();
        CR_LK4_MARK(st, st0 + 1);
        const u64* src = A;
        u64* dst = Bf;
        for (uint32_t shift = 0; shift < bits; shift += 8u) {
            const uint32_t left = bits - shift;
            const uint32_t nleft = left > 8u ? left - 8u : 0u;
            const uint32_t nmask = nleft == 0u ? 0u : nleft >= 8u ? 255u : (1u << nleft) - 1u;
            cr_lz2_scan16(S, cur, nmask ? nxt : nullptr);
            if (left > 4u) cr_lk4_place<8>(m, P, shift, nmask, src, dst, cur, nxt);
            else cr_lk4_place<4>(m, P, shift, nmask, src, dst, cur, nxt);
            const u64* t = src; src = dst; dst = const_cast<u64*>(t);
            uint16_t* h = cur; cur = nxt; nxt = h;
        }
        CR_LK4_MARK(st, st0 + 2);
        fn(gbase, m, src);
        CR_LK4_MARK(st, st0 + 3);
        gbase += m;
    }
}

/* the sorted records L[0 .. m) of a group -> fn(local index, record, key, first of its chain, last of its chain); a wave walks
 * its own range, the neighbours' keys come from the lanes next door */
template <class Fn>
CR_DEV void cr_lk4_walk(const u64* L, uint32_t m, const Fn& fn) {
    const uint32_t lane = cr_lane(), w = cr_wave_id();
    const CrLz2Plan P = cr_lz2_plan(m);
    const uint32_t lo = w * P.per < m ? w * P.per : m;
    const uint32_t hi = lo + P.per < m ? lo + P.per : m;
    uint32_t carry_k = 0xFFFFFFFFu;
    if (lo > 0u && lo < hi) carry_k = cr_lk4_key(L[lo - 1u]);
    for (uint32_t i0 = lo; i0 < hi; i0 += CRGPU_WAVE) {
        const uint32_t i = i0 + lane;
        const u64 r = i < m ? L[i] : 0ull;                        /* (lanes behind the wave's range still read their record for its key) */
        const uint32_t k = i < m ? cr_lk4_key(r) : 0xFFFFFFFFu;
        uint32_t kb = 0xFFFFFFFFu;                                /* the key behind this chunk */
        if (i0 + CRGPU_WAVE < m) kb = cr_lk4_key(L[i0 + CRGPU_WAVE]);
        const uint32_t kp = cr_shift_up1(k, carry_k);
        const uint32_t kn = cr_shift_down1(k, kb);
        carry_k = cr_lane_get(k, 63);
        if (i < hi) fn(i, r, k, i == 0u || kp != k, i + 1u == m || kn != k);
    }
    __syncthreads();
}

/* every thread of the workgroup (CR_LZ2_THREADS); 0 < nev <= 65 536; S from cr_lk4_carve. Same views as cr_rop_sort_events
 * (crgpu_rop2.h). Returns false when the keys do not split into groups (the block then goes to k_rop_links). */
CR_DEV bool cr_rop_sort_events_lk4(const CrLz2Shared& S, CrLz3Groups& G, CrLz3Groups& G3, CrLinks2Shared& sh, CrEvViews& V, uint32_t* last2 /* u32[65536], global */, uint32_t nev, u64* st = nullptr) {
    const uint32_t t = threadIdx.x;
    if (t == 0) { sh.n2 = 0; sh.n3 = 0; sh.front = 0; sh.back = 0; }
#ifdef CR_LK4_PROF
    if (st && t == 0) { for (int q = 0; q < 15; q++) st[q] = 0; st[15] = wall_clock64(); st[12] = nev; }
#endif
    __syncthreads();
    const uint32_t* const ev_ctx = V.ev_ctx;
    const uint16_t* const ev_sym = V.ev_sym;
    /* the bins of both keys in one sweep, then both cuts (the passes below use the counter arrays the bins sit in) */
    cr_lk4_bins(S, [ev_ctx](uint32_t i) { return ev_ctx[i] & 0xffffu; }, [ev_ctx](uint32_t i) { return cr_o3_key(ev_ctx[i]); }, nev);
    cr_lz3_cut(S, G, cr_lz2_hist16(S, 0), CR_LK4_CAP);
    cr_lz3_cut(S, G3, cr_lz2_hist16(S, 1), CR_LK4_CAP);
    if (G.ngroups == 0u || G3.ngroups == 0u) return false;
    CR_LK4_MARK(st, 0);
    /* ---- order-2 context, 16 bits; payload: event number (17 bits) | symbol << 17 */
    cr_lk4_sort(S, G, [ev_ctx, ev_sym](uint32_t i) { return ((u64)(ev_ctx[i] & 0xffffu) << CR_LK4_KEY_SHIFT) | ((u64)(ev_sym[i] & 0x1ffu) << 17) | i; },
                nev, 16u, V.sortA, [&V, &sh, last2](uint32_t gbase, uint32_t m, const u64* L) {
            cr_lk4_walk(L, m, [&V, &sh, last2, gbase](uint32_t li, u64 r, uint32_t k, bool first, bool last) {
                const uint32_t s = gbase + li, i = (uint32_t)r & 0x1ffffu, sy = (uint32_t)(r >> 17) & 0x1ffu;
                V.list2[s] = i;
                V.csym2[s] = (uint16_t)(sy | (last ? 0x8000u : 0u));
                V.cpred[s] = 0;                       /* (k_rop_o3's range walker only stores the predictions that are not 0) */
                V.slot2[i] = s;
                if (last) last2[k] = s + 1u;
                if (first) V.starts2[atomicAdd(&sh.n2, 1u)] = s | (k << 16);          /* (slots < 65 536) */
            });
        }, st, 0);
    cr_wg_sync_global();
    {   /* chains of 96 events and more first: a lane that meets one late would finish long after the others */
        const uint32_t n2 = sh.n2;
        for (uint32_t c = t; c < n2; c += blockDim.x) {
            const uint32_t sk = V.starts2[c], s0 = sk & 0xffffu;
            const uint32_t e = last2[sk >> 16];
            const uint32_t at = (e - s0 >= 96u) ? atomicAdd(&sh.front, 1u) : n2 - 1u - atomicAdd(&sh.back, 1u);
            V.chains2[at] = (u64)s0 | ((u64)e << 32);
        }
        __syncthreads();
    }
    CR_LK4_MARK(st, 4);
    /* ---- order-3 key, 22 bits; payload: the event's order-2 slot (17 bits) | symbol << 17 */
    const uint32_t* const slot2 = V.slot2;
    cr_lk4_sort(S, G3, [ev_ctx, ev_sym, slot2](uint32_t i) { return ((u64)cr_o3_key(ev_ctx[i]) << CR_LK4_KEY_SHIFT) | ((u64)(ev_sym[i] & 0x1ffu) << 17) | slot2[i]; },
                nev, 22u, V.sortA, [&V, &sh](uint32_t gbase, uint32_t m, const u64* L) {
            cr_lk4_walk(L, m, [&V, &sh, gbase](uint32_t li, u64 r, uint32_t k, bool first, bool last) {
                (void)k;
                const uint32_t s = gbase + li;
                V.csym3[s] = (uint16_t)(((uint32_t)(r >> 17) & 0x1ffu) | (last ? 0x8000u : 0u));
                V.cslot3[s] = (uint32_t)r & 0x1ffffu;
                if (first) V.starts3[atomicAdd(&sh.n3, 1u)] = s;
            });
        }, st, 5);
    __syncthreads();
    if (t == 0) { V.ctr[1] = sh.n2; V.ctr[2] = sh.n3; }
    return true;
}

#endif

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
 *   k_rop_links   per key, the next event of the same key + the chain heads   (2 waves / block)
 *   k_rop_o3      one LANE per order-3 chain: predicted byte of every event
 *   k_rop_o2      one LANE per order-2 chain: the node lives in that lane's registers/private
 *                 memory; emits (cum, frq, tot) per event and, for escapes, the exclusion set
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

/* per-block scratch (device memory owned by the context, one slot per block of the batch) */
struct CrEvViews {
    uint32_t* ctr;       /* [0] #events, [1] #order-2 heads, [2] #order-3 heads, [3] esc | stored<<8 */
    uint32_t* ev_ctx;    /* u32[cap] */
    uint32_t* cid2;      /* u32[cap]: order-2 chain number of the event */
    uint32_t* slot2;     /* u32[cap]: during the sweep the event's rank in its chain, then its slot in list2 */
    uint32_t* off2;      /* u32[cap+1]: chain c owns list2[off2[c] .. off2[c+1]) (lengths before the scan) */
    uint32_t* list2;     /* u32[cap]: event numbers, chain after chain, in coding order */
    uint32_t* next3;     /* u32[cap] */
    uint32_t* head3;     /* u32[cap] */
    uint16_t* csym;      /* u16[cap]: symbols in list2 order */
    uint8_t*  cpred;     /* u8[cap]: predicted bytes in list2 order (written by the order-3 pass) */
    u64*      trip;      /* u64[cap]: cum | tot << 20 | frq << 40 | type << 50 */
    uint32_t* mask;      /* u32[cap][8]: bit s set = byte s has a count in the node (escape events only) */
    uint16_t* ev_sym;    /* u16[cap] */
    uint8_t*  ev_pred;   /* u8[cap] */
    uint32_t  cap;
};

CR_DEV u64 cr_ev_slot_bytes(uint32_t cap) { return 64ull + (u64)cap * (4u * 7u + 8u + 32u + 2u + 2u + 1u + 1u) + 512u; }

CR_DEV CrEvViews cr_ev_views(uint8_t* base, uint32_t cap) {
    CrEvViews V;
    V.cap = cap;
    V.ctr = reinterpret_cast<uint32_t*>(base);
    uint8_t* p = base + 64;
    V.trip = reinterpret_cast<u64*>(p);            p += (u64)cap * 8u;
    V.mask = reinterpret_cast<uint32_t*>(p);       p += (u64)cap * 32u;
    V.ev_ctx = reinterpret_cast<uint32_t*>(p);     p += (u64)cap * 4u;
    V.cid2 = reinterpret_cast<uint32_t*>(p);       p += (u64)cap * 4u;
    V.slot2 = reinterpret_cast<uint32_t*>(p);      p += (u64)cap * 4u;
    V.off2 = reinterpret_cast<uint32_t*>(p);       p += (u64)cap * 4u + 64u;
    V.list2 = reinterpret_cast<uint32_t*>(p);      p += (u64)cap * 4u;
    V.next3 = reinterpret_cast<uint32_t*>(p);      p += (u64)cap * 4u;
    V.head3 = reinterpret_cast<uint32_t*>(p);      p += (u64)cap * 4u;
    V.ev_sym = reinterpret_cast<uint16_t*>(p);     p += (u64)cap * 2u;
    V.csym = reinterpret_cast<uint16_t*>(p);       p += (u64)cap * 2u;
    V.ev_pred = p;                                 p += (u64)cap;
    V.cpred = p;
    return V;
}

/* ------------------------------------------------------------------ k_rop_events */

/* token loop of lzencode (cr-coder.c:169-207) without the coding: one wave, events staged 64 at a time */
CR_DEV void cr_rop_emit_events(const uint8_t* src, uint32_t n, const uint8_t* lens, CrEvViews& V, CrShared& sh) {
    const uint32_t lane = cr_lane();
    if (n < 16u) {                                            /* cr-coder.c:140-142 */
        if (lane == 0) { V.ctr[0] = 0; V.ctr[1] = 0; V.ctr[2] = 0; V.ctr[3] = 0x100u; }
        return;
    }
    const uint32_t esc = cr_pick_escape(src, n, sh.hist);
    CrWindow win, lwin;
    cr_window_init(win, src, n, CR_LZP_SKIP);
    cr_window_init(lwin, lens, n, CR_LZP_SKIP);
    uint32_t pos = CR_LZP_SKIP, ctx = 0, nev = 0, held = 0;
    uint32_t my_ctx = 0, my_sym = 0;
#define CR_EMIT(c_, s_) do { if (lane == held) { my_ctx = (c_); my_sym = (s_); } held++; \
        if (held == CRGPU_WAVE) { V.ev_ctx[nev + lane] = my_ctx; V.ev_sym[nev + lane] = (uint16_t)my_sym; nev += CRGPU_WAVE; held = 0; } } while (0)
    while (pos < n) {
        uint32_t len = 1;
        if (pos + CR_LZP_TAIL < n) len = cr_window_at(lwin, pos);
        if (len > 1u) {                                                  /* esc, then the length in the context ending in esc */
            CR_EMIT(ctx, esc);
            CR_EMIT((ctx << 8) | esc, len | CR_EV_LAST);
            /* last four bytes of the match, from the byte window (at most one refill, which the
             * literals that follow need anyway) instead of a dependent load per match */
            const uint32_t e4 = pos + len - 4u;
            ctx = (cr_window_at(win, e4) << 24) | (cr_window_at(win, e4 + 1u) << 16) | (cr_window_at(win, e4 + 2u) << 8) | cr_window_at(win, e4 + 3u);
        } else {
            const uint32_t c = cr_window_at(win, pos);
            if (c == esc) {
                CR_EMIT(ctx, esc);
                CR_EMIT((ctx << 8) | esc, 0u | CR_EV_LAST);
                ctx = (ctx << 16) | (esc << 8) | esc;
            } else {
                CR_EMIT(ctx, c | CR_EV_LAST);
                ctx = (ctx << 8) | c;
            }
        }
        pos += len;
    }
    if (lane < held) { V.ev_ctx[nev + lane] = my_ctx; V.ev_sym[nev + lane] = (uint16_t)my_sym; }
    nev += held;
#undef CR_EMIT
    if (lane == 0) { V.ctr[0] = nev; V.ctr[1] = 0; V.ctr[2] = 0; V.ctr[3] = esc; }
}

/* ------------------------------------------------------------------ k_rop_links */

/* one wave, order-3 key (22 bits, hashed table): for every event the next event with the same key,
 * and the list of first events (chains are short here: they are walked through the links) */
CR_DEV void cr_rop_link_o3(const CrLzp& z, CrEvViews& V, uint32_t nev) {
    const uint32_t lane = cr_lane();
    uint32_t nheads = 0;
    for (uint32_t e0 = 0; e0 < nev; e0 += CRGPU_WAVE) {
        const uint32_t i = e0 + lane;
        const bool act = i < nev;
        uint32_t key = 0;
        if (act) key = cr_o3_key(V.ev_ctx[i]);
        int q = cr_prev_same_bits<22>(key, act);
        uint32_t prev = 0xFFFFFFFFu;
        if (act) {
            if (q >= 0) prev = e0 + (uint32_t)q;
            else prev = cr_htab_get(z, z.t8, key, 0xFFFFFFFFu);
            if (prev != 0xFFFFFFFFu) V.next3[prev] = i;
        }
        const u64 hm = cr_ballot(act && prev == 0xFFFFFFFFu);
        if (act && prev == 0xFFFFFFFFu) V.head3[nheads + (uint32_t)__builtin_popcountll(hm & ((1ull << lane) - 1ull))] = i;
        nheads += (uint32_t)__builtin_popcountll(hm);
        cr_wave_sync();
        if (act) cr_htab_learn(z, z.t8, key, i);
        cr_wave_sync();
    }
    if (lane == 0) V.ctr[2] = nheads;
}

/* one wave, order-2 key (dense table of 65 536 u32 = last event + 1): chain number and rank of every
 * event, chain lengths; then the exclusive scan of the lengths. The long order-2 chains are laid out
 * contiguously afterwards (cr_rop_scatter_o2) so that walking one is a sequential read, not a pointer chase. */
CR_DEV void cr_rop_number_o2(const CrLzp& z, CrEvViews& V, uint32_t nev) {
    const uint32_t lane = cr_lane();
    uint32_t nheads = 0;
    for (uint32_t e0 = 0; e0 < nev; e0 += CRGPU_WAVE) {
        const uint32_t i = e0 + lane;
        const bool act = i < nev;
        uint32_t key = 0;
        if (act) key = V.ev_ctx[i] & 0xffffu;
        const u64 same = cr_same_key_mask<16>(key, act);
        const u64 lower = same & ((1ull << lane) - 1ull);
        const bool first = act && lower == 0ull;                   /* first event of its key in this step */
        uint32_t cid = 0, rank = 0;
        bool fresh = false;
        if (first) {
            const uint32_t v = cr_ld32(z.t2 + key);
            if (v) { cid = V.cid2[v - 1u]; rank = V.slot2[v - 1u] + 1u; }
            else fresh = true;
        }
        const u64 fm = cr_ballot(fresh);
        if (fresh) cid = nheads + (uint32_t)__builtin_popcountll(fm & ((1ull << lane) - 1ull));
        nheads += (uint32_t)__builtin_popcountll(fm);
        /* the other events of the key take chain and rank from that first lane */
        const uint32_t src_lane = act ? (uint32_t)__builtin_ctzll(same) : lane;
        cid = (uint32_t)__shfl((int)cid, (int)src_lane);
        rank = (uint32_t)__shfl((int)rank, (int)src_lane) + (uint32_t)__builtin_popcountll(lower);
        if (act) {
            V.cid2[i] = cid;
            V.slot2[i] = rank;
            if ((same >> lane) >> 1 == 0ull) V.off2[cid] = rank + 1u;   /* last of its key here: chain length so far */
        }
        cr_wave_sync();
        if (act) atomicMax(z.t2 + key, i + 1u);
        cr_wave_sync();
    }
    /* exclusive scan of the chain lengths -> offsets, off2[nheads] = nev */
    uint32_t carry = 0;
    for (uint32_t c0 = 0; c0 < nheads; c0 += CRGPU_WAVE) {
        const uint32_t c = c0 + lane;
        const uint32_t len = c < nheads ? V.off2[c] : 0u;
        const uint32_t incl = cr_scan_incl(len);
        if (c < nheads) V.off2[c] = carry + incl - len;
        carry += cr_lane_get(incl, 63);
    }
    if (lane == 0) { V.off2[nheads] = carry; V.ctr[1] = nheads; }
}

/* all threads: lay the order-2 chains out contiguously (event number and symbol), remember each
 * event's slot for the order-3 pass to drop its predicted byte into */
CR_DEV void cr_rop_scatter_o2(CrEvViews& V, uint32_t nev) {
    for (uint32_t i = threadIdx.x; i < nev; i += blockDim.x) {
        const uint32_t slot = V.off2[V.cid2[i]] + V.slot2[i];
        V.slot2[i] = slot;
        V.list2[slot] = i;
        V.csym[slot] = V.ev_sym[i];
    }
}

/* ------------------------------------------------------------------ k_rop_o3 */

/* one lane walks one order-3 chain: ppm_update_o3 (cr-ppm.c:69-88) with the table entry in registers */
CR_DEV void cr_rop_o3_chain(CrEvViews& V, uint32_t first) {
    uint32_t pred = 0, conf = 0;
    for (uint32_t i = first; i != 0xFFFFFFFFu; i = V.next3[i]) {
        const uint32_t sym = V.ev_sym[i] & 0x1ffu;
        V.ev_pred[i] = (uint8_t)pred;
        V.cpred[V.slot2[i]] = (uint8_t)pred;
        if (sym == pred) {
            conf += conf < 15u ? 1u : 0u;
        } else {
            uint32_t c = (uint32_t)(conf > 1u) + (uint32_t)(conf > 2u) + (uint32_t)(conf > 4u) + (uint32_t)(conf > 8u);
            if (c == 0u) { pred = sym; c = 1u; }
            conf = c;
        }
    }
}

/* ------------------------------------------------------------------ k_rop_o2 */

/* node of one order-2 chain: 256 byte counts in the lane's private 256-byte slice of the node area,
 * eight 32-symbol group sums and the two flag counts in registers */
struct CrLaneNode {
    uint8_t* cnt;
    uint32_t g[8];
    uint32_t fh, fe;
};

CR_DEV uint32_t cr_ln_sum4(uint32_t w) { return __builtin_amdgcn_sad_u8(w, 0u, 0u); }

/* o2_model_update's halving pass (cr-o2model.c:54-71), lane-serial over the 256 counts */
CR_DEV void cr_ln_halve(CrLaneNode& nd) {
    uint32_t singles = 1;
    for (uint32_t gi = 0; gi < 8u; gi++) {
        uint32_t gs = 0;
        uint4* p = reinterpret_cast<uint4*>(nd.cnt + gi * 32u);
        for (uint32_t h = 0; h < 2u; h++) {
            uint4 v = p[h];
            v.x = (v.x >> 1) & 0x7f7f7f7fu; v.y = (v.y >> 1) & 0x7f7f7f7fu; v.z = (v.z >> 1) & 0x7f7f7f7fu; v.w = (v.w >> 1) & 0x7f7f7f7fu;
            p[h] = v;
            gs += cr_ln_sum4(v.x) + cr_ln_sum4(v.y) + cr_ln_sum4(v.z) + cr_ln_sum4(v.w);
            singles += cr_count_ones_bytes(v.x) + cr_count_ones_bytes(v.y) + cr_count_ones_bytes(v.z) + cr_count_ones_bytes(v.w);
        }
        nd.g[gi] = gs;
    }
    nd.fh = (nd.fh + 1u) >> 1;
    nd.fe = singles & 0xffu;
}

/* sum of the counts of the symbols below `sym` (o2_model_cum, cr-o2model.c:75-84) */
CR_DEV uint32_t cr_ln_below(const CrLaneNode& nd, uint32_t sym) {
    const uint32_t gi = sym >> 5;
    uint32_t acc = 0;
    for (uint32_t j = 0; j < 8u; j++) acc += j < gi ? nd.g[j] : 0u;
    const uint32_t* w = reinterpret_cast<const uint32_t*>(nd.cnt + gi * 32u);
    const uint32_t within = sym & 31u;
    for (uint32_t k = 0; k < 8u; k++) {
        int take = (int)within - (int)(k * 4u);
        if (take <= 0) break;
        uint32_t m = take >= 4 ? 0xFFFFFFFFu : ((1u << (8 * take)) - 1u);
        acc += cr_ln_sum4(w[k] & m);
    }
    return acc;
}

CR_DEV void cr_ln_emit_mask(const CrLaneNode& nd, uint32_t* out8) {
    for (uint32_t wi = 0; wi < 8u; wi++) {
        const uint4* p = reinterpret_cast<const uint4*>(nd.cnt + wi * 32u);
        uint32_t bits = 0;
        for (uint32_t h = 0; h < 2u; h++) {
            uint4 v = p[h];
            const uint32_t ws[4] = {v.x, v.y, v.z, v.w};
            for (uint32_t k = 0; k < 4u; k++) {
                uint32_t x = ws[k], b = 0;
                b |= (x & 0x000000ffu) ? 1u : 0u;
                b |= (x & 0x0000ff00u) ? 2u : 0u;
                b |= (x & 0x00ff0000u) ? 4u : 0u;
                b |= (x & 0xff000000u) ? 8u : 0u;
                bits |= b << (h * 16u + k * 4u);
            }
        }
        out8[wi] = bits;
    }
}

/* one coding step of an order-2 chain: the order-2 part of ppm_encode (cr-ppm.c:108-146,159-162) */
CR_DEV void cr_rop_o2_event(CrEvViews& V, CrLaneNode& nd, uint32_t i, uint32_t sym, uint32_t pred) {
    const uint32_t pf = nd.cnt[pred];
    uint32_t bytes = 0;
    for (uint32_t j = 0; j < 8u; j++) bytes += nd.g[j];
    const uint32_t tot = bytes + nd.fh + nd.fe - pf;
    uint32_t cum, frq, type;
    if (sym == pred) {                                               /* cr-ppm.c:119-126 */
        cum = bytes - pf; frq = nd.fh; type = CR_T_HIT;
        nd.fh = (nd.fh + 1u) & 0xffu;
        if (nd.fh > 250u) cr_ln_halve(nd);
    } else {
        const uint32_t fs = nd.cnt[sym];
        if (fs) {                                                    /* cr-ppm.c:129-139 */
            cum = cr_ln_below(nd, sym) - (sym > pred ? pf : 0u); frq = fs; type = CR_T_BYTE;
            nd.cnt[sym] = (uint8_t)(fs + 1u);
            nd.g[sym >> 5] += 1u;
            if (fs + 1u > 250u) cr_ln_halve(nd);
            else if (fs + 1u == 2u) { nd.fe = (nd.fe - 1u) & 0xffu; if (nd.fe > 250u) cr_ln_halve(nd); }
        } else {                                                     /* cr-ppm.c:141-163 */
            cum = bytes + nd.fh - pf; frq = nd.fe; type = CR_T_ESC;
            nd.fe = (nd.fe + 1u) & 0xffu;
            bool halved = false;
            if (nd.fe > 250u) { cr_ln_halve(nd); halved = true; }
            cr_ln_emit_mask(nd, V.mask + (u64)i * 8u);               /* what the node knows NOW */
            if (!halved) { nd.cnt[sym] = 1; nd.g[sym >> 5] += 1u; }
        }
    }
    V.trip[i] = (u64)cum | ((u64)tot << 20) | ((u64)frq << 40) | ((u64)type << 50);
}

/* every lane of the workgroup keeps pulling chains from a shared counter (chains differ in length by
 * three orders of magnitude: a lane that finishes a short one must not idle behind a long one) */
CR_DEV void cr_rop_o2_all(CrEvViews& V, uint8_t* lane_counts, uint32_t nheads, uint32_t* next_head) {
    /* the 256 counts of the chain a lane is walking live in that lane's slice of LDS, and the chain's
     * events are read sequentially from the contiguous layout, one step ahead of their use: a chain
     * is a strictly serial run, so its per-event latency is what bounds the kernel */
    CrLaneNode nd;
    nd.cnt = lane_counts; nd.fh = 1; nd.fe = 1;
    for (uint32_t j = 0; j < 8u; j++) nd.g[j] = 0;
    uint32_t at = 0, end = 0;                 /* slots [at, end) of the current chain are still to do */
    uint32_t n_i = 0, n_sp = 0;               /* event number and sym | pred << 16 of slot `at` (prefetched) */
    for (;;) {
        if (at == end) {
            const uint32_t h = atomicAdd(next_head, 1u);
            if (h >= nheads) break;
            at = V.off2[h]; end = V.off2[h + 1u];
            for (uint32_t q = 0; q < 16u; q++) reinterpret_cast<uint4*>(nd.cnt)[q] = make_uint4(0u, 0u, 0u, 0u);   /* o2_model_init */
            for (uint32_t j = 0; j < 8u; j++) nd.g[j] = 0;
            nd.fh = 1; nd.fe = 1;
            n_i = V.list2[at]; n_sp = (uint32_t)V.csym[at] | ((uint32_t)V.cpred[at] << 16);
        }
        const uint32_t i = n_i, sp = n_sp;
        at++;
        if (at < end) { n_i = V.list2[at]; n_sp = (uint32_t)V.csym[at] | ((uint32_t)V.cpred[at] << 16); }
        cr_rop_o2_event(V, nd, i, sp & 0x1ffu, sp >> 16);
    }
}

/* ------------------------------------------------------------------ k_rop_rc */

/* 64-event register window over the per-event arrays */
struct CrEvWindow {
    uint32_t base;
    u64 trip;
    uint32_t ctx, sympred;     /* sym | pred << 16 */
};
CR_DEV void cr_evwin_fill(CrEvWindow& w, const CrEvViews& V, uint32_t at, uint32_t nev) {
    w.base = at;
    const uint32_t i = at + cr_lane();
    w.trip = 0; w.ctx = 0; w.sympred = 0;
    if (i < nev) { w.trip = V.trip[i]; w.ctx = V.ev_ctx[i]; w.sympred = (uint32_t)V.ev_sym[i] | ((uint32_t)V.ev_pred[i] << 16); }
    cr_drain_loads();
}

/* range coder over the prepared events + order-1 step of the escapes (cr-ppm.c:148-157) + output */
CR_DEV uint32_t cr_rop_code_events(const uint8_t* src, uint32_t n, uint8_t* dst, CrEvViews& V, uint8_t* arena,
                                   const CrArenaLayout& L) {
    const uint32_t lane = cr_lane();
    const uint32_t nev = cr_uni(V.ctr[0]), info = cr_uni(V.ctr[3]);
    const uint32_t esc = info & 0xffu;
    if (info & 0x100u) { cr_rop_store_raw(src, n, dst); return CR_ROP_HEADER + n; }
    uint8_t* o1 = arena + L.off_o1;
    cr_fill(o1, 65536u, 0x01010101u);
    cr_wave_sync();
    CrSink out; out.dst = dst + CR_ROP_HEADER; out.n = 0;
    CrRc rc; cr_rc_init(rc);
    CrEvWindow w;
    cr_evwin_fill(w, V, 0, nev);
    uint32_t lr_idx = 0xFFFFFFFFu, lr_row = 0;
    /* the next escape inside the event window: its order-1 row and exclusion words are fetched as soon
     * as the previous escape has stored its row (rows only change at escapes, so the load is current
     * unless it is the very row just modified, which lr_row covers) */
    uint32_t pf_at = 0xFFFFFFFFu, pf_row = 0, pf_mask = 0;
#define CR_RC_PREFETCH(from_lane_) do { \
        const u64 em_ = cr_ballot(((uint32_t)(w.trip >> 50) & 3u) == CR_T_ESC) & ~((1ull << (from_lane_)) - 1ull); \
        pf_at = 0xFFFFFFFFu; \
        if (em_) { \
            const uint32_t nl_ = (uint32_t)__builtin_ctzll(em_); \
            pf_at = w.base + nl_; \
            const uint32_t ri_ = cr_lane_get(w.ctx, nl_) & 0xffu; \
            pf_row = reinterpret_cast<const uint32_t*>(o1 + (ri_ << 8))[lane]; \
            pf_mask = V.mask[(u64)pf_at * 8u + (lane >> 3)]; \
        } } while (0)
    CR_RC_PREFETCH(0u);
    bool stored = false;
    for (uint32_t i = 0; i < nev; i++) {
        if (i - w.base >= CRGPU_WAVE) { cr_evwin_fill(w, V, i, nev); CR_RC_PREFETCH(0u); }
        const uint32_t l = i - w.base;
        const u64 t = cr_lane_get64(w.trip, l);
        const uint32_t cum = (uint32_t)t & 0xfffffu, tot = (uint32_t)(t >> 20) & 0xfffffu, frq = (uint32_t)(t >> 40) & 0x3ffu, type = (uint32_t)(t >> 50) & 3u;
        const uint32_t sp = cr_lane_get(w.sympred, l);
        cr_rc_pin(rc); out.n = cr_uni(out.n);
        cr_rc_encode(rc, cum, frq, tot, out);
        if (type == CR_T_ESC) {
            const uint32_t sym = sp & 0x1ffu, pred = sp >> 16, ridx = cr_lane_get(w.ctx, l) & 0xffu;
            uint8_t* rowp = o1 + (ridx << 8);
            uint32_t row, mw;
            if (pf_at == i) { row = pf_row; mw = pf_mask; }
            else { row = reinterpret_cast<const uint32_t*>(rowp)[lane]; mw = V.mask[(u64)i * 8u + (lane >> 3)]; }
            if (ridx == lr_idx) row = lr_row;
            const uint32_t present = (mw >> ((lane & 7u) * 4u)) & 0xfu;          /* bit j: byte 4*lane+j has a count */
            uint32_t keep = 0;
            if (!(present & 1u)) keep |= 0x000000ffu;
            if (!(present & 2u)) keep |= 0x0000ff00u;
            if (!(present & 4u)) keep |= 0x00ff0000u;
            if (!(present & 8u)) keep |= 0xff000000u;
            if (lane == (pred >> 2)) keep &= ~(0xffu << ((pred & 3u) * 8u));
            const uint32_t all = cr_sum(cr_o1_weight_sum(row, keep));
            const uint32_t lo = cr_sum(cr_o1_weight_sum(row, keep & cr_mask_below(lane, sym)));
            const uint32_t fo = cr_table_byte(row, sym) * 8u - 7u;
            cr_rc_pin(rc); out.n = cr_uni(out.n);
            cr_rc_encode(rc, lo, fo, all, out);
            /* ppm_update_o1, cr-ppm.c:90-97 */
            const uint32_t cur = cr_table_byte(row, sym);
            if (lane == (sym >> 2)) row += 1u << ((sym & 3u) * 8u);
            if (cur + 1u >= 255u) { row -= (row >> 1) & 0x7f7f7f7fu; reinterpret_cast<uint32_t*>(rowp)[lane] = row; }
            else if (lane == (sym >> 2)) reinterpret_cast<uint32_t*>(rowp)[lane] = row;
            lr_idx = ridx; lr_row = row;
            if (l + 1u < CRGPU_WAVE) CR_RC_PREFETCH(l + 1u); else pf_at = 0xFFFFFFFFu;
        }
        if ((sp & CR_EV_LAST) && CR_ROP_HEADER + out.n >= n) { stored = true; break; }   /* cr-coder.c:204-206 */
    }
#undef CR_RC_PREFETCH
    if (stored) { cr_wave_sync(); cr_rop_store_raw(src, n, dst); return CR_ROP_HEADER + n; }
    cr_rc_pin(rc);
    cr_rc_flush(rc, out);
    if (lane < CR_ROP_HEADER) {                                          /* cr-coder.c:213-216 */
        uint32_t v = 0;
        if (lane == 0) v = 1;
        else if (lane >= 4 && lane < 8) v = (n >> (8u * (lane - 4u))) & 0xffu;
        else if (lane == 8) v = esc;
        else if (lane >= 9 && lane < 18) v = src[lane - 9u];
        dst[lane] = (uint8_t)v;
    }
    return CR_ROP_HEADER + out.n;
}

#endif

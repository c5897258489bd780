/*
 * comprox_amd/csrc/crgpu_links2.h — the event sorts of the chain encoder in LDS (kernel k_rop_links_lds, 8 waves per
 * datablock, blocks of up to 28 672 events).
 *
 * Reference: what is being grouped are the model look-ups of /root/reference/src/cr-ppm.c:103-167 — the order-2 node
 * `o2_models[context & 0xffff]` and the order-3 entry of key cr-ppm.c:66 — per event, in coding order (crgpu_rop2.h
 * explains why every component of ppm_encode but the range coder is a per-key recurrence).
 *
 * Same views as cr_rop_sort_events (crgpu_rop2.h: list2 / csym2 / slot2 / chains2, csym3 / cslot3 / starts3), same
 * stable order. k_rop_links sorts {key, event} pairs of 8 bytes through global memory, five passes of 8 bits, tile by
 * tile through a 32 KB LDS buffer: 4.5 ms and 6.4 GB of traffic on the bench shard for 24 900 events per block. Here the
 * records are u16 event numbers ping-ponging between two LDS buffers (the machinery of crgpu_lzp2.h: cr_lz2_pass), and
 * a pass only needs ONE BYTE of every event's key: that digit is laid out per event in LDS (28 KB, where the LZP kernel
 * keeps the block) from a coalesced read of the contexts in front of every pass. After the last pass of a sort the
 * free record buffer takes the rest of the key, so that "same key as my neighbour" is answered out of LDS as well.
 * Blocks with more events (codec stage alone: ~43 000 per 64 KiB block) keep k_rop_links.
 */
#ifndef CRGPU_LINKS2_H
#define CRGPU_LINKS2_H

#include "crgpu_lzp2.h"

struct CrEvDigitKey {                         /* "key" of a record for cr_lz2_pass: the digit laid out for this pass */
    const uint8_t* d;
    CR_DEV uint32_t operator()(uint32_t i) const { return d[i]; }
};

struct CrLinks2Shared { uint32_t n2, n3, front, back; };

/* digit `which` of every event's key into S.src: 0 / 1 = order-2 context bits 0-7 / 8-15, 2 / 3 / 4 = order-3 key bits
 * 0-7 / 8-15 / 16-21 */
CR_DEV void cr_links2_digits(const CrLz2Shared& S, const uint32_t* ev_ctx, uint32_t nev, int which) {
    for (uint32_t i = threadIdx.x; i < nev; i += blockDim.x) {
        const uint32_t c = ev_ctx[i];
        const uint32_t k = which < 2 ? (c & 0xffffu) : cr_o3_key(c);
        const uint32_t sh = which < 2 ? (uint32_t)which * 8u : (uint32_t)(which - 2) * 8u;
        S.src[i] = (uint8_t)(k >> sh);
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

#endif

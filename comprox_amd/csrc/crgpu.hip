/*
 * comprox_amd/csrc/crgpu.hip — kernels and C-ABI of libcrgpu.so (declared in include/crgpu.h).
 *
 * Kernels (one wavefront per workgroup, persistent, ticket-scheduled):
 *   k_rop_encode   reset_models()+lzencode() per datablock   (ropmain/cr-coder.c:73-83,119-229)
 *   k_rop_decode   reset_models()+lzdecode() per datablock   (ropmain/cr-coder.c:231-292)
 * Host side: context (device, stream, arena, HIP events), batched entry points with device or
 * host pointers, and the reference's data_block_t / reset_models / lzencode / lzdecode symbols.
 * There is no CPU path: without a gfx950 device every call fails with CRGPU_E_NODEVICE.
 */
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "../../include/crgpu.h"
#include "crgpu_rop.h"
#include "crgpu_dict.h"
#include "crgpu_rox.h"
#include "crgpu_rolz.h"
#include "crgpu_rop2.h"
#include "crgpu_rop5.h"
#include "crgpu_rox5.h"
#include "crgpu_rolz5.h"
#include "crgpu_rox2.h"
#include "crgpu_rolz2.h"
#include "crgpu_lzp2.h"
#include "crgpu_links2.h"
#include "crgpu_rolz3.h"
#include "crgpu_rox3.h"

/* ------------------------------------------------------------------ kernels */

__global__ __launch_bounds__(CRGPU_WAVE) void k_rop_encode(CrBatch B, CrArenaLayout L) {
    __shared__ CrShared sh;
    uint8_t* arena = B.arena + (u64)blockIdx.x * L.stride;
    for (;;) {
        uint32_t t = 0;
        if (threadIdx.x == 0) t = atomicAdd(B.ticket, 1u);
        const uint32_t b = cr_uni(t);
        if (b >= B.nblocks) break;
        uint32_t n = B.in_size[b];
        uint32_t r;
        if (n > L.max_block) r = 0xFFFFFFFFu;
        else r = cr_rop_encode_block(B.in + B.in_off[b], n, B.out + B.out_off[b], B.lens + (u64)b * B.lens_stride, arena, L, B.fresh, B.persist, sh,
                                         B.stats ? B.stats + (u64)b * 16u : nullptr);
        if (threadIdx.x == 0) B.out_size[b] = r;
        cr_wave_sync();
    }
}

/* LZP agreement lengths for every position of every block: 4 waves per datablock, persistent */
/* word of CrBatch::ticket in which the 28 KiB pre-pass kernel of a launch (k_rop_lzp_lds / k_rox_links_lds / k_rolz_match_lds) counts
 * the blocks it leaves to the kernels behind it, and the one in which k_rop_links_lds does: with nothing left (the bench's batch:
 * every block shrinks below 28 KiB) those kernels return at once instead of walking 1 526 tickets each (4 x ~30 us per step) */
#define CR_TK_LZP_LEFT   12
#define CR_TK_LINKS_LEFT 13
__global__ __launch_bounds__(256, 6) void k_rop_lzp(CrBatch B, CrArenaLayout L) {
    __shared__ uint32_t s_ticket;
    uint8_t* arena = B.arena + (u64)blockIdx.x * L.stride;
    if (B.lzp_lds && B.ticket[CR_TK_LZP_LEFT] == 0u) return;
    for (;;) {
        if (threadIdx.x == 0) s_ticket = atomicAdd(B.ticket + 1, 1u);
        __syncthreads();
        const uint32_t b = s_ticket;
        __syncthreads();
        if (b >= B.nblocks) break;
        const uint32_t n = B.in_size[b];
        if (n > L.max_block || n <= CR_LZP_TAIL + CR_LZP_SKIP) continue;
        if (B.lzp_lds && B.pre_done[b]) continue;            /* an LDS kernel did this block */
        CrLzp z;
        cr_lzp_attach(z, arena, L, cr_log2_ceil_pow2(2u * n, 1024u, L.cap_lz));
        cr_lzp_reset_wg(z);
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        __syncthreads();
        CrLzpScratch sc;
        sc.c8 = reinterpret_cast<uint32_t*>(arena + L.off_cand);
        sc.c4 = sc.c8 + L.max_block;
        sc.c2 = sc.c4 + L.max_block;
        cr_lzp_block_parallel(z, sc, B.in + B.in_off[b], n, B.lens + (u64)b * B.lens_stride);
        __syncthreads();
    }
}

/* the same answers for blocks of up to 28 672 bytes without tables: positions sorted by key in LDS (crgpu_lzp2.h) */
__global__ __launch_bounds__(CR_LZ2_THREADS) void k_rop_lzp_lds(CrBatch B, CrArenaLayout L) {
    extern __shared__ __attribute__((aligned(16))) uint8_t s_lz2[];
    __shared__ uint32_t s_ticket;
    const CrLz2Shared S = cr_lz2_carve(s_lz2, CR_LZ2_THREADS / 64u);
    uint8_t* arena = B.arena + (u64)blockIdx.x * L.stride;
    for (;;) {
        if (threadIdx.x == 0) s_ticket = atomicAdd(B.ticket + 8, 1u);
        __syncthreads();
        const uint32_t b = s_ticket;
        __syncthreads();
        if (b >= B.nblocks) break;
        const uint32_t n = B.in_size[b];
        if (n > CR_LZ2_MAXN && n <= L.max_block && threadIdx.x == 0) atomicAdd(B.ticket + CR_TK_LZP_LEFT, 1u);   /* one for the kernels behind this one */
        if (n > CR_LZ2_MAXN || n > L.max_block || n <= CR_LZP_TAIL + CR_LZP_SKIP) continue;
        CrLzpScratch sc;
        sc.c8 = reinterpret_cast<uint32_t*>(arena + L.off_cand);
        sc.c4 = sc.c8 + L.max_block;
        sc.c2 = sc.c4 + L.max_block;
        cr_lzp_block_lds(S, sc, B.in + B.in_off[b], n, B.lens + (u64)b * B.lens_stride);
        if (threadIdx.x == 0) B.pre_done[b] = 1;
        __syncthreads();
    }
}

/* blocks of 28 673 .. 65 537 bytes: the same sort in groups by key beside the staged block (crgpu_lzp2.h, round 4); a block
 * whose keys do not split into groups (one key for a third of its positions) stays unmarked and goes to the table sweep */
__global__ __launch_bounds__(CR_LZ2_THREADS) void k_rop_lzp_lds64(CrBatch B, CrArenaLayout L) {
    extern __shared__ __attribute__((aligned(16))) uint8_t s_lz2[];
    __shared__ uint32_t s_ticket;
    __shared__ CrLz3Groups s_groups;
    const CrLz2Shared S = cr_lz3_carve(s_lz2, CR_LZ2_THREADS / 64u);
    uint8_t* arena = B.arena + (u64)blockIdx.x * L.stride;
    if (B.ticket[CR_TK_LZP_LEFT] == 0u) return;
    for (;;) {
        if (threadIdx.x == 0) s_ticket = atomicAdd(B.ticket + 10, 1u);
        __syncthreads();
        const uint32_t b = s_ticket;
        __syncthreads();
        if (b >= B.nblocks) break;
        const uint32_t n = B.in_size[b];
        if (n <= CR_LZ2_MAXN || n > CR_LZ3_MAXN || n > L.max_block) continue;
        CrLzpScratch sc;
        sc.c8 = reinterpret_cast<uint32_t*>(arena + L.off_cand);
        sc.c4 = sc.c8 + L.max_block;
        sc.c2 = sc.c4 + L.max_block;
        const bool ok = cr_lzp_block_lds64(S, s_groups, sc, B.in + B.in_off[b], n, B.lens + (u64)b * B.lens_stride, B.stats ? B.stats + (u64)b * 16u : nullptr);
        if (ok && threadIdx.x == 0) B.pre_done[b] = 2;
        __syncthreads();
    }
}

__global__ __launch_bounds__(CRGPU_WAVE) void k_rop_decode(CrBatch B, CrArenaLayout L) {
    __shared__ CrShared sh;
    uint8_t* arena = B.arena + (u64)blockIdx.x * L.stride;
    for (;;) {
        uint32_t t = 0;
        if (threadIdx.x == 0) t = atomicAdd(B.ticket, 1u);
        const uint32_t b = cr_uni(t);
        if (b >= B.nblocks) break;
        uint32_t r = cr_rop_decode_block(B.in + B.in_off[b], B.in_size[b], B.out + B.out_off[b], B.out_cap[b],
                                         arena, L, B.fresh, B.persist, sh, B.stats ? B.stats + (u64)b * 16u : nullptr);
        if (threadIdx.x == 0) B.out_size[b] = r;
        cr_wave_sync();
    }
}

/* batched API: every block starts from a fresh model; the coding step in assembly (crgpu_rop5.h) */
/* 256 bytes of LDS per decoding wave: where a line's {symbol, count} pairs are scattered into the 256-byte layout of the
 * order-1 rows / dense nodes (crgpu_rop5.h); the statement gets the buffer's LDS address */
#define CR_V5_LDS_SCRATCH(name_) CR_V5_LDS_SCRATCH_N(name_, 68u)
#define CR_V5_LDS_SCRATCH_N(name_, words_) __shared__ __attribute__((aligned(256))) uint32_t name_[words_]; \
    const uint32_t name_##_at = (uint32_t)reinterpret_cast<uintptr_t>(&name_[0])

template <int DL> CR_DEV void cr_rop_decode_v5_blocks(const CrBatch& B, const CrArenaLayout& L, uint32_t lds_at) {
    __builtin_amdgcn_s_setprio(3);                      /* a dependent chain: its instructions go first when another stream's kernels share the SIMD */
    uint8_t* arena = B.arena + (u64)blockIdx.x * L.stride;
    for (;;) {
        uint32_t t = 0;
        if (threadIdx.x == 0) t = atomicAdd(B.ticket, 1u);
        const uint32_t b = cr_uni(t);
        if (b >= B.nblocks) break;
        uint32_t r = cr_rop_decode_v5<0, DL>(B.in + B.in_off[b], B.in_size[b], B.out + B.out_off[b], B.out_cap[b], arena, L, lds_at,
                                             B.stats ? B.stats + (u64)b * 16u : nullptr);
        if (threadIdx.x == 0) B.out_size[b] = r;
        cr_wave_sync();
    }
}
/* k_rop_decode_v5: the wave's LDS also holds the first CR_V5_DLDS dense nodes (8.7 KB per workgroup; crgpu_rop5.h).
 * k_rop_decode_v5s (CRGPU_OPT_DECODER_LDS_NODES 0): 272 bytes — six of them fit on a CU beside a sorting kernel's 152 KB, which is what
 * a caller wants who runs another context's encode calls beside this one's decodes (bench.py's two steps in flight). */
__global__ __launch_bounds__(CRGPU_WAVE) void k_rop_decode_v5(CrBatch B, CrArenaLayout L) {
    CR_V5_LDS_SCRATCH_N(s_px, CR_V5_LDS_BYTES / 4u);
    cr_rop_decode_v5_blocks<1>(B, L, s_px_at);
}
__global__ __launch_bounds__(CRGPU_WAVE) void k_rop_decode_v5s(CrBatch B, CrArenaLayout L) {
    CR_V5_LDS_SCRATCH(s_px);
    cr_rop_decode_v5_blocks<0>(B, L, s_px_at);
}

/* the same with a helper wave per block (CRGPU_OPT_DECODER_HELPER, crgpu_rop5.h: CR_V5_ASM_MODE_HW): wave 0 is the coder, wave 1
 * prepares every step's order-1 sums from what the coder posts in LDS and ends on the coder's stop word */
__global__ __launch_bounds__(2 * CRGPU_WAVE) void k_rop_decode_v5h(CrBatch B, CrArenaLayout L) {
    __shared__ __attribute__((aligned(256))) uint32_t s_px[CR_V5_HW_LDS_BYTES / 4u];
    const uint32_t s_px_at = (uint32_t)reinterpret_cast<uintptr_t>(&s_px[0]);
    if (threadIdx.x < 4u) s_px[68u + threadIdx.x] = 0u;                /* post word, answer word, total, the coder's sequence number (LDS keeps the last launch's stop word) */
    __syncthreads();
    if (threadIdx.x >= CRGPU_WAVE) { cr_rop_decode_helper(s_px_at + 272u); return; }
    __builtin_amdgcn_s_setprio(3);
    uint8_t* arena = B.arena + (u64)blockIdx.x * L.stride;
    for (;;) {
        uint32_t t = 0;
        if (threadIdx.x == 0) t = atomicAdd(B.ticket, 1u);
        const uint32_t b = cr_uni(t);
        if (b >= B.nblocks) break;
        uint32_t r = cr_rop_decode_v5<1, 0>(B.in + B.in_off[b], B.in_size[b], B.out + B.out_off[b], B.out_cap[b], arena, L, s_px_at,
                                         B.stats ? B.stats + (u64)b * 16u : nullptr);
        if (threadIdx.x == 0) B.out_size[b] = r;
        cr_wave_sync();
    }
    if (threadIdx.x == 0) s_px[68u] = 0xffffffffu;                       /* every wave reaches its exit: the helper ends on this word */
}

/* comprop, context-partitioned encoder (crgpu_rop2.h) ---------------------------------------- */

#define CR_TICKET_LOOP(word_, body_) \
    for (;;) { \
        __shared__ uint32_t s_tk; \
        if (threadIdx.x == 0) s_tk = atomicAdd(B.ticket + (word_), 1u); \
        __syncthreads(); \
        const uint32_t b = s_tk; \
        __syncthreads(); \
        if (b >= B.nblocks) break; \
        body_ \
        __syncthreads(); \
    }

__global__ __launch_bounds__(CRGPU_WAVE) void k_rop_events(CrBatch B, CrArenaLayout L) {
    __shared__ CrShared sh;
    CR_TICKET_LOOP(2, {
        const uint32_t n = B.in_size[b];
        CrEvViews V = cr_ev_views(B.ev + (u64)b * B.ev_stride, B.ev_cap);
        if (n <= L.max_block) cr_rop_emit_events(B.in + B.in_off[b], n, B.lens + (u64)b * B.lens_stride, V, sh);
        else if (threadIdx.x == 0) { V.ctr[0] = 0; V.ctr[1] = 0; V.ctr[2] = 0; V.ctr[3] = 0x200u; }
    })
}

__global__ __launch_bounds__(CR_SORT_THREADS) void k_rop_links(CrBatch B, CrArenaLayout L) {
    __shared__ CrSortShared sh;
    uint8_t* arena = B.arena + (u64)blockIdx.x * L.stride;
    if (B.links_lds && B.ticket[CR_TK_LINKS_LEFT] == 0u) return;
    CR_TICKET_LOOP(3, {
        CrEvViews V = cr_ev_views(B.ev + (u64)b * B.ev_stride, B.ev_cap);
        const uint32_t nev = V.ctr[0];
        if (nev && !(B.links_lds && (B.pre_done[b] & 0x30u)))  /* (an LDS kernel has sorted the block's events) */
            cr_rop_sort_events(sh, V, reinterpret_cast<uint32_t*>(arena + L.off_lz2), nev, B.stats ? B.stats + (u64)b * 16u : nullptr);
    })
}

/* the same views for blocks of up to 28 672 events: both sorts in LDS (crgpu_links2.h), one block per CU at a time */
__global__ __launch_bounds__(CR_LZ2_THREADS) void k_rop_links_lds(CrBatch B, CrArenaLayout L) {
    extern __shared__ __attribute__((aligned(16))) uint8_t s_lz2[];
    __shared__ CrLinks2Shared sh;
    const CrLz2Shared S = cr_lz2_carve(s_lz2, CR_LZ2_THREADS / 64u);
    uint8_t* arena = B.arena + (u64)blockIdx.x * L.stride;
    CR_TICKET_LOOP(9, {
        CrEvViews V = cr_ev_views(B.ev + (u64)b * B.ev_stride, B.ev_cap);
        const uint32_t nev = V.ctr[0];
        if (nev && nev <= CR_LZ2_MAXN) {
            cr_rop_sort_events_lds(S, sh, V, reinterpret_cast<uint32_t*>(arena + L.off_lz2), nev);
            if (threadIdx.x == 0) B.pre_done[b] |= 0x10u;
        } else if (nev && threadIdx.x == 0) {
            atomicAdd(B.ticket + CR_TK_LINKS_LEFT, 1u);
        }
    })
}

/* 28 673 .. 65 536 events (a 64 KiB block of text is ~43 000): records that carry their key, sorted in groups by key (crgpu_links2.h,
 * round 4); a block whose keys do not split stays unmarked and goes to k_rop_links */
__global__ __launch_bounds__(CR_LK4_THREADS) void k_rop_links_lds64(CrBatch B, CrArenaLayout L) {
    extern __shared__ __attribute__((aligned(16))) uint8_t s_lz2[];
    __shared__ CrLinks2Shared sh;
    __shared__ CrLz3Groups s_groups, s_groups3;
    const CrLz2Shared S = cr_lk4_carve(s_lz2, CR_LK4_THREADS / 64u);
    uint8_t* arena = B.arena + (u64)blockIdx.x * L.stride;
    if (B.ticket[CR_TK_LINKS_LEFT] == 0u) return;
    CR_TICKET_LOOP(11, {
        CrEvViews V = cr_ev_views(B.ev + (u64)b * B.ev_stride, B.ev_cap);
        const uint32_t nev = V.ctr[0];
        if (nev > CR_LZ2_MAXN && nev <= 65536u) {
            const bool ok = cr_rop_sort_events_lk4(S, s_groups, s_groups3, sh, V, reinterpret_cast<uint32_t*>(arena + L.off_lz2), nev, B.stats ? B.stats + (u64)b * 16u : nullptr);
            if (ok && threadIdx.x == 0) B.pre_done[b] |= 0x20u;
        }
    })
}

__global__ __launch_bounds__(256) void k_rop_o3(CrBatch B, CrArenaLayout L) {
    (void)L;
    __shared__ CrO2Ranges s_ranges;
    CR_TICKET_LOOP(4, {
        CrEvViews V = cr_ev_views(B.ev + (u64)b * B.ev_stride, B.ev_cap);
        const uint32_t nev = V.ctr[0];
        /* (the triples start out as CR_TRIP_FRESH: k_rop_o2, the next kernel, stores only the ones that differ) */
        for (uint32_t i = threadIdx.x * 2u; i < nev; i += blockDim.x * 2u) { const u64 f = CR_TRIP_FRESH; *reinterpret_cast<ulonglong2*>(V.trip + i) = make_ulonglong2(f, f); }
        if (nev && nev <= CR_O2R_MAXEV && !B.o2_tickets) {
            cr_rop_o3_ranges(V, s_ranges, nev);
        } else {
            const uint32_t nc = nev ? V.ctr[2] : 0u;
            for (uint32_t c = threadIdx.x; c < nc; c += blockDim.x) cr_rop_o3_chain(V, V.starts3[c]);
        }
    })
}

#define CR_O2_THREADS 64u
__global__ __launch_bounds__(CR_O2_THREADS) void k_rop_o2(CrBatch B, CrArenaLayout L) {
    (void)L;
    __shared__ uint32_t s_next_chain;
    __shared__ __attribute__((aligned(16))) uint8_t s_nodes[CR_O2_THREADS * CR_LN_STRIDE];
    __shared__ CrO2Ranges s_ranges;
    CR_TICKET_LOOP(5, {
        CrEvViews V = cr_ev_views(B.ev + (u64)b * B.ev_stride, B.ev_cap);
        const uint32_t nev = V.ctr[0];
        if (nev && nev <= CR_O2R_MAXEV && !B.o2_tickets) {
#ifdef CR_O2_PROF                                                 /* tools/o2_profile.py: stamps in a second stats region no later kernel writes */
            cr_rop_o2_ranges(V, s_nodes + threadIdx.x * CR_LN_STRIDE, s_ranges, nev, B.stats ? B.stats + ((u64)B.nblocks + b) * 16u : nullptr);
#else
            cr_rop_o2_ranges(V, s_nodes + threadIdx.x * CR_LN_STRIDE, s_ranges, nev);
#endif
        } else {
            const uint32_t nc = nev ? V.ctr[1] : 0u;
            if (threadIdx.x == 0) s_next_chain = 0;
            __syncthreads();
            cr_rop_o2_all(V, s_nodes + threadIdx.x * CR_LN_STRIDE, nc, &s_next_chain);
        }
    })
}

__global__ __launch_bounds__(CR_O1_THREADS) void k_rop_o1(CrBatch B, CrArenaLayout L) {
    (void)L;
    __shared__ CrSortShared sh;
    /* the rows' LDS (exclusion sets staged per wave, 10 KB) lies over the sort pass's tile buffer, which is dead by then: 38 KB per
     * workgroup instead of 48, i.e. four workgroups per CU instead of three (every phase of this kernel is latency-bound) */
    static_assert(sizeof(sh.buf) >= CR_SORT_WAVES * CR_O1_LDS_PER_WAVE * 4u, "the order-1 rows' LDS lies over the sort tile");
    uint32_t* const s_masks = reinterpret_cast<uint32_t*>(sh.buf);
    CR_TICKET_LOOP(7, {
        CrEvViews V = cr_ev_views(B.ev + (u64)b * B.ev_stride, B.ev_cap);
        const uint32_t nev = (V.ctr[3] & 0x300u) ? 0u : V.ctr[0];
        /* (B.o2_tickets = CRGPU_OPT_LZP_TABLES, the switch for the older kernels: one escape per wave-step throughout) */
        if (nev) cr_rop_o1_all(sh, V, s_masks, nev, B.o2_tickets, B.stats ? B.stats + (u64)b * 16u : nullptr);
    })
}

__global__ __launch_bounds__(CRGPU_WAVE) void k_rop_rc(CrBatch B, CrArenaLayout L) {
    (void)L;
    __shared__ u64 s_ring[CR_RC_RING];
    CR_TICKET_LOOP(6, {
        CrEvViews V = cr_ev_views(B.ev + (u64)b * B.ev_stride, B.ev_cap);
        uint32_t r = 0xFFFFFFFFu;
        if (!(V.ctr[3] & 0x200u)) {
            r = cr_rop_code_events_fast(B.in + B.in_off[b], B.in_size[b], B.out + B.out_off[b], V, s_ring);
            if (r == 0u) r = cr_rop_code_events(B.in + B.in_off[b], B.in_size[b], B.out + B.out_off[b], V);
        }
        if (threadIdx.x == 0) B.out_size[b] = r;
    })
}

/* comprox codec ---------------------------------------------------------------------------- */

CR_DEV CrRoxTables cr_rox_tables(const CrBatch& B, const CrArenaLayout& L, uint32_t b, uint8_t* arena) {
    CrRoxTables T;
    uint8_t* base = B.rox + (u64)b * B.rox_stride;
    const u64 n4 = (B.rox_stride / 16u) & ~(u64)63u;           /* positions the slot was sized for */
    T.prev = reinterpret_cast<uint32_t*>(base);
    T.nprev = T.prev + n4;
    T.ml_pos = T.nprev + n4;
    T.ml_len = reinterpret_cast<uint8_t*>(T.ml_pos + n4);
    T.nl_len = T.ml_len + n4;
    T.m0_len = T.nl_len + n4;
    T.cls_last = arena ? reinterpret_cast<uint32_t*>(arena + L.off_rox_cls) : nullptr;
    T.near_last = arena ? reinterpret_cast<uint32_t*>(arena + L.off_rox_near) : nullptr;
    return T;
}

/* per-position match tables for every block: chains + short cache (2 sweep waves), then all waves */
__global__ __launch_bounds__(256) void k_rox_match(CrBatch B, CrArenaLayout L) {
    __shared__ uint32_t s_ticket;
    uint8_t* arena = B.arena + (u64)blockIdx.x * L.stride;
    for (;;) {
        if (threadIdx.x == 0) s_ticket = atomicAdd(B.ticket + 1, 1u);
        __syncthreads();
        const uint32_t b = s_ticket;
        __syncthreads();
        if (b >= B.nblocks) break;
        const uint32_t n = B.in_size[b];
        if (n > L.max_block || n <= CR_ROX_TAIL) continue;
        const uint8_t* src = B.in + B.in_off[b];
        const uint32_t long_min = 10u + (n > 16777216u ? 1u : 0u);
        CrRoxTables T = cr_rox_tables(B, L, b, arena);
        if (!(B.lzp_lds && B.pre_done[b])) {                     /* (an LDS kernel has laid the block's links) */
            const u64 cls_bytes = ((u64)20u * (20u + n / 25u) * 4u + 15u) & ~(u64)15u;
            cr_fill_wg(reinterpret_cast<uint8_t*>(T.cls_last), cls_bytes, 0u);
            cr_fill_wg(reinterpret_cast<uint8_t*>(T.near_last), 65536u * 4u, 0u);
            cr_fill_wg(reinterpret_cast<uint8_t*>(T.prev), ((u64)n * 4u + 15u) & ~(u64)15u, 0xFFFFFFFFu);
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
            __syncthreads();
            if (cr_wave_id() == 0) cr_rox_sweep_chains(src, n, long_min, T);
            else if (cr_wave_id() == 1) cr_rox_sweep_near(src, n, T);
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
            __syncthreads();
        }
        if (B.flexible) cr_rox_flex_all(src, n, long_min, B.rox_limit, T);
        else cr_rox_match_all(src, n, long_min, B.rox_limit, T);
        __syncthreads();
    }
}

/* chain and short-cache links of the blocks of up to 28 672 bytes by sorting their positions in LDS (crgpu_rox3.h) */
__global__ __launch_bounds__(CR_LZ2_THREADS) void k_rox_links_lds(CrBatch B, CrArenaLayout L) {
    extern __shared__ __attribute__((aligned(16))) uint8_t s_lz2[];
    __shared__ uint32_t s_ticket;
    const CrLz2Shared S = cr_lz2_carve(s_lz2, CR_LZ2_THREADS / 64u);
    for (;;) {
        if (threadIdx.x == 0) s_ticket = atomicAdd(B.ticket + 8, 1u);
        __syncthreads();
        const uint32_t b = s_ticket;
        __syncthreads();
        if (b >= B.nblocks) break;
        const uint32_t n = B.in_size[b];
        if (n > CR_LZ2_MAXN && n <= L.max_block && threadIdx.x == 0) atomicAdd(B.ticket + CR_TK_LZP_LEFT, 1u);
        if (n > CR_LZ2_MAXN || n > L.max_block || n <= CR_ROX_TAIL) continue;
        CrRoxTables T = cr_rox_tables(B, L, b, nullptr);
        cr_rox_links_block_lds(S, B.in + B.in_off[b], n, 10u, T);       /* match_min = 10 below 16 MiB (roxmain/cr-coder.c:192) */
        if (threadIdx.x == 0) B.pre_done[b] = 1;
        __syncthreads();
    }
}

/* the same for blocks of 28 673 .. 65 537 bytes: the sort in groups by key (crgpu_rox3.h, round 4) */
__global__ __launch_bounds__(CR_LZ2_THREADS) void k_rox_links_lds64(CrBatch B, CrArenaLayout L) {
    extern __shared__ __attribute__((aligned(16))) uint8_t s_lz2[];
    __shared__ uint32_t s_ticket;
    __shared__ CrLz3Groups s_groups;
    const CrLz2Shared S = cr_lz3_carve(s_lz2, CR_LZ2_THREADS / 64u);
    if (B.ticket[CR_TK_LZP_LEFT] == 0u) return;
    for (;;) {
        if (threadIdx.x == 0) s_ticket = atomicAdd(B.ticket + 10, 1u);
        __syncthreads();
        const uint32_t b = s_ticket;
        __syncthreads();
        if (b >= B.nblocks) break;
        const uint32_t n = B.in_size[b];
        if (n <= CR_LZ2_MAXN || n > CR_LZ3_MAXN || n > L.max_block) continue;
        CrRoxTables T = cr_rox_tables(B, L, b, nullptr);
        const bool ok = cr_rox_links_block_lds64(S, s_groups, B.in + B.in_off[b], n, 10u, T);
        if (ok && threadIdx.x == 0) B.pre_done[b] = 2;
        __syncthreads();
    }
}

__global__ __launch_bounds__(CRGPU_WAVE) void k_rox_encode(CrBatch B, CrArenaLayout L) {
    __shared__ CrRoxShared sh;
    uint8_t* arena = B.arena + (u64)blockIdx.x * L.stride;
    for (;;) {
        uint32_t t = 0;
        if (threadIdx.x == 0) t = atomicAdd(B.ticket, 1u);
        const uint32_t b = cr_uni(t);
        if (b >= B.nblocks) break;
        const uint32_t n = B.in_size[b];
        uint32_t r = 0xFFFFFFFFu;
        if (n <= L.max_block) {
            CrRoxTables T = cr_rox_tables(B, L, b, nullptr);
            r = cr_rox_encode_block(B.in + B.in_off[b], n, B.out + B.out_off[b], T, arena + L.off_side, L.side_stride, arena, L, B.fresh, B.persist, sh);
        }
        if (threadIdx.x == 0) B.out_size[b] = r;
        cr_wave_sync();
    }
}

__global__ __launch_bounds__(CRGPU_WAVE) void k_rox_decode(CrBatch B, CrArenaLayout L) {
    __shared__ CrRoxShared sh;
    uint8_t* arena = B.arena + (u64)blockIdx.x * L.stride;
    for (;;) {
        uint32_t t = 0;
        if (threadIdx.x == 0) t = atomicAdd(B.ticket, 1u);
        const uint32_t b = cr_uni(t);
        if (b >= B.nblocks) break;
        uint32_t r = cr_rox_decode_block(B.in + B.in_off[b], B.in_size[b], B.out + B.out_off[b], B.out_cap[b], arena, L, B.fresh, B.persist, sh);
        if (threadIdx.x == 0) B.out_size[b] = r;
        cr_wave_sync();
    }
}

/* comprox encoder on the comprop kernel pipeline (crgpu_rox2.h): token loop -> events + side streams ... */
__global__ __launch_bounds__(256) void k_rox_events(CrBatch B, CrArenaLayout L) {
    __shared__ CrRoxShared sh;
    CR_TICKET_LOOP(2, {
        const uint32_t n = B.in_size[b];
        CrEvViews V = cr_ev_views(B.ev + (u64)b * B.ev_stride, B.ev_cap);
        uint8_t* side = B.side + (u64)b * 3u * L.side_stride;
        if (n <= L.max_block) {
            if (cr_wave_id() == 0) {                        /* wave 0: the token loop ... */
                cr_side_reset(sh);
                CrRoxTables T = cr_rox_tables(B, L, b, nullptr);
                cr_rox_emit_events(B.in + B.in_off[b], n, T, side, L.side_stride, V, sh);
            }
            cr_wg_sync_global();
            if (cr_wave_id() < 3u) cr_rox_code_side(cr_wave_id(), side, L.side_stride, V, sh);   /* ... then a wave per side stream */
        } else if (threadIdx.x == 0) { V.ctr[0] = 0; V.ctr[1] = 0; V.ctr[2] = 0; V.ctr[3] = 0x200u; }
    })
}

/* ... and, behind k_rop_links / _o3 / _o2 / _o1, the main stream's range coder and the block's assembly */
__global__ __launch_bounds__(CRGPU_WAVE) void k_rox_rc(CrBatch B, CrArenaLayout L) {
    __shared__ u64 s_ring[CR_RC_RING];
    CR_TICKET_LOOP(6, {
        CrEvViews V = cr_ev_views(B.ev + (u64)b * B.ev_stride, B.ev_cap);
        uint32_t r = 0xFFFFFFFFu;
        if (!(V.ctr[3] & 0x200u)) r = cr_rox_finish(B.in + B.in_off[b], B.in_size[b], B.out + B.out_off[b], B.side + (u64)b * 3u * L.side_stride, L.side_stride, V, s_ring);
        if (threadIdx.x == 0) B.out_size[b] = r;
    })
}

/* same contract, the PPM main stream in assembly (crgpu_rox5.h); fresh models per block only */
__global__ __launch_bounds__(CRGPU_WAVE) void k_rox_decode_v5(CrBatch B, CrArenaLayout L) {
    __builtin_amdgcn_s_setprio(3);                      /* a dependent chain: its instructions go first when another stream's kernels share the SIMD */
    __shared__ CrRoxShared sh;
    CR_V5_LDS_SCRATCH_N(s_px, CR_V5_SIDE_LDS_WORDS);
    uint8_t* arena = B.arena + (u64)blockIdx.x * L.stride;
    for (;;) {
        uint32_t t = 0;
        if (threadIdx.x == 0) t = atomicAdd(B.ticket, 1u);
        const uint32_t b = cr_uni(t);
        if (b >= B.nblocks) break;
        uint32_t r = cr_rox_decode_v5(B.in + B.in_off[b], B.in_size[b], B.out + B.out_off[b], B.out_cap[b], arena, L, sh, s_px_at);
        if (threadIdx.x == 0) B.out_size[b] = r;
        cr_wave_sync();
    }
}

/* comprolz codec --------------------------------------------------------------------------- */

CR_DEV CrRolzTables cr_rolz_tables_enc(const CrBatch& B, const CrArenaLayout& L, uint32_t b, uint8_t* arena) {
    CrRolzTables T;
    uint8_t* base = B.rox + (u64)b * B.rox_stride;              /* the per-block slot of the comprox match tables */
    const u64 n4 = (B.rox_stride / 16u) & ~(u64)63u;
    T.ring_prev = reinterpret_cast<uint32_t*>(base);
    T.row_prev = T.ring_prev + n4;
    T.rank = reinterpret_cast<uint8_t*>(T.row_prev + n4);
    T.len = T.rank + n4;
    T.raw16 = reinterpret_cast<uint16_t*>(T.rank + 2u * n4);                                     /* bytes 10-11 */
    T.row16 = reinterpret_cast<uint16_t*>(T.rank + 6u * n4);                                     /* bytes 14-15 */
    T.ring16 = B.in_size[b] <= 65536u ? reinterpret_cast<uint16_t*>(T.rank + 4u * n4) : nullptr;   /* bytes 12-13 of the 16 per position; links stay below n - 768 */
    T.ring_head = arena ? reinterpret_cast<uint32_t*>(arena + L.off_rolz_head) : nullptr;
    return T;
}

/* parse result at every position of every block: ring / row links (2 sweep waves), then all threads */
__global__ __launch_bounds__(256) void k_rolz_match(CrBatch B, CrArenaLayout L) {
    __shared__ uint32_t s_ticket;
    __shared__ uint32_t s_rows[256];
    uint8_t* arena = B.arena + (u64)blockIdx.x * L.stride;
    if (B.lzp_lds && B.ticket[CR_TK_LZP_LEFT] == 0u) return;   /* k_rolz_match_lds has done every block */
    for (;;) {
        if (threadIdx.x == 0) s_ticket = atomicAdd(B.ticket + 1, 1u);
        __syncthreads();
        const uint32_t b = s_ticket;
        __syncthreads();
        if (b >= B.nblocks) break;
        const uint32_t n = B.in_size[b];
        if (n > L.max_block || n <= CR_ROLZ_TAIL + CR_ROLZ_WARM) continue;
        if (B.lzp_lds && B.pre_done[b] == 1) continue;           /* k_rolz_match_lds did this block */
        const uint32_t mark = B.lzp_lds ? B.pre_done[b] : 0u;
        const bool rings_done = mark == 2u || mark == 3u;        /* k_rolz_rings_lds64 has laid the ring links and done the plain lookups */
        const bool rows_done = mark == 2u;                       /* ... and the row searches */
        const uint8_t* src = B.in + B.in_off[b];
        const bool ctx4 = n >= 4194304u;                        /* using_ctx4, cr-coder.c:158 */
        CrRolzTables T = cr_rolz_tables_enc(B, L, b, arena);
#ifdef CR_ROLZ_PROF                                              /* tools/rolz_match_profile.py: stamps (100 MHz) in a second stats region, which no later kernel writes: 6 start | 7 heads cleared | 8 links | 9 plain lookups | 10 parse */
        u64* const st = B.stats ? B.stats + ((u64)B.nblocks + b) * 16u : nullptr;
#else
        u64* const st = nullptr;
#endif
        cr_wg_stamp(st, 6);
        if (!rings_done) cr_fill_wg(reinterpret_cast<uint8_t*>(T.ring_head), (u64)CR_ROLZ_BUCKETS * 4u, 0u);   /* the 1 MB of ring heads only the sweep uses */
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        __syncthreads();
        cr_wg_stamp(st, 7);
        /* lookups happen below n - 1024; lazy evaluation reads the links of up to four positions more */
        const uint32_t link_limit = n - CR_ROLZ_TAIL + (B.flexible ? CR_ROLZ_MAX + 1u : CR_ROLZ_MIN);
        if (cr_wave_id() == 0) { if (!rings_done) cr_rolz_sweep_rings(src, link_limit, ctx4, T); }
        else if (cr_wave_id() == 1) { if (!rows_done) cr_rolz_sweep_rows(src, link_limit, T, s_rows); }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        __syncthreads();
        cr_wg_stamp(st, 8);
        cr_rolz_find_all(src, n, link_limit, ctx4, B.flexible != 0u, T, T.raw16, rings_done, rows_done, st);
        __syncthreads();
        cr_wg_stamp(st, 10);
    }
}

/* the same for blocks of up to 28 672 bytes with the block, the ring links and the plain lookups in LDS (crgpu_rolz3.h) */
#define CR_ROLZ3_THREADS 1024u         /* the searches are chains of LDS round trips: twice the lanes of k_rop_lzp_lds */
__global__ __launch_bounds__(CR_ROLZ3_THREADS) void k_rolz_match_lds(CrBatch B, CrArenaLayout L) {
    extern __shared__ __attribute__((aligned(16))) uint8_t s_lz2[];
    __shared__ uint32_t s_ticket;
    const CrLz2Shared S = cr_lz2_carve(s_lz2, CR_ROLZ3_THREADS / 64u);
    for (;;) {
        if (threadIdx.x == 0) s_ticket = atomicAdd(B.ticket + 8, 1u);
        __syncthreads();
        const uint32_t b = s_ticket;
        __syncthreads();
        if (b >= B.nblocks) break;
        const uint32_t n = B.in_size[b];
        if (n > CR_LZ2_MAXN && n <= L.max_block && threadIdx.x == 0) atomicAdd(B.ticket + CR_TK_LZP_LEFT, 1u);
        if (n > CR_LZ2_MAXN || n > L.max_block || n <= CR_ROLZ_TAIL + CR_ROLZ_WARM) continue;
        CrRolzTables T = cr_rolz_tables_enc(B, L, b, nullptr);
        cr_rolz_match_block_lds(S, B.in + B.in_off[b], n, B.flexible != 0u, T, B.stats ? B.stats + (u64)b * 16u : nullptr);
        if (threadIdx.x == 0) B.pre_done[b] = 1;
        __syncthreads();
    }
}

/* ring links of the blocks of 28 673 .. 65 537 bytes by the sort in groups by key (crgpu_rolz3.h, round 4) */
__global__ __launch_bounds__(CR_LZ2_THREADS) void k_rolz_rings_lds64(CrBatch B, CrArenaLayout L) {
    extern __shared__ __attribute__((aligned(16))) uint8_t s_lz2[];
    __shared__ uint32_t s_ticket;
    __shared__ CrLz3Groups s_groups;
    const CrLz2Shared S = cr_lz3_carve(s_lz2, CR_LZ2_THREADS / 64u);
    if (B.ticket[CR_TK_LZP_LEFT] == 0u) return;
    for (;;) {
        if (threadIdx.x == 0) s_ticket = atomicAdd(B.ticket + 10, 1u);
        __syncthreads();
        const uint32_t b = s_ticket;
        __syncthreads();
        if (b >= B.nblocks) break;
        const uint32_t n = B.in_size[b];
        if (n <= CR_LZ2_MAXN || n > CR_LZ3_MAXN || n > L.max_block) continue;
        CrRolzTables T = cr_rolz_tables_enc(B, L, b, nullptr);
#ifdef CR_LZ3_PROF
        u64* const st = B.stats ? B.stats + ((u64)B.nblocks + b) * 16u : nullptr;
#else
        u64* const st = nullptr;
#endif
        const uint32_t mark = cr_rolz_rings_block_lds64(S, s_groups, B.in + B.in_off[b], n, B.flexible != 0u, T, st);
        if (mark && threadIdx.x == 0) B.pre_done[b] = (uint8_t)mark;
        __syncthreads();
    }
}

__global__ __launch_bounds__(CRGPU_WAVE) void k_rolz_encode(CrBatch B, CrArenaLayout L) {
    __shared__ CrRoxShared sh;
    uint8_t* arena = B.arena + (u64)blockIdx.x * L.stride;
    for (;;) {
        uint32_t t = 0;
        if (threadIdx.x == 0) t = atomicAdd(B.ticket, 1u);
        const uint32_t b = cr_uni(t);
        if (b >= B.nblocks) break;
        const uint32_t n = B.in_size[b];
        uint32_t r = 0xFFFFFFFFu;
        if (n <= L.max_block) {
            CrRolzTables T = cr_rolz_tables_enc(B, L, b, nullptr);
            r = cr_rolz_encode_block(B.in + B.in_off[b], n, B.out + B.out_off[b], T, arena + L.off_side, arena, L, B.fresh, B.persist, sh);
        }
        if (threadIdx.x == 0) B.out_size[b] = r;
        cr_wave_sync();
    }
}

__global__ __launch_bounds__(CRGPU_WAVE) void k_rolz_decode(CrBatch B, CrArenaLayout L) {
    __shared__ CrRoxShared sh;
    __shared__ uint32_t s_rows[256];
    uint8_t* arena = B.arena + (u64)blockIdx.x * L.stride;
    for (;;) {
        uint32_t t = 0;
        if (threadIdx.x == 0) t = atomicAdd(B.ticket, 1u);
        const uint32_t b = cr_uni(t);
        if (b >= B.nblocks) break;
        CrRolzTables T;
        T.ring_prev = reinterpret_cast<uint32_t*>(arena + L.off_cand);          /* u32[3][max_block]: two of the three */
        T.row_prev = T.ring_prev + L.max_block;
        T.rank = nullptr; T.len = nullptr; T.ring16 = nullptr;
        T.ring_head = reinterpret_cast<uint32_t*>(arena + L.off_rolz_head);
        uint32_t r = cr_rolz_decode_block(B.in + B.in_off[b], B.in_size[b], B.out + B.out_off[b], B.out_cap[b], T, s_rows, arena, L, B.fresh, B.persist, sh);
        if (threadIdx.x == 0) B.out_size[b] = r;
        cr_wave_sync();
    }
}

/* comprolz encoder on the comprop kernel pipeline (crgpu_rolz2.h) */
__global__ __launch_bounds__(CRGPU_WAVE) void k_rolz_events(CrBatch B, CrArenaLayout L) {
    __shared__ CrRoxShared sh;
    CR_TICKET_LOOP(2, {
        const uint32_t n = B.in_size[b];
        CrEvViews V = cr_ev_views(B.ev + (u64)b * B.ev_stride, B.ev_cap);
        if (n <= L.max_block) {
            CrRolzTables T = cr_rolz_tables_enc(B, L, b, nullptr);
            cr_rolz_emit_events(B.in + B.in_off[b], n, T, B.side + (u64)b * 3u * L.side_stride, V, sh);
        } else if (threadIdx.x == 0) { V.ctr[0] = 0; V.ctr[1] = 0; V.ctr[2] = 0; V.ctr[3] = 0x200u; }
    })
}

__global__ __launch_bounds__(CRGPU_WAVE) void k_rolz_rc(CrBatch B, CrArenaLayout L) {
    __shared__ u64 s_ring[CR_RC_RING];
    CR_TICKET_LOOP(6, {
        CrEvViews V = cr_ev_views(B.ev + (u64)b * B.ev_stride, B.ev_cap);
        uint32_t r = 0xFFFFFFFFu;
        if (!(V.ctr[3] & 0x200u)) r = cr_rolz_finish(B.in + B.in_off[b], B.in_size[b], B.out + B.out_off[b], B.side + (u64)b * 3u * L.side_stride, V, s_ring);
        if (threadIdx.x == 0) B.out_size[b] = r;
    })
}

/* same contract, the PPM main stream in assembly (crgpu_rolz5.h); fresh models per block only */
__global__ __launch_bounds__(CRGPU_WAVE) void k_rolz_decode_v5(CrBatch B, CrArenaLayout L) {
    __builtin_amdgcn_s_setprio(3);                      /* a dependent chain: its instructions go first when another stream's kernels share the SIMD */
    __shared__ CrRoxShared sh;
    __shared__ uint32_t s_rows[256];
    CR_V5_LDS_SCRATCH_N(s_px, CR_V5_SIDE_LDS_WORDS);
    uint8_t* arena = B.arena + (u64)blockIdx.x * L.stride;
    for (;;) {
        uint32_t t = 0;
        if (threadIdx.x == 0) t = atomicAdd(B.ticket, 1u);
        const uint32_t b = cr_uni(t);
        if (b >= B.nblocks) break;
        CrRolzTables T;
        T.ring_prev = reinterpret_cast<uint32_t*>(arena + L.off_cand);
        T.row_prev = T.ring_prev + L.max_block;
        T.rank = nullptr; T.len = nullptr; T.ring16 = nullptr;
        T.ring_head = reinterpret_cast<uint32_t*>(arena + L.off_rolz_head);
        uint32_t r = cr_rolz_decode_v5(B.in + B.in_off[b], B.in_size[b], B.out + B.out_off[b], B.out_cap[b], T, s_rows, arena, L, sh,
                                       L.off_hist ? reinterpret_cast<uint32_t*>(arena + L.off_hist) : nullptr, s_px_at,
                                       B.stats ? B.stats + (u64)b * 16u : nullptr);
        if (threadIdx.x == 0) B.out_size[b] = r;
        cr_wave_sync();
    }
}

/* static-dictionary stage: dictionary_encode / dictionary_decode per datablock (1 wave per block) */
struct CrDictBatch {
    CrDict          dict;
    uint8_t*        tmp;          /* encode: per-workgroup scratch of tmp_stride bytes */
    u64             tmp_stride;
    uint32_t*       match;        /* encode: what the trie says about every position, block b at match + b * match_stride */
    u64             match_stride; /* in entries */
    uint32_t        max_block;
};

/* the trie's answer for every position of every block (crgpu_dict.h). Only about one position in six can start a word,
 * and a wave is as slow as its longest walk, so a workgroup first collects the word starts of its 2 048 positions in LDS
 * (everything else gets its 0 right away) and then walks them with full waves. (1 024 / 2 048 / 4 096 / 8 192 positions per
 * workgroup: 0.86 / 0.79 / 0.78 / 1.18 ms on the bench shard, tools/dm_chunk_exp.sh.) */
#ifndef CR_DM_CHUNK
#define CR_DM_CHUNK 2048u
#endif
__global__ __launch_bounds__(256) void k_dict_match(CrBatch B, CrDictBatch DB) {
    __shared__ uint32_t s_start[CR_DM_CHUNK];
    __shared__ uint32_t s_count;
    const uint32_t b = blockIdx.x;
    const uint32_t n = B.in_size[b];
    const uint32_t base = blockIdx.y * CR_DM_CHUNK;
    if (n > DB.max_block || base >= n) return;
    const uint8_t* src = B.in + B.in_off[b];
    uint32_t* m = DB.match + (u64)b * DB.match_stride;
    const uint32_t end = base + CR_DM_CHUNK < n ? base + CR_DM_CHUNK : n;
    if (threadIdx.x == 0) s_count = 0;
    __syncthreads();
    for (uint32_t p0 = base; p0 < end; p0 += blockDim.x) {
        const uint32_t p = p0 + threadIdx.x;
        bool cand = false;
        if (p < end) {
            /* pieces of 1 000 000 bytes are coded on their own (cr-diccode.c:176-206): positions are piece-relative */
            const uint32_t q = p / CR_DIC_PIECE, first = q * CR_DIC_PIECE;
            const uint32_t psize = n - first < CR_DIC_PIECE ? n - first : CR_DIC_PIECE;
            cand = cr_dict_word_start(src + first, psize, p - first);
            if (!cand) m[p] = 0u;
        }
        const u64 mask = cr_ballot(cand);                        /* one LDS atomic per wave */
        uint32_t at = 0;
        if (cr_lane() == 0 && mask) at = atomicAdd(&s_count, (uint32_t)__builtin_popcountll(mask));
        at = cr_uni(at);
        if (cand) s_start[at + (uint32_t)__builtin_popcountll(mask & ((1ull << cr_lane()) - 1ull))] = p;
    }
    __syncthreads();
    const uint32_t count = s_count;
    for (uint32_t i = threadIdx.x; i < count; i += blockDim.x) {
        const uint32_t p = s_start[i];
        const uint32_t q = p / CR_DIC_PIECE, first = q * CR_DIC_PIECE;
        const uint32_t psize = n - first < CR_DIC_PIECE ? n - first : CR_DIC_PIECE;
        m[p] = cr_dict_match_at(DB.dict, src + first, psize, p - first);
    }
}

__global__ __launch_bounds__(CRGPU_WAVE) void k_dict_encode(CrBatch B, CrDictBatch DB) {
    __shared__ CrDictShared sh;
    uint8_t* tmp = DB.tmp + (u64)blockIdx.x * DB.tmp_stride;
    for (;;) {
        uint32_t t = 0;
        if (threadIdx.x == 0) t = atomicAdd(B.ticket, 1u);
        const uint32_t b = cr_uni(t);
        if (b >= B.nblocks) break;
        const uint32_t n = B.in_size[b];
        uint32_t r = n > DB.max_block ? 0xFFFFFFFFu
                   : cr_dict_encode_block(DB.dict, sh, B.in + B.in_off[b], n, DB.match + (u64)b * DB.match_stride, B.out + B.out_off[b], tmp);
        if (threadIdx.x == 0) B.out_size[b] = r;
        cr_wave_sync();
    }
}

__global__ __launch_bounds__(CRGPU_WAVE) void k_dict_decode(CrBatch B, CrDictBatch DB) {
    __shared__ CrDictShared sh;
    __shared__ __attribute__((aligned(16))) uint8_t s_ring[CR_DD_RING];
    for (;;) {
        uint32_t t = 0;
        if (threadIdx.x == 0) t = atomicAdd(B.ticket, 1u);
        const uint32_t b = cr_uni(t);
        if (b >= B.nblocks) break;
        uint32_t r = cr_dict_decode_block(DB.dict, sh, B.in + B.in_off[b], B.in_size[b], B.out + B.out_off[b], B.out_cap[b], s_ring);
        if (threadIdx.x == 0) B.out_size[b] = r;
        cr_wave_sync();
    }
}

/* k_pack: the container writer's concatenation (src/main.c:198-205: blocks go out one after the other, each behind
 * its packed {u32 size, u8 filt, u8 prec} header, empty blocks not at all) done on the device, so that a batch leaves
 * the GPU as ONE contiguous run. Two launches: an exclusive scan of the slot sizes by one workgroup, then a copy. */
#define CR_PACK_SCAN_THREADS 1024u
struct CrPack {
    const uint8_t*  in;
    const u64*      in_off;
    const uint32_t* in_size;
    const uint8_t*  filt;       /* per block m_filt, or NULL */
    uint32_t        nblocks;
    uint32_t        head;       /* 6: write the block headers, 0: payloads only */
    uint32_t        prec;       /* m_prec of every header */
    uint8_t*        out;
    u64*            out_off;    /* position of block b's PAYLOAD in out */
    u64*            total;      /* [0] bytes laid out, [1] number of failed blocks (size 0xFFFFFFFF) */
};

__global__ __launch_bounds__(CR_PACK_SCAN_THREADS) void k_pack_scan(CrPack P) {
    __shared__ u64 s_wave[CR_PACK_SCAN_THREADS / 64u];
    __shared__ u64 s_carry;
    __shared__ uint32_t s_bad;
    if (threadIdx.x == 0) { s_carry = 0; s_bad = 0; }
    __syncthreads();
    for (uint32_t base = 0; base < P.nblocks; base += CR_PACK_SCAN_THREADS) {
        const uint32_t b = base + threadIdx.x;
        uint32_t sz = b < P.nblocks ? P.in_size[b] : 0u;
        if (sz == 0xFFFFFFFFu) { atomicAdd(&s_bad, 1u); sz = 0; }
        const u64 w = sz ? (u64)sz + P.head : 0u;                  /* if(yb->m_size > 0), src/main.c:198 */
        /* inclusive scan over the wave (a slot is < 2^32 + 6: the low 16 bits and the rest are scanned apart, both sums
         * fit 32 bits), then the 16 wave totals through LDS */
        const u64 incl = ((u64)cr_scan_incl((uint32_t)(w >> 16)) << 16) + (u64)cr_scan_incl((uint32_t)(w & 0xffffu));
        if (cr_lane() == 63u) s_wave[cr_wave_id()] = incl;
        __syncthreads();
        u64 before = s_carry;
        for (uint32_t k = 0; k < cr_wave_id(); k++) before += s_wave[k];
        if (b < P.nblocks) P.out_off[b] = before + incl - w + (sz ? P.head : 0u);
        __syncthreads();
        if (threadIdx.x == CR_PACK_SCAN_THREADS - 1u) s_carry = before + incl;
        __syncthreads();
    }
    if (threadIdx.x == 0) { P.total[0] = s_carry; P.total[1] = s_bad; }
}

__global__ __launch_bounds__(256) void k_pack_copy(CrPack P) {
    for (uint32_t b = blockIdx.x; b < P.nblocks; b += gridDim.x) {
        const uint32_t sz = P.in_size[b];
        if (sz == 0u || sz == 0xFFFFFFFFu) continue;
        const uint8_t* src = P.in + P.in_off[b];
        uint8_t* dst = P.out + P.out_off[b];
        if (P.head && threadIdx.x < 6u) {                          /* packed {u32 m_size; u8 m_filt; u8 m_prec}, src/main.c:90-94 */
            const uint32_t t = threadIdx.x;
            dst[(int)t - 6] = t < 4u ? (uint8_t)(sz >> (8u * t)) : t == 4u ? (P.filt ? P.filt[b] : (uint8_t)0) : (uint8_t)P.prec;
        }
        /* bytes up to the first 16-byte boundary of dst, 16-byte pieces (source read unaligned), the rest */
        const uint32_t lead = (uint32_t)((16u - ((u64)(uintptr_t)dst & 15u)) & 15u);
        const uint32_t head = lead < sz ? lead : sz;
        if (threadIdx.x < head) dst[threadIdx.x] = src[threadIdx.x];
        const uint32_t body = (sz - head) / 16u;
        for (uint32_t i = threadIdx.x; i < body; i += blockDim.x)
            { uint4 v; __builtin_memcpy(&v, src + head + (u64)i * 16u, 16); *reinterpret_cast<uint4*>(dst + head + (u64)i * 16u) = v; }
        const uint32_t done = head + body * 16u;
        if (threadIdx.x < sz - done) dst[done + threadIdx.x] = src[done + threadIdx.x];
    }
}

/* size dictionary_decode() will produce for a dictionary-stage block (cr-diccode.c:208-217,359-360): raw + flag 0, or
 * groups of two pieces {u32 size1, u32 size2, piece1, piece2}, every piece ending with its u32 original size, then the
 * ten escape bytes and flag 1. One thread per block; 0xFFFFFFFF = malformed. */
__global__ __launch_bounds__(256) void k_dict_sizes(const uint8_t* in, const u64* in_off, const uint32_t* in_size, uint32_t nblocks, uint32_t* out) {
    const uint32_t b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= nblocks) return;
    const uint8_t* p = in + in_off[b];
    const uint32_t n = in_size[b];
    uint32_t r = 0xFFFFFFFFu;
    if (n != 0u && n != 0xFFFFFFFFu) {
        if (p[n - 1u] == 0u) r = n - 1u <= CRGPU_MAX_BLOCK ? n - 1u : 0xFFFFFFFFu;
        else if (n >= 11u) {
            u64 total = 0;
            u64 pos = 0;
            bool ok = true;
            while (pos + 11u < n) {
                if (pos + 8u > n) { ok = false; break; }
                const uint32_t a = *reinterpret_cast<const cr_u32u*>(p + pos), c = *reinterpret_cast<const cr_u32u*>(p + pos + 4u);
                pos += 8u;
                /* every piece ends with its u32 original size (cr-diccode.c:359-360), an empty one included */
                if (a < 4u || c < 4u || pos + a + c + 11u > n) { ok = false; break; }
                total += *reinterpret_cast<const cr_u32u*>(p + pos + a - 4u);
                total += *reinterpret_cast<const cr_u32u*>(p + pos + a + c - 4u);
                if (total > CRGPU_MAX_BLOCK) { ok = false; break; }        /* a crafted size must not drive the caller's allocation */
                pos += (u64)a + c;
            }
            if (ok) r = (uint32_t)total;
        }
    }
    out[b] = r;
}

/* self-test of the wave primitives (tests/ call this through crgpu_selftest) */
__global__ __launch_bounds__(CRGPU_WAVE) void k_selftest(const uint32_t* in, uint32_t* out) {
    uint32_t v = in[threadIdx.x];
    out[threadIdx.x] = cr_scan_incl(v);
    out[64 + threadIdx.x] = cr_sum(v);
    out[128 + threadIdx.x] = cr_bytesum(v);
    out[192 + threadIdx.x] = cr_mask_below(threadIdx.x, in[64]);
    out[256 + threadIdx.x] = (uint32_t)cr_prev_same(v & 7u, (threadIdx.x % 5u) != 0u);
    int a1 = cr_prev_same_shift(v & 0x3ffu, (threadIdx.x % 7u) != 0u), a2 = cr_prev_same_bits<16>(v & 0x3ffu, (threadIdx.x % 7u) != 0u);
    int a3 = cr_prev_same((v >> 3) & 0x3ffu, (threadIdx.x % 7u) != 0u), a4 = cr_prev_same_bits<24>((v >> 3) & 0x3ffu, (threadIdx.x % 7u) != 0u);
    out[384 + threadIdx.x] = (uint32_t)((a1 == a2 ? 0 : 1) | (a3 == a4 ? 0 : 2));
    out[320 + threadIdx.x] = cr_table_byte(v, in[65] & 255u);
}

/* ------------------------------------------------------------------ context */

struct crgpu_dict;

#define CRGPU_MAX_STAGES 16
struct crgpu_ctx {
    int         device;
    hipStream_t own_stream;
    hipStream_t stream;
    hipEvent_t  ev0, ev1;
    int         num_cu;
    int         wg_per_cu;
    /* per-workgroup table arenas, one per kind of work so that a process that only encodes (or only decodes) never
     * allocates the other's tables: [0] the kernel-pipeline encoders (match tables only, no models, never zeroed),
     * [1] the batched decoders (model tables + what their codec's matcher needs), [2] everything (the model-carrying
     * one-wave coders of the shims and of the diagnostic switches) */
    struct arena_set { uint8_t* p; size_t bytes; uint32_t wgs; CrArenaLayout L; uint32_t use; } arenas[3];
    uint32_t*   ticket;
    float       last_ms;
    u64*        stats;
    int         timed;
    char        err[256];
    /* host-pointer staging */
    uint8_t*    d_in;  size_t d_in_cap;
    uint8_t*    d_out; size_t d_out_cap;
    uint8_t*    d_meta; size_t d_meta_cap;
    uint8_t*    d_lens; size_t d_lens_cap;      /* encode: LZP lengths for the whole batch */
    uint8_t*    d_done; size_t d_done_cap;      /* encode: CrBatch::pre_done, one byte per block */
    uint32_t    done_blocks;                    /* blocks of the most recent encode launch (crgpu_last_prepass_paths) */
    int         lzp64_ready, rox64_ready, rolz64_ready;
    uint8_t*    d_rox; size_t d_rox_cap;        /* comprox encode: per-position match tables */
    uint8_t*    d_ev; size_t d_ev_cap;          /* comprop chain encoder: per-block event scratch */
    uint8_t*    d_side; size_t d_side_cap;      /* comprox chain encoder: per-block side-stream staging */
    int         one_wave_encoder;   /* CRGPU_OPT_ONE_WAVE_ENCODER: the model-carrying one-wave coders instead of the kernel pipeline */
    int         one_wave_decoder;   /* CRGPU_OPT_ONE_WAVE_DECODER: the model-carrying C++ decoders instead of the assembly step */
    uint32_t    lzp_grid, match_grid;   /* experiments: at most this many workgroups for the pre-pass kernels (0 = no limit) */
    int         decoder_helper;     /* CRGPU_OPT_DECODER_HELPER: comprop's batched decoder with a helper wave per block (k_rop_decode_v5h) */
    int         decoder_small_lds;  /* CRGPU_OPT_DECODER_LDS_NODES 0: k_rop_decode_v5s, no dense nodes in LDS */
    struct { const void* k; uint32_t dyn; int per_cu; } occ[48];   /* cr_resident_grid: workgroups per CU the runtime reports, per kernel */
    int         n_occ;
    int         lzp_tables_only;    /* CRGPU_OPT_LZP_TABLES: every block through the table sweep k_rop_lzp, none through k_rop_lzp_lds */
    int         lzp_lds_ready;      /* the LDS kernel's dynamic shared memory size has been raised */
    int         rolz_lds_ready, rox_lds_ready, links_lds_ready;
    uint32_t    rox_limit;
    int         flexible;       /* -f: flexible parsing for comprox / comprolz */
    int         persist;        /* shim context: one slot, models survive the call */
    int         next_fresh;     /* persist mode: reset_models() was called since the last block */
    hipEvent_t  ev_mid;
    float       last_lzp_ms;
    hipEvent_t  ev_stage[CRGPU_MAX_STAGES + 1];   /* boundaries of the kernels of the last call */
    int         n_stages;
    const char* stage_name[CRGPU_MAX_STAGES];
    /* CRGPU_OPT_STAGE_LOG: every call takes fresh boundary events from a pool and appends its kernels to a log that
     * crgpu_stage_log_read folds up later — nobody has to wait for an event between the calls of a timed loop */
    hipEvent_t  ev_own[CRGPU_MAX_STAGES + 1];
    int         log_on;
    hipEvent_t* pool; int pool_n, pool_used;
    struct stage_rec { const char* name; hipEvent_t a, b; }* log;
    int         log_n, log_cap;
};

static int stage_begin(crgpu_ctx* c) {
    if (!c->log_on) return CRGPU_OK;
    if (c->pool_used + CRGPU_MAX_STAGES + 1 > c->pool_n) {
        const int want = c->pool_n ? c->pool_n * 2 : 64 * (CRGPU_MAX_STAGES + 1);
        hipEvent_t* np = (hipEvent_t*)realloc(c->pool, sizeof(hipEvent_t) * (size_t)want);
        if (!np) return CRGPU_E_NOMEM;
        c->pool = np;
        for (; c->pool_n < want; c->pool_n++) if (hipEventCreate(&c->pool[c->pool_n]) != hipSuccess) return CRGPU_E_NODEVICE;
    }
    for (int i = 0; i <= CRGPU_MAX_STAGES; i++) c->ev_stage[i] = c->pool[c->pool_used++];
    return CRGPU_OK;
}

static int stage_end(crgpu_ctx* c) {
    if (!c->log_on) return CRGPU_OK;
    if (c->log_n + c->n_stages > c->log_cap) {
        const int want = c->log_cap ? c->log_cap * 2 : 1024;
        crgpu_ctx::stage_rec* nl = (crgpu_ctx::stage_rec*)realloc(c->log, sizeof *nl * (size_t)want);
        if (!nl) return CRGPU_E_NOMEM;
        c->log = nl; c->log_cap = want;
    }
    for (int i = 0; i < c->n_stages; i++) { c->log[c->log_n].name = c->stage_name[i]; c->log[c->log_n].a = c->ev_stage[i]; c->log[c->log_n].b = c->ev_stage[i + 1]; c->log_n++; }
    return CRGPU_OK;
}

static int fail(crgpu_ctx* c, hipError_t e, const char* what) {
    if (c) snprintf(c->err, sizeof c->err, "%s: %s", what, hipGetErrorString(e));
    return CRGPU_E_NODEVICE;
}
#define CR_TRY(c, call) do { hipError_t e_ = (call); if (e_ != hipSuccess) return fail(c, e_, #call); } while (0)

static u64 align_up(u64 v, u64 a) { return (v + a - 1) / a * a; }
static u64 cr_ev_slot_bytes_host(uint32_t cap) { return 64ull + (u64)cap * (8u + 32u + 4u * 5u + 8u + 2u * 3u + 2u) + ((u64)cap / 1024u + 2u) * 1024u + 512u; }

static uint32_t pow2_at_least(u64 want, uint32_t lo, uint32_t hi) {
    uint32_t c = lo;
    while (c < want && c < hi) c <<= 1;
    return c;
}

/* which tables a workgroup's arena holds */
#define CR_USE_MODEL     1u     /* directory, order-2 nodes, order-1 rows, direct order-3 table: the fixed head of crgpu_device.h */
#define CR_USE_O3HASH    2u     /* the one-wave encoder's order-3 hash table */
#define CR_USE_LZP       4u     /* lzp8 / lzp4 hash tables */
#define CR_USE_LZ2       8u     /* lzp2 table; also k_rop_links' sort scratch */
#define CR_USE_LENS     16u
#define CR_USE_CAND     32u     /* u32[3][max_block]: LZP candidates / comprolz decoder links */
#define CR_USE_HIST     64u     /* comprolz decoder: ring history */
#define CR_USE_ROX     128u     /* comprox match sweep: class heads, short-cache heads */
#define CR_USE_RHEAD   256u     /* comprolz ring heads */
#define CR_USE_KEEP    512u     /* parked state + side-stream staging of the one-wave coders */
#define CR_USE_ALL    1023u
#define CR_USE_DMODEL 2048u     /* the batched decoders' model tables: directory, node lines, order-1 rows, direct order-3 table, dense slots */
enum { CR_AR_ENC = 0, CR_AR_DEC = 1, CR_AR_ALL = 2 };

static CrArenaLayout make_layout(uint32_t max_block, uint32_t use) {
    CrArenaLayout L;
    memset(&L, 0, sizeof L);
    L.max_block = max_block;
    L.max_nodes = 65536u;                                   /* direct-indexed by the 16-bit context */
    L.cap_o3 = pow2_at_least(2u * (u64)max_block, 1024u, 1u << 23);
    L.cap_lz = pow2_at_least(2u * (u64)max_block, 1024u, 1u << 26);
    L.cap_lz2 = 65536u;
    u64 o = 0;
#define CR_REGION(field_, bit_, bytes_) do { L.field_ = o; if (use & (bit_)) o = align_up(o + (u64)(bytes_), 256); } while (0)
    /* the decoder's hot tables sit at offsets that do not depend on the block size (crgpu_rop5.h uses them as immediates) */
    CR_REGION(off_dir, CR_USE_MODEL | CR_USE_DMODEL, 65536ull * 4u);
    L.node_area = (use & CR_USE_DMODEL) ? (u64)CRGPU_LINE_AREA : (u64)CRGPU_NODE_AREA;
    CR_REGION(off_nodes, CR_USE_MODEL | CR_USE_DMODEL, L.node_area);
    CR_REGION(off_o1, CR_USE_MODEL | CR_USE_DMODEL, 65536ull);
    CR_REGION(off_o3d, CR_USE_MODEL | CR_USE_DMODEL, (u64)CR_O3D_ENTRIES * 2u);
    /* a node leaves its line with its 63rd symbol, every symbol of a node costs one coded escape, and a block of n bytes is
     * at most 2 n coding steps (an escape byte + a length symbol per token, damaged streams included): at most 2 n / 63
     * nodes of a block ever do */
    L.dense_slots = 2u * (max_block / (CRGPU_LINE_PAIRS + 1u)) + 4u;
    CR_REGION(off_dense, CR_USE_DMODEL, (u64)L.dense_slots * CRGPU_NODE_BYTES);
    CR_REGION(off_o3, CR_USE_O3HASH, (u64)L.cap_o3 * 8u);
    CR_REGION(off_lz8, CR_USE_LZP, (u64)L.cap_lz * 8u);
    CR_REGION(off_lz4, CR_USE_LZP, (u64)L.cap_lz * 8u);
    CR_REGION(off_lz2, CR_USE_LZ2, 65536ull * 4u);
    CR_REGION(off_lens, CR_USE_LENS, (u64)max_block + 256u);
    CR_REGION(off_cand, CR_USE_CAND, (u64)max_block * 12u);
    if ((use & CR_USE_HIST) && max_block <= (1u << 20)) { L.off_hist = o; o = align_up(o + (u64)max_block * 32u, 256); }   /* (1 GB per arena at the largest block size otherwise) */
    CR_REGION(off_rox_cls, CR_USE_ROX, (u64)20u * (20u + max_block / 25u) * 4u + 64u);
    CR_REGION(off_rox_near, CR_USE_ROX, 65536ull * 4u);
    CR_REGION(off_rolz_head, CR_USE_RHEAD, (u64)CR_ROLZ_BUCKETS * 4u);
    CR_REGION(off_keep, CR_USE_KEEP, 8192u);
    L.side_stride = align_up((u64)max_block * 2u + 256u, 256);
    CR_REGION(off_side, CR_USE_KEEP, 3u * L.side_stride);
#undef CR_REGION
    L.stride = align_up(o ? o : 256u, 4096);
    /* invariants of the layout, not run-time conditions: the head of the arena sits at the offsets the assembly uses as
     * immediates, and crgpu_rop5.h addresses the LZP tables with 32-bit arena offsets (true up to CRGPU_MAX_BLOCK + 1).
     * A layout that breaks them is a programming error; it is reported as "no layout" (stride 0 -> CRGPU_E_NOMEM). */
    if ((use & CR_USE_MODEL) && (L.off_dir != CRGPU_OFF_DIR || L.off_nodes != CRGPU_OFF_NODES || L.off_o1 != CRGPU_OFF_O1 || L.off_o3d != CRGPU_OFF_O3D)) L.stride = 0;
    if ((use & CR_USE_DMODEL) && ((use & CR_USE_MODEL) || L.off_dir != CRGPU_OFF_DIR || L.off_nodes != CRGPU_OFF_NODES || L.off_o1 != CRGPU_DEC_OFF_O1 ||
                                  L.off_o3d != CRGPU_DEC_OFF_O3D || L.off_dense + (u64)L.dense_slots * CRGPU_NODE_BYTES > 0xFFFFFFFFull)) L.stride = 0;
    if (L.off_lz2 + 65536ull * 4u > 0xFFFFFFFFull) L.stride = 0;
    return L;
}

extern "C" uint32_t crgpu_bound(int codec, uint32_t n) {
    /* comprox / comprolz only test their MAIN stream against the input size, so header + streams can exceed
     * n + header (an empty comprox block codes to 52 bytes): room for the side streams, derived in crgpu_device.h */
    if (codec == CRGPU_CODEC_ROX) return cr_bound_rox(n);
    if (codec == CRGPU_CODEC_ROLZ) return cr_bound_rolz(n);
    return n + CRGPU_ROP_HEADER;
}

static bool create_stage_events(crgpu_ctx* c) {
    for (int i = 0; i <= CRGPU_MAX_STAGES; i++) { if (hipEventCreate(&c->ev_own[i]) != hipSuccess) return false; c->ev_stage[i] = c->ev_own[i]; }
    return true;
}

extern "C" int crgpu_create(crgpu_ctx** out, int device) {
    if (!out) return CRGPU_E_ARG;
    *out = NULL;
    int count = 0;
    if (hipGetDeviceCount(&count) != hipSuccess || device < 0 || device >= count) return CRGPU_E_NODEVICE;
    crgpu_ctx* c = (crgpu_ctx*)calloc(1, sizeof *c);
    if (!c) return CRGPU_E_NOMEM;
    c->device = device;
    hipDeviceProp_t prop;
    if (hipSetDevice(device) != hipSuccess || hipGetDeviceProperties(&prop, device) != hipSuccess) { free(c); return CRGPU_E_NODEVICE; }
    if (strncmp(prop.gcnArchName, "gfx950", 6) != 0) {      /* the code object only holds gfx950 */
        free(c);
        return CRGPU_E_NODEVICE;
    }
    c->num_cu = prop.multiProcessorCount;
    /* resident workgroups per CU (each owns a 34 MB model arena): batches larger than 256 x this are worked off in
     * rounds. The block decoders are latency-bound chains, so their throughput on big batches is the number of
     * chains in flight: 1e9 B (15 259 blocks) decode in 476 / 310 / 278 ms with 8 / 16 / 24 per CU. 16 = 140 GB
     * of arena on a 288 GB card at most (ensure_arena shrinks it to what is free). */
    c->wg_per_cu = 16;
    if (hipStreamCreateWithFlags(&c->own_stream, hipStreamNonBlocking) != hipSuccess ||
        hipEventCreate(&c->ev0) != hipSuccess || hipEventCreate(&c->ev1) != hipSuccess || hipEventCreate(&c->ev_mid) != hipSuccess || !create_stage_events(c) ||
        hipMalloc((void**)&c->ticket, 256) != hipSuccess) {
        free(c);
        return CRGPU_E_NODEVICE;
    }
    c->stream = c->own_stream;
    c->rox_limit = CR_ROX_LIMIT;
    /* diagnostic switches: the environment is read HERE, once; crgpu_set_option changes them on a live context */
    static const struct { const char* env; int opt; } k_env[] = {
        {"CRGPU_WG_PER_CU", CRGPU_OPT_WG_PER_CU}, {"CRGPU_ONE_WAVE_ENCODER", CRGPU_OPT_ONE_WAVE_ENCODER},
        {"CRGPU_ONE_WAVE_DECODER", CRGPU_OPT_ONE_WAVE_DECODER}, {"CRGPU_LZP_GRID", CRGPU_OPT_LZP_GRID}, {"CRGPU_MATCH_GRID", CRGPU_OPT_MATCH_GRID},
        {"CRGPU_LZP_TABLES", CRGPU_OPT_LZP_TABLES}, {"CRGPU_DECODER_HELPER", CRGPU_OPT_DECODER_HELPER},
        {"CRGPU_DECODER_LDS_NODES", CRGPU_OPT_DECODER_LDS_NODES}};
    for (size_t i = 0; i < sizeof k_env / sizeof k_env[0]; i++) {
        const char* e = getenv(k_env[i].env);
        if (e && *e) (void)crgpu_set_option(c, k_env[i].opt, atoi(e));
    }
    *out = c;
    return CRGPU_OK;
}

extern "C" int crgpu_set_option(crgpu_ctx* c, int option, int value) {
    if (!c || value < 0) return CRGPU_E_ARG;
    switch (option) {
        case CRGPU_OPT_WG_PER_CU:        c->wg_per_cu = value < 1 ? 1 : value > 32 ? 32 : value; return CRGPU_OK;
        case CRGPU_OPT_ONE_WAVE_ENCODER: c->one_wave_encoder = value != 0; return CRGPU_OK;
        case CRGPU_OPT_ONE_WAVE_DECODER: c->one_wave_decoder = value != 0; return CRGPU_OK;
        case CRGPU_OPT_LZP_GRID:         c->lzp_grid = (uint32_t)value; return CRGPU_OK;
        case CRGPU_OPT_MATCH_GRID:       c->match_grid = (uint32_t)value; return CRGPU_OK;
        case CRGPU_OPT_LZP_TABLES:       c->lzp_tables_only = value != 0; return CRGPU_OK;
        case CRGPU_OPT_DECODER_HELPER:   c->decoder_helper = value != 0; return CRGPU_OK;
        case CRGPU_OPT_DECODER_LDS_NODES: c->decoder_small_lds = value == 0; return CRGPU_OK;
        case CRGPU_OPT_STAGE_LOG:
            /* events of the pool may still be pending on the stream: let them pass before the pool is handed out again */
            if (c->pool_used && (hipSetDevice(c->device) != hipSuccess || hipStreamSynchronize(c->stream) != hipSuccess)) return CRGPU_E_NODEVICE;
            c->log_on = value != 0; c->log_n = 0; c->pool_used = 0;
            for (int i = 0; i <= CRGPU_MAX_STAGES; i++) c->ev_stage[i] = c->ev_own[i];
            return CRGPU_OK;
    }
    return CRGPU_E_ARG;
}

extern "C" void crgpu_destroy(crgpu_ctx* c) {
    if (!c) return;
    (void)hipSetDevice(c->device);
    (void)hipStreamSynchronize(c->stream);
    for (int i = 0; i < 3; i++) (void)hipFree(c->arenas[i].p); (void)hipFree(c->ticket); (void)hipFree(c->d_in); (void)hipFree(c->d_out); (void)hipFree(c->d_meta); (void)hipFree(c->d_lens); (void)hipFree(c->d_done); (void)hipFree(c->d_rox); (void)hipFree(c->d_ev); (void)hipFree(c->d_side); (void)hipEventDestroy(c->ev_mid); for (int i = 0; i <= CRGPU_MAX_STAGES; i++) if (c->ev_own[i]) (void)hipEventDestroy(c->ev_own[i]);
    for (int i = 0; i < c->pool_n; i++) (void)hipEventDestroy(c->pool[i]);
    free(c->pool); free(c->log);
    (void)hipEventDestroy(c->ev0); (void)hipEventDestroy(c->ev1);
    (void)hipStreamDestroy(c->own_stream);
    free(c);
}

extern "C" const char* crgpu_last_error(const crgpu_ctx* c) { return c ? c->err : "no context"; }

extern "C" int crgpu_rox_set_chain_limit(crgpu_ctx* c, uint32_t limit) {
    if (!c || limit == 0) return CRGPU_E_ARG;
    c->rox_limit = limit;
    return CRGPU_OK;
}

extern "C" int crgpu_set_flexible_parsing(crgpu_ctx* c, int on) {
    if (!c) return CRGPU_E_ARG;
    c->flexible = on != 0;
    return CRGPU_OK;
}

extern "C" int crgpu_set_stream(crgpu_ctx* c, void* s) {
    if (!c) return CRGPU_E_ARG;
    c->stream = s ? (hipStream_t)s : c->own_stream;
    return CRGPU_OK;
}

/* milliseconds of the LZP pre-pass (k_rop_lzp) of the most recent encode call, -1 if none */
extern "C" float crgpu_last_lzp_ms(const crgpu_ctx* c) {
    if (!c || !c->timed) return -1.0f;
    float ms = -1.0f;
    if (hipEventSynchronize(c->ev1) != hipSuccess) return -1.0f;
    if (hipEventElapsedTime(&ms, c->ev0, c->ev_mid) != hipSuccess) return -1.0f;
    return ms;
}

extern "C" int crgpu_debug_stats(crgpu_ctx* c, uint64_t* dev_stats) {
    if (!c) return CRGPU_E_ARG;
    c->stats = (u64*)dev_stats;
    return CRGPU_OK;
}

extern "C" int crgpu_last_stage_ms(const crgpu_ctx* c, const char** names, float* ms, int room) {
    if (!c || !c->timed || room < 0) return -1;
    if (hipEventSynchronize(c->ev1) != hipSuccess) return -1;
    int n = c->n_stages < room ? c->n_stages : room;
    for (int i = 0; i < n; i++) {
        if (names) names[i] = c->stage_name[i];
        if (ms && hipEventElapsedTime(&ms[i], c->ev_stage[i], c->ev_stage[i + 1]) != hipSuccess) return -1;
    }
    return c->n_stages;
}

/* CRGPU_OPT_STAGE_LOG: per kernel name, the summed milliseconds and the number of launches since the log was switched
 * on / last read; waits for the stream once, then empties the log. Returns the number of distinct kernels. */
extern "C" int crgpu_stage_log_read(crgpu_ctx* c, const char** names, float* total_ms, uint32_t* launches, int room) {
    if (!c || !c->log_on || room < 0) return -1;
    int n = 0, bad = 0;
    if (hipSetDevice(c->device) != hipSuccess || hipStreamSynchronize(c->stream) != hipSuccess) bad = 1;
    for (int i = 0; i < c->log_n && !bad; i++) {
        float ms = 0.0f;
        if (hipEventElapsedTime(&ms, c->log[i].a, c->log[i].b) != hipSuccess) { bad = 1; break; }
        int k = 0;
        while (k < n && (k >= room || names[k] != c->log[i].name)) k++;     /* names beyond `room` are counted, not stored */
        if (k == n) {
            if (k < room) { names[k] = c->log[i].name; total_ms[k] = 0.0f; launches[k] = 0; }
            else {                                                           /* a name that did not fit: seen before? */
                int seen = 0;
                for (int j = 0; j < i && !seen; j++) seen = c->log[j].name == c->log[i].name;
                if (seen) continue;
            }
            n++;
        }
        if (k < room) { total_ms[k] += ms; launches[k]++; }
    }
    c->log_n = 0; c->pool_used = 0;             /* the log is emptied whatever happened: a failed read must not poison the next one */
    return bad ? -1 : n;                        /* the true number of distinct kernels: more than `room` = the caller's arrays were too short */
}

/* which pre-pass took the blocks of the most recent encode launch: counts[0] the table sweep, [1] the LDS kernel for blocks
 * of up to 28 672 bytes, [2] the LDS kernel for blocks of up to 65 537 bytes. Waits for the stream. */
extern "C" int crgpu_last_prepass_paths(crgpu_ctx* c, uint32_t counts[3]) {
    if (!c || !counts) return CRGPU_E_ARG;
    counts[0] = counts[1] = counts[2] = 0;
    if (!c->d_done || c->done_blocks == 0) return CRGPU_OK;
    uint8_t* h = (uint8_t*)malloc(c->done_blocks);
    if (!h) return CRGPU_E_NOMEM;
    if (hipSetDevice(c->device) != hipSuccess || hipStreamSynchronize(c->stream) != hipSuccess ||
        hipMemcpy(h, c->d_done, c->done_blocks, hipMemcpyDeviceToHost) != hipSuccess) { free(h); return CRGPU_E_NODEVICE; }
    for (uint32_t i = 0; i < c->done_blocks; i++) counts[(h[i] & 3u) < 3u ? (h[i] & 3u) : 2u]++;      /* 3: comprolz, the 64 KiB kernel did the rings and left the rows to the sweep */
    free(h);
    return CRGPU_OK;
}

extern "C" float crgpu_last_kernel_ms(const crgpu_ctx* c) {
    if (!c || !c->timed) return -1.0f;
    float ms = -1.0f;
    if (hipEventSynchronize(c->ev1) != hipSuccess) return -1.0f;
    if (hipEventElapsedTime(&ms, c->ev0, c->ev1) != hipSuccess) return -1.0f;
    return ms;
}

/* make sure arena set `which` can serve `wgs` resident workgroups of blocks up to max_block bytes with the tables of `use` */
static int ensure_arena(crgpu_ctx* c, int which, uint32_t max_block, uint32_t wgs, uint32_t use) {
    crgpu_ctx::arena_set* A = &c->arenas[which];
    if (max_block < 1024u) max_block = 1024u;
    max_block = (uint32_t)align_up(max_block, 1024u);
    if (A->p && A->L.max_block >= max_block && A->wgs >= wgs && (A->use & use) == use) return CRGPU_OK;
    if (A->p) { use |= A->use; if (A->L.max_block > max_block) max_block = A->L.max_block; if (A->wgs > wgs) wgs = A->wgs; }
    CrArenaLayout L = make_layout(max_block, use);
    if (L.stride == 0) { snprintf(c->err, sizeof c->err, "internal: arena layout violates its invariants"); return CRGPU_E_NOMEM; }
    size_t free_b = 0, total_b = 0;
    CR_TRY(c, hipStreamSynchronize(c->stream));
    if (A->p) { (void)hipFree(A->p); A->p = NULL; A->wgs = 0; A->bytes = 0; }
    CR_TRY(c, hipMemGetInfo(&free_b, &total_b));
    u64 budget = (u64)(free_b * 0.85);
    while (wgs > 1 && (u64)wgs * L.stride > budget) wgs--;
    if ((u64)wgs * L.stride > budget) { snprintf(c->err, sizeof c->err, "arena does not fit device memory"); return CRGPU_E_NOMEM; }
    if (hipMalloc((void**)&A->p, (size_t)((u64)wgs * L.stride)) != hipSuccess) {
        A->p = NULL;
        snprintf(c->err, sizeof c->err, "hipMalloc(arena %llu bytes) failed", (unsigned long long)((u64)wgs * L.stride));
        return CRGPU_E_NOMEM;
    }
    /* model tables are zeroed once: generation words start at 0 and every node tag is stale. The encoders' match tables
     * are laid out afresh by their kernels for every block: nothing to zero. */
    if ((use & (CR_USE_MODEL | CR_USE_DMODEL)) && hipMemsetAsync(A->p, 0, (size_t)((u64)wgs * L.stride), c->stream) != hipSuccess) return CRGPU_E_NODEVICE;
    A->bytes = (size_t)((u64)wgs * L.stride);
    A->wgs = wgs;
    A->L = L;
    A->use = use;
    return CRGPU_OK;
}

static int grow(crgpu_ctx* c, uint8_t** p, size_t* cap, size_t want);

/* A ticket-loop kernel is launched with as many workgroups as can be RESIDENT, not more: the blocks are pulled from a counter, so
 * workgroups beyond the chip's capacity only start when others end and leave a tail (config 3's decoder: 16 per CU launched, 12
 * resident by its registers, 140 ms; 12 launched, 133 ms — profiles/r06m). Workgroups per CU from the runtime's occupancy query,
 * remembered per kernel. */
static uint32_t cr_resident_grid(crgpu_ctx* c, const void* kernel, uint32_t threads, uint32_t dyn_lds, uint32_t grid) {
    int per_cu = 0;                                                      /* (the memo lives in the context: a context belongs to one thread at a time) */
    for (int i = 0; i < c->n_occ; i++) if (c->occ[i].k == kernel && c->occ[i].dyn == dyn_lds) { per_cu = c->occ[i].per_cu; break; }
    if (per_cu == 0) {
        int n = 0;
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, kernel, (int)threads, dyn_lds) != hipSuccess || n < 1) { (void)hipGetLastError(); n = 1 << 20; }
        per_cu = n;
        if (getenv("CRGPU_DEBUG_GRID")) fprintf(stderr, "[crgpu] kernel %p: %u threads, %u B dynamic LDS -> %d workgroups per CU\n", kernel, threads, dyn_lds, n);
        if (c->n_occ < (int)(sizeof c->occ / sizeof c->occ[0])) { c->occ[c->n_occ].k = kernel; c->occ[c->n_occ].dyn = dyn_lds; c->occ[c->n_occ].per_cu = n; c->n_occ++; }
    }
    const u64 cap = (u64)per_cu * (u64)c->num_cu;
    return cap < grid ? (uint32_t)cap : grid;
}
#define CR_G(kernel_, threads_, dyn_) dim3(cr_resident_grid(c, reinterpret_cast<const void*>(kernel_), (threads_), (dyn_), grid))

static int launch(crgpu_ctx* c, int codec, int decode, CrBatch& B, uint32_t max_block, int sync) {
    if (codec != CRGPU_CODEC_ROP && codec != CRGPU_CODEC_ROX && codec != CRGPU_CODEC_ROLZ) { snprintf(c->err, sizeof c->err, "codec %d not available", codec); return CRGPU_E_ARG; }
    if (max_block > CRGPU_MAX_BLOCK + 1u) { snprintf(c->err, sizeof c->err, "block of %u bytes exceeds CRGPU_MAX_BLOCK + 1", max_block); return CRGPU_E_ARG; }
    CR_TRY(c, hipSetDevice(c->device));
    uint32_t want = (uint32_t)c->num_cu * (uint32_t)c->wg_per_cu;
    if (want > B.nblocks) want = B.nblocks;
    if (want == 0) return CRGPU_OK;
    /* which tables this call needs: the model-carrying one-wave coders everything (persist mode keeps its tables across
     * calls, so that slot is sized once for the largest block), the kernel pipeline its codec's match tables, the
     * batched decoders the models and their codec's matcher */
    const int pipeline_enc = !decode && !c->one_wave_encoder && !c->persist;
    const int batched_dec = decode && !c->one_wave_decoder && !c->persist;
    int which = CR_AR_ALL;
    uint32_t use = CR_USE_ALL;
    if (pipeline_enc) {
        which = CR_AR_ENC;
        use = CR_USE_LZ2 | (codec == CRGPU_CODEC_ROP ? CR_USE_LZP | CR_USE_CAND : codec == CRGPU_CODEC_ROX ? CR_USE_ROX : CR_USE_RHEAD);
    } else if (batched_dec) {
        which = CR_AR_DEC;
        use = CR_USE_DMODEL | (codec == CRGPU_CODEC_ROP ? CR_USE_LZP | CR_USE_LZ2 : codec == CRGPU_CODEC_ROLZ ? CR_USE_CAND | CR_USE_HIST | CR_USE_RHEAD : 0u);
    }
    int rc = ensure_arena(c, which, c->persist ? CRGPU_MAX_BLOCK + 1u : max_block, want, use);
    if (rc != CRGPU_OK) return rc;
    const crgpu_ctx::arena_set* A = &c->arenas[which];
    const CrArenaLayout& LY = A->L;
    uint32_t grid = want < A->wgs ? want : A->wgs;
    B.ticket = c->ticket;
    B.arena = A->p;
    B.fresh = 1;
    B.persist = 0;
    if (c->persist) {                      /* reference-signature shims: one block, model carried across calls */
        if (B.nblocks != 1) return CRGPU_E_ARG;
        B.persist = 1;
        B.fresh = c->next_fresh ? 1u : 0u;
        c->next_fresh = 0;
    }
    B.stats = c->stats;
    CR_TRY(c, hipMemsetAsync(c->ticket, 0, 64, c->stream));
    const int chains = !decode && codec == CRGPU_CODEC_ROP && !c->one_wave_encoder && !c->persist;
    const int rox_chains = !decode && (codec == CRGPU_CODEC_ROX || codec == CRGPU_CODEC_ROLZ) && !c->one_wave_encoder && !c->persist;
    const int old_decoder = c->persist || c->one_wave_decoder;
    if (chains || rox_chains) {
        B.ev_cap = (uint32_t)align_up((u64)(max_block < 1024u ? 1024u : max_block) + max_block / 64u + 128u, 64);
        B.ev_stride = align_up(cr_ev_slot_bytes_host(B.ev_cap), 256);
        rc = grow(c, &c->d_ev, &c->d_ev_cap, (size_t)(B.ev_stride * B.nblocks));
        if (rc != CRGPU_OK) return rc;
        B.ev = c->d_ev;
    }
    if (rox_chains) {
        rc = grow(c, &c->d_side, &c->d_side_cap, (size_t)(3u * LY.side_stride * B.nblocks));
        if (rc != CRGPU_OK) return rc;
        B.side = c->d_side;
    }
    if (!decode && (codec == CRGPU_CODEC_ROX || codec == CRGPU_CODEC_ROLZ)) {
        B.rox_stride = align_up(((u64)(max_block < 1024u ? 1024u : max_block) + 64u) * 16u, 1024);
        rc = grow(c, &c->d_rox, &c->d_rox_cap, (size_t)(B.rox_stride * B.nblocks));
        if (rc != CRGPU_OK) return rc;
        B.rox = c->d_rox;
        B.rox_limit = c->rox_limit;
        B.flexible = c->flexible ? 1u : 0u;
    } else if (!decode) {
        B.lens_stride = align_up(max_block < 1024u ? 1024u : max_block, 256);
        rc = grow(c, &c->d_lens, &c->d_lens_cap, (size_t)(B.lens_stride * B.nblocks));
        if (rc != CRGPU_OK) return rc;
        B.lens = c->d_lens;
    }
    if (!decode) {
        rc = grow(c, &c->d_done, &c->d_done_cap, (size_t)B.nblocks + 16u);
        if (rc != CRGPU_OK) return rc;
        B.pre_done = c->d_done;
        c->done_blocks = B.nblocks;
        CR_TRY(c, hipMemsetAsync(c->d_done, 0, (size_t)B.nblocks, c->stream));
    }
    const uint32_t match_grid = c->match_grid && c->match_grid < grid ? c->match_grid : grid;   /* experiment: fewer resident workgroups for the match kernels */
    c->n_stages = 0;
    rc = stage_begin(c);
    if (rc != CRGPU_OK) return rc;
#ifdef CR_DEC_OCC_EXP                                    /* diagnostic build: $CRGPU_DEC_LDS_PAD bytes of dynamic LDS per decoder wave cap its residency */
#define CR_DEC_PAD() (getenv("CRGPU_DEC_LDS_PAD") ? (unsigned)atoi(getenv("CRGPU_DEC_LDS_PAD")) : 0u)
#else
#define CR_DEC_PAD() 0
#endif
#define CR_STAGE(name_, ...) do { \
        if (c->n_stages >= CRGPU_MAX_STAGES) { snprintf(c->err, sizeof c->err, "internal: more than %d kernels in one call", CRGPU_MAX_STAGES); return CRGPU_E_ARG; } \
        CR_TRY(c, hipEventRecord(c->ev_stage[c->n_stages], c->stream)); \
        __VA_ARGS__; \
        c->stage_name[c->n_stages++] = name_; \
    } while (0)
    /* the event sorts shared by the three chain encoders: blocks of up to 28 672 events in LDS, the rest through global memory */
#define CR_LINKS_STAGES() do { \
        B.links_lds = 0; \
        B.o2_tickets = c->lzp_tables_only ? 1u : 0u; \
        if (!c->lzp_tables_only) { \
            if (!c->links_lds_ready) { \
                CR_TRY(c, hipFuncSetAttribute(reinterpret_cast<const void*>(k_rop_links_lds), hipFuncAttributeMaxDynamicSharedMemorySize, (int)CR_LZ2_LDS_BYTES)); \
                CR_TRY(c, hipFuncSetAttribute(reinterpret_cast<const void*>(k_rop_links_lds64), hipFuncAttributeMaxDynamicSharedMemorySize, (int)CR_LK4_LDS_BYTES)); \
                c->links_lds_ready = 1; \
            } \
            B.links_lds = 1; \
            const uint32_t lg_ = (uint32_t)c->num_cu < grid ? (uint32_t)c->num_cu : grid; \
            CR_STAGE("k_rop_links_lds", hipLaunchKernelGGL(k_rop_links_lds, dim3(lg_), dim3(CR_LZ2_THREADS), CR_LZ2_LDS_BYTES, c->stream, B, LY)); \
            CR_TRY(c, hipGetLastError()); \
            if (max_block > CR_LZ2_MAXN / 2u) {                  /* (a block can hold two events per byte) */ \
                CR_STAGE("k_rop_links_lds64", hipLaunchKernelGGL(k_rop_links_lds64, dim3(lg_), dim3(CR_LK4_THREADS), CR_LK4_LDS_BYTES, c->stream, B, LY)); \
                CR_TRY(c, hipGetLastError()); \
            } \
        } \
        CR_STAGE("k_rop_links", hipLaunchKernelGGL(k_rop_links, CR_G(k_rop_links, CR_SORT_THREADS, 0), dim3(CR_SORT_THREADS), 0, c->stream, B, LY)); \
    } while (0)
    CR_TRY(c, hipEventRecord(c->ev0, c->stream));
    if (codec == CRGPU_CODEC_ROLZ && decode) {
        if (old_decoder) CR_STAGE("k_rolz_decode", hipLaunchKernelGGL(k_rolz_decode, CR_G(k_rolz_decode, CRGPU_WAVE, 0), dim3(CRGPU_WAVE), 0, c->stream, B, LY));
        else CR_STAGE("k_rolz_decode_v5", hipLaunchKernelGGL(k_rolz_decode_v5, CR_G(k_rolz_decode_v5, CRGPU_WAVE, 0), dim3(CRGPU_WAVE), 0, c->stream, B, LY));
    } else if (codec == CRGPU_CODEC_ROLZ) {
        B.lzp_lds = 0;
        if (!c->lzp_tables_only) {                           /* blocks of up to 28 672 bytes: links by sorting in LDS, searches out of LDS */
            if (!c->rolz_lds_ready) {
                CR_TRY(c, hipFuncSetAttribute(reinterpret_cast<const void*>(k_rolz_match_lds), hipFuncAttributeMaxDynamicSharedMemorySize, (int)CR_LZ2_LDS_BYTES_FOR(CR_ROLZ3_THREADS / 64u)));
                c->rolz_lds_ready = 1;
            }
            B.lzp_lds = 1;
            const uint32_t lds_grid = (uint32_t)c->num_cu < grid ? (uint32_t)c->num_cu : grid;
            CR_STAGE("k_rolz_match_lds", hipLaunchKernelGGL(k_rolz_match_lds, dim3(lds_grid), dim3(CR_ROLZ3_THREADS), CR_LZ2_LDS_BYTES_FOR(CR_ROLZ3_THREADS / 64u), c->stream, B, LY));
            CR_TRY(c, hipGetLastError());
            if (max_block > CR_LZ2_MAXN) {
                if (!c->rolz64_ready) {
                    CR_TRY(c, hipFuncSetAttribute(reinterpret_cast<const void*>(k_rolz_rings_lds64), hipFuncAttributeMaxDynamicSharedMemorySize, (int)CR_LZ3_LDS_BYTES));
                    c->rolz64_ready = 1;
                }
                CR_STAGE("k_rolz_rings_lds64", hipLaunchKernelGGL(k_rolz_rings_lds64, dim3(lds_grid), dim3(CR_LZ2_THREADS), CR_LZ3_LDS_BYTES, c->stream, B, LY));
                CR_TRY(c, hipGetLastError());
            }
        }
        CR_STAGE("k_rolz_match", hipLaunchKernelGGL(k_rolz_match, dim3(match_grid), dim3(256), 0, c->stream, B, LY));
        CR_TRY(c, hipGetLastError());
        CR_TRY(c, hipEventRecord(c->ev_mid, c->stream));
        if (rox_chains) {
            CR_STAGE("k_rolz_events", hipLaunchKernelGGL(k_rolz_events, CR_G(k_rolz_events, CRGPU_WAVE, 0), dim3(CRGPU_WAVE), 0, c->stream, B, LY));
            CR_LINKS_STAGES();
            CR_STAGE("k_rop_o3", hipLaunchKernelGGL(k_rop_o3, CR_G(k_rop_o3, 256, 0), dim3(256), 0, c->stream, B, LY));
            CR_STAGE("k_rop_o2", hipLaunchKernelGGL(k_rop_o2, CR_G(k_rop_o2, CR_O2_THREADS, 0), dim3(CR_O2_THREADS), 0, c->stream, B, LY));
            CR_STAGE("k_rop_o1", hipLaunchKernelGGL(k_rop_o1, CR_G(k_rop_o1, CR_O1_THREADS, 0), dim3(CR_O1_THREADS), 0, c->stream, B, LY));
            CR_STAGE("k_rolz_rc", hipLaunchKernelGGL(k_rolz_rc, CR_G(k_rolz_rc, CRGPU_WAVE, 0), dim3(CRGPU_WAVE), 0, c->stream, B, LY));
        } else {
            CR_STAGE("k_rolz_encode", hipLaunchKernelGGL(k_rolz_encode, CR_G(k_rolz_encode, CRGPU_WAVE, 0), dim3(CRGPU_WAVE), 0, c->stream, B, LY));
        }
    } else if (codec == CRGPU_CODEC_ROX && decode) {
        if (old_decoder) CR_STAGE("k_rox_decode", hipLaunchKernelGGL(k_rox_decode, CR_G(k_rox_decode, CRGPU_WAVE, 0), dim3(CRGPU_WAVE), 0, c->stream, B, LY));
        else CR_STAGE("k_rox_decode_v5", hipLaunchKernelGGL(k_rox_decode_v5, CR_G(k_rox_decode_v5, CRGPU_WAVE, 0), dim3(CRGPU_WAVE), 0, c->stream, B, LY));
    } else if (codec == CRGPU_CODEC_ROX) {
        B.lzp_lds = 0;
        if (!c->lzp_tables_only) {                           /* blocks of up to 28 672 bytes: chain and short-cache links by sorting in LDS */
            if (!c->rox_lds_ready) {
                CR_TRY(c, hipFuncSetAttribute(reinterpret_cast<const void*>(k_rox_links_lds), hipFuncAttributeMaxDynamicSharedMemorySize, (int)CR_LZ2_LDS_BYTES));
                c->rox_lds_ready = 1;
            }
            B.lzp_lds = 1;
            const uint32_t lds_grid = (uint32_t)c->num_cu < grid ? (uint32_t)c->num_cu : grid;
            CR_STAGE("k_rox_links_lds", hipLaunchKernelGGL(k_rox_links_lds, dim3(lds_grid), dim3(CR_LZ2_THREADS), CR_LZ2_LDS_BYTES, c->stream, B, LY));
            CR_TRY(c, hipGetLastError());
            if (max_block > CR_LZ2_MAXN) {
                if (!c->rox64_ready) {
                    CR_TRY(c, hipFuncSetAttribute(reinterpret_cast<const void*>(k_rox_links_lds64), hipFuncAttributeMaxDynamicSharedMemorySize, (int)CR_LZ3_LDS_BYTES));
                    c->rox64_ready = 1;
                }
                CR_STAGE("k_rox_links_lds64", hipLaunchKernelGGL(k_rox_links_lds64, dim3(lds_grid), dim3(CR_LZ2_THREADS), CR_LZ3_LDS_BYTES, c->stream, B, LY));
                CR_TRY(c, hipGetLastError());
            }
        }
        CR_STAGE("k_rox_match", hipLaunchKernelGGL(k_rox_match, dim3(match_grid), dim3(256), 0, c->stream, B, LY));
        CR_TRY(c, hipGetLastError());
        CR_TRY(c, hipEventRecord(c->ev_mid, c->stream));
        if (rox_chains) {
            CR_STAGE("k_rox_events", hipLaunchKernelGGL(k_rox_events, CR_G(k_rox_events, 256, 0), dim3(256), 0, c->stream, B, LY));
            CR_LINKS_STAGES();
            CR_STAGE("k_rop_o3", hipLaunchKernelGGL(k_rop_o3, CR_G(k_rop_o3, 256, 0), dim3(256), 0, c->stream, B, LY));
            CR_STAGE("k_rop_o2", hipLaunchKernelGGL(k_rop_o2, CR_G(k_rop_o2, CR_O2_THREADS, 0), dim3(CR_O2_THREADS), 0, c->stream, B, LY));
            CR_STAGE("k_rop_o1", hipLaunchKernelGGL(k_rop_o1, CR_G(k_rop_o1, CR_O1_THREADS, 0), dim3(CR_O1_THREADS), 0, c->stream, B, LY));
            CR_STAGE("k_rox_rc", hipLaunchKernelGGL(k_rox_rc, CR_G(k_rox_rc, CRGPU_WAVE, 0), dim3(CRGPU_WAVE), 0, c->stream, B, LY));
        } else {
            CR_STAGE("k_rox_encode", hipLaunchKernelGGL(k_rox_encode, CR_G(k_rox_encode, CRGPU_WAVE, 0), dim3(CRGPU_WAVE), 0, c->stream, B, LY));
        }
    } else if (decode) {
        if (old_decoder) CR_STAGE("k_rop_decode", hipLaunchKernelGGL(k_rop_decode, CR_G(k_rop_decode, CRGPU_WAVE, 0), dim3(CRGPU_WAVE), 0, c->stream, B, LY));
        else if (c->decoder_helper) CR_STAGE("k_rop_decode_v5h", hipLaunchKernelGGL(k_rop_decode_v5h, CR_G(k_rop_decode_v5h, 2 * CRGPU_WAVE, CR_DEC_PAD()), dim3(2 * CRGPU_WAVE), CR_DEC_PAD(), c->stream, B, LY));
        else if (c->decoder_small_lds) CR_STAGE("k_rop_decode_v5s", hipLaunchKernelGGL(k_rop_decode_v5s, CR_G(k_rop_decode_v5s, CRGPU_WAVE, CR_DEC_PAD()), dim3(CRGPU_WAVE), CR_DEC_PAD(), c->stream, B, LY));
        else CR_STAGE("k_rop_decode_v5", hipLaunchKernelGGL(k_rop_decode_v5, CR_G(k_rop_decode_v5, CRGPU_WAVE, CR_DEC_PAD()), dim3(CRGPU_WAVE), CR_DEC_PAD(), c->stream, B, LY));
    } else {
        const uint32_t lzp_grid = c->lzp_grid && c->lzp_grid < grid ? c->lzp_grid : grid;   /* experiment: fewer resident workgroups keep the LZP tables in the Infinity Cache */
        B.lzp_lds = 0;
        if (!c->lzp_tables_only) {                           /* blocks of up to 28 672 bytes: sorted in LDS, one block per CU at a time */
            if (!c->lzp_lds_ready) {
                CR_TRY(c, hipFuncSetAttribute(reinterpret_cast<const void*>(k_rop_lzp_lds), hipFuncAttributeMaxDynamicSharedMemorySize, (int)CR_LZ2_LDS_BYTES));
                c->lzp_lds_ready = 1;
            }
            B.lzp_lds = 1;
            const uint32_t lds_grid = (uint32_t)c->num_cu < grid ? (uint32_t)c->num_cu : grid;
            CR_STAGE("k_rop_lzp_lds", hipLaunchKernelGGL(k_rop_lzp_lds, dim3(lds_grid), dim3(CR_LZ2_THREADS), CR_LZ2_LDS_BYTES, c->stream, B, LY));
            CR_TRY(c, hipGetLastError());
            if (max_block > CR_LZ2_MAXN) {                   /* blocks of up to 65 537 bytes: the same sort in groups by key (round 4) */
                if (!c->lzp64_ready) {
                    CR_TRY(c, hipFuncSetAttribute(reinterpret_cast<const void*>(k_rop_lzp_lds64), hipFuncAttributeMaxDynamicSharedMemorySize, (int)CR_LZ3_LDS_BYTES));
                    c->lzp64_ready = 1;
                }
                CR_STAGE("k_rop_lzp_lds64", hipLaunchKernelGGL(k_rop_lzp_lds64, dim3(lds_grid), dim3(CR_LZ2_THREADS), CR_LZ3_LDS_BYTES, c->stream, B, LY));
                CR_TRY(c, hipGetLastError());
            }
        }
        CR_STAGE("k_rop_lzp", hipLaunchKernelGGL(k_rop_lzp, dim3(lzp_grid), dim3(256), 0, c->stream, B, LY));
        CR_TRY(c, hipGetLastError());
        CR_TRY(c, hipEventRecord(c->ev_mid, c->stream));
        if (chains) {
            CR_STAGE("k_rop_events", hipLaunchKernelGGL(k_rop_events, CR_G(k_rop_events, CRGPU_WAVE, 0), dim3(CRGPU_WAVE), 0, c->stream, B, LY));
            CR_LINKS_STAGES();
            CR_STAGE("k_rop_o3", hipLaunchKernelGGL(k_rop_o3, CR_G(k_rop_o3, 256, 0), dim3(256), 0, c->stream, B, LY));
            CR_STAGE("k_rop_o2", hipLaunchKernelGGL(k_rop_o2, CR_G(k_rop_o2, CR_O2_THREADS, 0), dim3(CR_O2_THREADS), 0, c->stream, B, LY));
            CR_STAGE("k_rop_o1", hipLaunchKernelGGL(k_rop_o1, CR_G(k_rop_o1, CR_O1_THREADS, 0), dim3(CR_O1_THREADS), 0, c->stream, B, LY));
            CR_STAGE("k_rop_rc", hipLaunchKernelGGL(k_rop_rc, CR_G(k_rop_rc, CRGPU_WAVE, 0), dim3(CRGPU_WAVE), 0, c->stream, B, LY));
        } else {
            CR_STAGE("k_rop_encode", hipLaunchKernelGGL(k_rop_encode, CR_G(k_rop_encode, CRGPU_WAVE, 0), dim3(CRGPU_WAVE), 0, c->stream, B, LY));
        }
    }
#undef CR_LINKS_STAGES
#undef CR_STAGE
    CR_TRY(c, hipGetLastError());
    CR_TRY(c, hipEventRecord(c->ev_stage[c->n_stages], c->stream));
    CR_TRY(c, hipEventRecord(c->ev1, c->stream));
    c->timed = 1;
    rc = stage_end(c);
    if (rc != CRGPU_OK) return rc;
    if (sync) CR_TRY(c, hipStreamSynchronize(c->stream));
    return CRGPU_OK;
}

extern "C" int crgpu_encode_blocks_dev(crgpu_ctx* c, int codec, const uint8_t* in, const uint64_t* in_off,
                                       const uint32_t* in_size, uint32_t nblocks, uint32_t max_block,
                                       uint8_t* out, const uint64_t* out_off, uint32_t* out_size, int sync) {
    if (!c || (nblocks && (!in || !in_off || !in_size || !out || !out_off || !out_size))) return CRGPU_E_ARG;
    CrBatch B; memset(&B, 0, sizeof B);
    B.in = in; B.in_off = (const u64*)in_off; B.in_size = in_size;
    B.out = out; B.out_off = (const u64*)out_off; B.out_size = out_size; B.nblocks = nblocks;
    return launch(c, codec, 0, B, max_block, sync);
}

extern "C" int crgpu_decode_blocks_dev(crgpu_ctx* c, int codec, const uint8_t* in, const uint64_t* in_off,
                                       const uint32_t* in_size, uint32_t nblocks, uint32_t max_block,
                                       uint8_t* out, const uint64_t* out_off, const uint32_t* out_cap,
                                       uint32_t* out_size, int sync) {
    if (!c || (nblocks && (!in || !in_off || !in_size || !out || !out_off || !out_cap || !out_size))) return CRGPU_E_ARG;
    CrBatch B; memset(&B, 0, sizeof B);
    B.in = in; B.in_off = (const u64*)in_off; B.in_size = in_size;
    B.out = out; B.out_off = (const u64*)out_off; B.out_cap = out_cap; B.out_size = out_size; B.nblocks = nblocks;
    return launch(c, codec, 1, B, max_block, sync);
}

/* ------------------------------------------------------------------ static dictionary (device copy) */

struct crgpu_dict {
    crgpu_ctx* ctx;
    uint32_t*  d_next;
    int32_t*   d_ids;
    uint8_t*   d_words;
    uint8_t*   d_wlen;
    uint32_t   nwords, nnodes, trie_words;
    uint8_t*   d_tmp; size_t tmp_cap;
    uint8_t*   d_match; size_t match_cap;
};

/* dictionary_load(text, 1) — cr-diccode.c:76-118 — then flattened for the device */
extern "C" int crgpu_dict_create(crgpu_ctx* c, const char* text, crgpu_dict** out) {
    if (!c || !text || !out) return CRGPU_E_ARG;
    *out = NULL;
    const uint32_t MAXW = 25000u;                                  /* cr-diccode.h:39 */
    uint8_t* words = (uint8_t*)calloc(MAXW + 1, CR_DIC_WORD_STRIDE);
    uint8_t* wlen = (uint8_t*)calloc(MAXW + 1, 1);
    if (!words || !wlen) { free(words); free(wlen); return CRGPU_E_NOMEM; }
    uint32_t nw = 0, p = 0;
    for (size_t i = 0; text[i]; i++) {
        if (nw >= MAXW) { free(words); free(wlen); return CRGPU_E_ARG; }
        uint8_t* w = words + (size_t)nw * CR_DIC_WORD_STRIDE;
        if (text[i] == '\n') {
            if (p > 0 && p + 2 <= CR_DIC_WORD_STRIDE && ((uint32_t)((w[p - 1] | 0x20u) - 'a') < 26u)) { w[p++] = ' '; }   /* words ending in a letter own their space */
            wlen[nw] = (uint8_t)p;
            p = 0;
            nw++;
        } else if (p + 2 < CR_DIC_WORD_STRIDE) {
            w[p++] = (uint8_t)text[i];
        }
    }
    /* trie: node 0 is the root; a node that gains a child stops being terminal (cr-diccode.c:47-70) */
    uint32_t cap = 4096, nn = 1, tw = 0;
    uint32_t* next = (uint32_t*)calloc((size_t)cap * 128, 4);
    int32_t* ids = (int32_t*)calloc(cap, 4);
    if (!next || !ids) { free(words); free(wlen); free(next); free(ids); return CRGPU_E_NOMEM; }
    for (uint32_t k = 0; k < nw; k++) {
        const uint8_t* w = words + (size_t)k * CR_DIC_WORD_STRIDE;
        uint32_t at = 0;
        for (uint32_t i = 0; i < wlen[k]; i++) {
            uint32_t ch = w[i] & 127u;
            if (next[(size_t)at * 128 + ch] == 0) {
                if (nn >= cap) {
                    uint32_t ncap = cap * 2;
                    next = (uint32_t*)realloc(next, (size_t)ncap * 128 * 4);
                    ids = (int32_t*)realloc(ids, (size_t)ncap * 4);
                    memset(next + (size_t)cap * 128, 0, (size_t)(ncap - cap) * 128 * 4);
                    memset(ids + cap, 0, (size_t)(ncap - cap) * 4);
                    cap = ncap;
                }
                ids[at] = -1;
                next[(size_t)at * 128 + ch] = nn++;
            }
            at = next[(size_t)at * 128 + ch];
        }
        ids[at] = (int32_t)tw++;
    }
    for (uint32_t ch = 'A'; ch < 'Z'; ch++) next[ch] = next[ch + 32];          /* cr-diccode.c:107-109 ('Z' excluded) */
    for (uint32_t i = 0; i < nn; i++) {                                         /* cr-diccode.c:110-117 */
        uint32_t sp = next[(size_t)i * 128 + ' '];
        if (sp) {
            const char alias[4] = {'.', ',', ':', ';'};
            for (int a = 0; a < 4; a++) if (!next[(size_t)i * 128 + alias[a]]) next[(size_t)i * 128 + alias[a]] = sp;
        }
    }
    for (size_t e = 0; e < (size_t)nn * 128; e++)                               /* terminal flag rides on the link */
        if (next[e] && ids[next[e]] != -1) next[e] |= CR_DIC_TERMINAL;
    crgpu_dict* d = (crgpu_dict*)calloc(1, sizeof *d);
    int rc = CRGPU_OK;
    if (!d) rc = CRGPU_E_NOMEM;
    if (rc == CRGPU_OK && hipSetDevice(c->device) != hipSuccess) rc = CRGPU_E_NODEVICE;
    if (rc == CRGPU_OK) {
        d->ctx = c; d->nwords = nw; d->nnodes = nn; d->trie_words = tw;
        if (hipMalloc((void**)&d->d_next, (size_t)nn * 128 * 4) != hipSuccess || hipMalloc((void**)&d->d_ids, (size_t)nn * 4) != hipSuccess ||
            hipMalloc((void**)&d->d_words, (size_t)(nw + 1) * CR_DIC_WORD_STRIDE) != hipSuccess || hipMalloc((void**)&d->d_wlen, nw + 1) != hipSuccess)
            rc = CRGPU_E_NOMEM;
    }
    if (rc == CRGPU_OK) {
        if (hipMemcpy(d->d_next, next, (size_t)nn * 128 * 4, hipMemcpyHostToDevice) != hipSuccess ||
            hipMemcpy(d->d_ids, ids, (size_t)nn * 4, hipMemcpyHostToDevice) != hipSuccess ||
            hipMemcpy(d->d_words, words, (size_t)(nw + 1) * CR_DIC_WORD_STRIDE, hipMemcpyHostToDevice) != hipSuccess ||
            hipMemcpy(d->d_wlen, wlen, nw + 1, hipMemcpyHostToDevice) != hipSuccess)
            rc = CRGPU_E_NODEVICE;
    }
    free(words); free(wlen); free(next); free(ids);
    if (rc != CRGPU_OK) {
        if (d) { (void)hipFree(d->d_next); (void)hipFree(d->d_ids); (void)hipFree(d->d_words); (void)hipFree(d->d_wlen); free(d); }
        return rc;
    }
    *out = d;
    return CRGPU_OK;
}

extern "C" void crgpu_dict_destroy(crgpu_dict* d) {
    if (!d) return;
    (void)hipSetDevice(d->ctx->device);
    (void)hipStreamSynchronize(d->ctx->stream);
    (void)hipFree(d->d_next); (void)hipFree(d->d_ids); (void)hipFree(d->d_words); (void)hipFree(d->d_wlen); (void)hipFree(d->d_tmp); (void)hipFree(d->d_match);
    free(d);
}

extern "C" int crgpu_dict_words(const crgpu_dict* d) { return d ? (int)d->trie_words : 0; }

static int dict_launch(crgpu_ctx* c, crgpu_dict* d, int decode, CrBatch& B, uint32_t max_block, int sync) {
    if (!d || d->ctx != c) return CRGPU_E_ARG;
    if (max_block > CRGPU_MAX_BLOCK + 1u) return CRGPU_E_ARG;        /* a raw block travels as n + 1 bytes (cr-diccode.c:208-217) */
    CR_TRY(c, hipSetDevice(c->device));
    uint32_t grid = (uint32_t)c->num_cu * 16u;
    if (grid > B.nblocks) grid = B.nblocks;
    if (grid == 0) return CRGPU_OK;
    CrDictBatch DB; memset(&DB, 0, sizeof DB);
    DB.dict.next = d->d_next; DB.dict.ids = d->d_ids; DB.dict.words = d->d_words; DB.dict.wlen = d->d_wlen;
    DB.dict.nwords = d->nwords;
    DB.dict.level1 = (uint32_t)((65535 - (int)d->nwords) / 255 - 1);            /* cr-diccode.h:40 */
    if (!decode) {
        /* worst case per piece pair: 3 bytes per input byte + 8 + 2*4, plus the 11-byte trailer */
        DB.tmp_stride = align_up((u64)max_block * 3u + 64u * (max_block / 2000000u + 2u), 256);
        int rc = grow(c, &d->d_tmp, &d->tmp_cap, (size_t)(DB.tmp_stride * grid));
        if (rc != CRGPU_OK) return rc;
        DB.tmp = d->d_tmp;
        DB.max_block = max_block;
        DB.match_stride = align_up((u64)(max_block ? max_block : 1u), 64);
        rc = grow(c, &d->d_match, &d->match_cap, (size_t)(DB.match_stride * 4u * B.nblocks));
        if (rc != CRGPU_OK) return rc;
        DB.match = (uint32_t*)d->d_match;
    }
    B.ticket = c->ticket;
    CR_TRY(c, hipMemsetAsync(c->ticket, 0, 8, c->stream));
    { const int brc = stage_begin(c); if (brc != CRGPU_OK) return brc; }
    CR_TRY(c, hipEventRecord(c->ev0, c->stream));
    CR_TRY(c, hipEventRecord(c->ev_stage[0], c->stream));
    c->n_stages = 0;
    if (decode) {
        hipLaunchKernelGGL(k_dict_decode, CR_G(k_dict_decode, CRGPU_WAVE, 0), dim3(CRGPU_WAVE), 0, c->stream, B, DB);
        c->stage_name[c->n_stages++] = "k_dict_decode";
    } else {
        const uint32_t chunks = (max_block + CR_DM_CHUNK - 1u) / CR_DM_CHUNK;
        hipLaunchKernelGGL(k_dict_match, dim3(B.nblocks, chunks ? chunks : 1u), dim3(256), 0, c->stream, B, DB);
        CR_TRY(c, hipGetLastError());
        c->stage_name[c->n_stages++] = "k_dict_match";
        CR_TRY(c, hipEventRecord(c->ev_stage[c->n_stages], c->stream));
        hipLaunchKernelGGL(k_dict_encode, CR_G(k_dict_encode, CRGPU_WAVE, 0), dim3(CRGPU_WAVE), 0, c->stream, B, DB);
        c->stage_name[c->n_stages++] = "k_dict_encode";
    }
    CR_TRY(c, hipGetLastError());
    CR_TRY(c, hipEventRecord(c->ev_stage[c->n_stages], c->stream));
    CR_TRY(c, hipEventRecord(c->ev_mid, c->stream));
    CR_TRY(c, hipEventRecord(c->ev1, c->stream));
    c->timed = 1;
    { const int erc = stage_end(c); if (erc != CRGPU_OK) return erc; }
    if (sync) CR_TRY(c, hipStreamSynchronize(c->stream));
    return CRGPU_OK;
}

extern "C" int crgpu_dict_encode_blocks_dev(crgpu_ctx* c, crgpu_dict* d, const uint8_t* in, const uint64_t* in_off,
                                            const uint32_t* in_size, uint32_t nblocks, uint32_t max_block,
                                            uint8_t* out, const uint64_t* out_off, uint32_t* out_size, int sync) {
    if (!c || !d || (nblocks && (!in || !in_off || !in_size || !out || !out_off || !out_size))) return CRGPU_E_ARG;
    CrBatch B; memset(&B, 0, sizeof B);
    B.in = in; B.in_off = (const u64*)in_off; B.in_size = in_size;
    B.out = out; B.out_off = (const u64*)out_off; B.out_size = out_size; B.nblocks = nblocks;
    return dict_launch(c, d, 0, B, max_block, sync);
}

extern "C" int crgpu_dict_decode_blocks_dev(crgpu_ctx* c, crgpu_dict* d, const uint8_t* in, const uint64_t* in_off,
                                            const uint32_t* in_size, uint32_t nblocks, uint32_t max_block,
                                            uint8_t* out, const uint64_t* out_off, const uint32_t* out_cap,
                                            uint32_t* out_size, int sync) {
    if (!c || !d || (nblocks && (!in || !in_off || !in_size || !out || !out_off || !out_cap || !out_size))) return CRGPU_E_ARG;
    CrBatch B; memset(&B, 0, sizeof B);
    B.in = in; B.in_off = (const u64*)in_off; B.in_size = in_size;
    B.out = out; B.out_off = (const u64*)out_off; B.out_cap = out_cap; B.out_size = out_size; B.nblocks = nblocks;
    return dict_launch(c, d, 1, B, max_block, sync);
}

/* ------------------------------------------------------------------ device pack (the container's concatenation) */

extern "C" int crgpu_pack_blocks_dev(crgpu_ctx* c, const uint8_t* in, const uint64_t* in_off, const uint32_t* in_size,
                                     uint32_t nblocks, const uint8_t* filt, int prec, int with_headers,
                                     uint8_t* out, uint64_t* out_off, uint64_t* total, int sync) {
    if (!c || !total || (nblocks && (!in || !in_off || !in_size || !out || !out_off))) return CRGPU_E_ARG;
    CR_TRY(c, hipSetDevice(c->device));
    CrPack P; memset(&P, 0, sizeof P);
    P.in = in; P.in_off = (const u64*)in_off; P.in_size = in_size; P.filt = filt; P.nblocks = nblocks;
    P.head = with_headers ? 6u : 0u; P.prec = prec ? 1u : 0u;
    P.out = out; P.out_off = (u64*)out_off; P.total = (u64*)total;
    { const int brc = stage_begin(c); if (brc != CRGPU_OK) return brc; }
    CR_TRY(c, hipEventRecord(c->ev0, c->stream));
    CR_TRY(c, hipEventRecord(c->ev_stage[0], c->stream));
    hipLaunchKernelGGL(k_pack_scan, dim3(1), dim3(CR_PACK_SCAN_THREADS), 0, c->stream, P);
    CR_TRY(c, hipEventRecord(c->ev_stage[1], c->stream));
    if (nblocks) {
        uint32_t grid = (uint32_t)c->num_cu * 8u;
        if (grid > nblocks) grid = nblocks;
        hipLaunchKernelGGL(k_pack_copy, dim3(grid), dim3(256), 0, c->stream, P);
    }
    CR_TRY(c, hipGetLastError());
    CR_TRY(c, hipEventRecord(c->ev_stage[2], c->stream));
    c->n_stages = 2; c->stage_name[0] = "k_pack_scan"; c->stage_name[1] = "k_pack_copy";
    CR_TRY(c, hipEventRecord(c->ev_mid, c->stream));
    CR_TRY(c, hipEventRecord(c->ev1, c->stream));
    c->timed = 1;
    { const int erc = stage_end(c); if (erc != CRGPU_OK) return erc; }
    if (sync) CR_TRY(c, hipStreamSynchronize(c->stream));
    return CRGPU_OK;
}

extern "C" int crgpu_dict_decoded_sizes_dev(crgpu_ctx* c, const uint8_t* in, const uint64_t* in_off, const uint32_t* in_size,
                                            uint32_t nblocks, uint32_t* out_size, int sync) {
    if (!c || (nblocks && (!in || !in_off || !in_size || !out_size))) return CRGPU_E_ARG;
    CR_TRY(c, hipSetDevice(c->device));
    if (nblocks) hipLaunchKernelGGL(k_dict_sizes, dim3((nblocks + 255u) / 256u), dim3(256), 0, c->stream, in, (const u64*)in_off, in_size, nblocks, out_size);
    CR_TRY(c, hipGetLastError());
    if (sync) CR_TRY(c, hipStreamSynchronize(c->stream));
    return CRGPU_OK;
}

extern "C" int crgpu_offsets_dev(crgpu_ctx* c, const uint32_t* sizes, uint32_t nblocks, uint64_t* out_off, uint64_t* total, int sync) {
    if (!c || !total || (nblocks && (!sizes || !out_off))) return CRGPU_E_ARG;
    CR_TRY(c, hipSetDevice(c->device));
    CrPack P; memset(&P, 0, sizeof P);
    P.in_size = sizes; P.nblocks = nblocks; P.out_off = (u64*)out_off; P.total = (u64*)total;
    hipLaunchKernelGGL(k_pack_scan, dim3(1), dim3(CR_PACK_SCAN_THREADS), 0, c->stream, P);
    CR_TRY(c, hipGetLastError());
    if (sync) CR_TRY(c, hipStreamSynchronize(c->stream));
    return CRGPU_OK;
}

/* ------------------------------------------------------------------ host-pointer wrappers */

static int grow(crgpu_ctx* c, uint8_t** p, size_t* cap, size_t want) {
    if (*cap >= want) return CRGPU_OK;
    if (*p) (void)hipFree(*p);
    *p = NULL; *cap = 0;
    size_t sz = want + want / 4 + 4096;
    if (hipMalloc((void**)p, sz) != hipSuccess) { snprintf(c->err, sizeof c->err, "hipMalloc(%zu) failed", sz); return CRGPU_E_NOMEM; }
    *cap = sz;
    return CRGPU_OK;
}

static int host_call(crgpu_ctx* c, int codec, crgpu_dict* dict, int decode, const uint8_t* in, const uint64_t* in_off,
                     const uint32_t* in_size, uint32_t nblocks, uint8_t* out, const uint64_t* out_off,
                     const uint32_t* out_cap, uint32_t* out_size) {
    if (!c) return CRGPU_E_ARG;
    if (nblocks == 0) return CRGPU_OK;
    if (!in || !in_off || !in_size || !out || !out_off || !out_size || (decode && !out_cap)) return CRGPU_E_ARG;
    CR_TRY(c, hipSetDevice(c->device));
    /* pack blocks back to back on the device (16-byte aligned slots) */
    u64* h_in_off = (u64*)malloc(sizeof(u64) * nblocks * 2);
    uint32_t* h_cap = (uint32_t*)malloc(sizeof(uint32_t) * nblocks);
    if (!h_in_off || !h_cap) { free(h_in_off); free(h_cap); return CRGPU_E_NOMEM; }
    u64* h_out_off = h_in_off + nblocks;
    u64 in_total = 0, out_total = 0;
    uint32_t max_block = 0;
    for (uint32_t b = 0; b < nblocks; b++) {
        h_in_off[b] = in_total;  in_total = align_up(in_total + in_size[b], 16);
        uint32_t room;
        if (decode) {
            room = out_cap[b];
            if (room > max_block) max_block = room;
        } else {
            room = dict ? in_size[b] + 1u : crgpu_bound(codec, in_size[b]);
            if (in_size[b] > max_block) max_block = in_size[b];
        }
        h_cap[b] = room;
        h_out_off[b] = out_total; out_total = align_up(out_total + room, 16);
    }
    int rc = CRGPU_OK;
    if (max_block > CRGPU_MAX_BLOCK + 1u) { snprintf(c->err, sizeof c->err, "block of %u bytes exceeds CRGPU_MAX_BLOCK + 1", max_block); rc = CRGPU_E_ARG; }
    size_t meta = (size_t)nblocks * (8 + 8 + 4 + 4 + 4);
    if (rc == CRGPU_OK) rc = grow(c, &c->d_in, &c->d_in_cap, (size_t)in_total + 16);
    if (rc == CRGPU_OK) rc = grow(c, &c->d_out, &c->d_out_cap, (size_t)out_total + 16);
    if (rc == CRGPU_OK) rc = grow(c, &c->d_meta, &c->d_meta_cap, meta);
    if (rc != CRGPU_OK) { free(h_in_off); free(h_cap); return rc; }
    uint64_t* d_in_off = (uint64_t*)c->d_meta;
    uint64_t* d_out_off = d_in_off + nblocks;
    uint32_t* d_in_size = (uint32_t*)(d_out_off + nblocks);
    uint32_t* d_cap = d_in_size + nblocks;
    uint32_t* d_out_size = d_cap + nblocks;
    hipError_t e = hipSuccess;
    /* one copy per run of blocks that lie back to back on both sides (a stream cut into blocks is ONE run) */
    for (uint32_t b = 0; b < nblocks && e == hipSuccess;) {
        uint32_t last = b;
        while (last + 1u < nblocks && in_off[last + 1u] - in_off[b] == h_in_off[last + 1u] - h_in_off[b] &&
               in_off[last + 1u] == in_off[last] + in_size[last]) last++;
        const u64 bytes = in_off[last] + in_size[last] - in_off[b];
        if (bytes) e = hipMemcpyAsync(c->d_in + h_in_off[b], in + in_off[b], (size_t)bytes, hipMemcpyHostToDevice, c->stream);
        b = last + 1u;
    }
    if (e == hipSuccess) e = hipMemcpyAsync(d_in_off, h_in_off, sizeof(u64) * nblocks * 2, hipMemcpyHostToDevice, c->stream);
    if (e == hipSuccess) e = hipMemcpyAsync(d_in_size, in_size, 4u * nblocks, hipMemcpyHostToDevice, c->stream);
    if (e == hipSuccess) e = hipMemcpyAsync(d_cap, h_cap, 4u * nblocks, hipMemcpyHostToDevice, c->stream);
    if (e != hipSuccess) { free(h_in_off); free(h_cap); return fail(c, e, "hipMemcpyAsync(H2D)"); }
    if (dict && decode) rc = crgpu_dict_decode_blocks_dev(c, dict, c->d_in, d_in_off, d_in_size, nblocks, max_block, c->d_out, d_out_off, d_cap, d_out_size, 0);
    else if (dict)      rc = crgpu_dict_encode_blocks_dev(c, dict, c->d_in, d_in_off, d_in_size, nblocks, max_block, c->d_out, d_out_off, d_out_size, 0);
    else if (decode)    rc = crgpu_decode_blocks_dev(c, codec, c->d_in, d_in_off, d_in_size, nblocks, max_block, c->d_out, d_out_off, d_cap, d_out_size, 0);
    else                rc = crgpu_encode_blocks_dev(c, codec, c->d_in, d_in_off, d_in_size, nblocks, max_block, c->d_out, d_out_off, d_out_size, 0);
    if (rc == CRGPU_OK) {
        e = hipMemcpyAsync(out_size, d_out_size, 4u * nblocks, hipMemcpyDeviceToHost, c->stream);
        if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
        /* runs of blocks whose slots are spaced alike on both sides come back in one copy, gaps included, unless the
         * gaps outweigh the data (the caller owns the whole of every slot: out_off[b] .. + room) */
        for (uint32_t b = 0; b < nblocks && e == hipSuccess;) {
            if (out_size[b] == 0xFFFFFFFFu) { rc = CRGPU_E_CORRUPT; b++; continue; }
            uint32_t last = b;
            u64 useful = out_size[b];
            while (last + 1u < nblocks && out_size[last + 1u] != 0xFFFFFFFFu && out_size[last + 1u] <= h_cap[last + 1u] &&
                   out_off[last + 1u] - out_off[b] == h_out_off[last + 1u] - h_out_off[b]) { last++; useful += out_size[last]; }
            const u64 span = h_out_off[last] + out_size[last] - h_out_off[b];
            if (last > b && span <= 4u * useful + 65536u) {
                e = hipMemcpyAsync(out + out_off[b], c->d_out + h_out_off[b], (size_t)span, hipMemcpyDeviceToHost, c->stream);
                b = last + 1u;
            } else {
                if (out_size[b]) e = hipMemcpyAsync(out + out_off[b], c->d_out + h_out_off[b], out_size[b], hipMemcpyDeviceToHost, c->stream);
                b++;
            }
        }
        if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
        if (e != hipSuccess) rc = fail(c, e, "D2H");
    }
    free(h_in_off); free(h_cap);
    return rc;
}

extern "C" int crgpu_encode_blocks(crgpu_ctx* c, int codec, const uint8_t* in, const uint64_t* in_off,
                                   const uint32_t* in_size, uint32_t nblocks, uint8_t* out,
                                   const uint64_t* out_off, uint32_t* out_size) {
    return host_call(c, codec, NULL, 0, in, in_off, in_size, nblocks, out, out_off, NULL, out_size);
}
extern "C" int crgpu_decode_blocks(crgpu_ctx* c, int codec, const uint8_t* in, const uint64_t* in_off,
                                   const uint32_t* in_size, uint32_t nblocks, uint8_t* out,
                                   const uint64_t* out_off, const uint32_t* out_cap, uint32_t* out_size) {
    return host_call(c, codec, NULL, 1, in, in_off, in_size, nblocks, out, out_off, out_cap, out_size);
}
extern "C" int crgpu_dict_encode_blocks(crgpu_ctx* c, crgpu_dict* d, const uint8_t* in, const uint64_t* in_off,
                                        const uint32_t* in_size, uint32_t nblocks, uint8_t* out,
                                        const uint64_t* out_off, uint32_t* out_size) {
    if (!d) return CRGPU_E_ARG;
    return host_call(c, 0, d, 0, in, in_off, in_size, nblocks, out, out_off, NULL, out_size);
}
extern "C" int crgpu_dict_decode_blocks(crgpu_ctx* c, crgpu_dict* d, const uint8_t* in, const uint64_t* in_off,
                                        const uint32_t* in_size, uint32_t nblocks, uint8_t* out,
                                        const uint64_t* out_off, const uint32_t* out_cap, uint32_t* out_size) {
    if (!d) return CRGPU_E_ARG;
    return host_call(c, 0, d, 1, in, in_off, in_size, nblocks, out, out_off, out_cap, out_size);
}

/* wave-primitive self test: in = 66 u32 (64 lane values, mask limit, table index), out = 384 u32 */
extern "C" int crgpu_selftest(crgpu_ctx* c, const uint32_t* in, uint32_t* out) {
    if (!c || !in || !out) return CRGPU_E_ARG;
    CR_TRY(c, hipSetDevice(c->device));
    uint32_t *d_in = NULL, *d_out = NULL;
    CR_TRY(c, hipMalloc((void**)&d_in, 66 * 4));
    CR_TRY(c, hipMalloc((void**)&d_out, 448 * 4));
    CR_TRY(c, hipMemcpy(d_in, in, 66 * 4, hipMemcpyHostToDevice));
    hipLaunchKernelGGL(k_selftest, dim3(1), dim3(CRGPU_WAVE), 0, c->stream, d_in, d_out);
    CR_TRY(c, hipStreamSynchronize(c->stream));
    CR_TRY(c, hipMemcpy(out, d_out, 448 * 4, hipMemcpyDeviceToHost));
    (void)hipFree(d_in); (void)hipFree(d_out);
    return CRGPU_OK;
}

/* ------------------------------------------------------------------ data_block_t (cr-datablock.c:31-56) */

extern "C" void data_block_reserve(data_block_t* b, uint32_t size) {
    if (size > b->m_capacity || size < b->m_capacity / 2) {       /* grows, and shrinks below half */
        b->m_capacity = (uint32_t)(size * 1.2);
        b->m_data = (uint8_t*)realloc(b->m_data, b->m_capacity);
    }
}
extern "C" void data_block_resize(data_block_t* b, uint32_t size) {
    data_block_reserve(b, size);
    b->m_size = size;
}
extern "C" void data_block_add(data_block_t* b, uint8_t byte) {
    if (b->m_size == b->m_capacity) {
        b->m_capacity = (uint32_t)(b->m_size * 1.2 + 1);
        b->m_data = (uint8_t*)realloc(b->m_data, b->m_capacity);
    }
    b->m_data[b->m_size++] = byte;
}
extern "C" void data_block_destroy(data_block_t* b) { free(b->m_data); }

/* ------------------------------------------------------------------ reference-signature shims */
/* State contract of the reference: the models persist across calls until
 * reset_models(). The shims run on a one-slot "persist" context: table capacities are fixed, the
 * PPM context register / node generation (and comprox's side models) are parked in the arena at
 * the end of a call and picked up by the next one unless reset_models() came in between.
 *
 * The reference signatures are void and there is no CPU fallback, so a failure (no gfx950 device, a HIP
 * error, a malformed block) is REPORTED: it is recorded (crgpu_shim_status / crgpu_shim_last_error), the
 * handler installed with crgpu_shim_set_error_handler is called, and the shim returns with an empty output
 * block. Without a handler the message goes to stderr and the process exits with status 1 — a tool that
 * carried on would write a broken file. */

/* switches the reference's front-ends assign directly (roxmain/main.c:88,99, rolzmain/main.c:87; declared
 * extern in roxmain/cr-matcher.h:52,56 and rolzmain/cr-matcher.h:43); the shims read them at every call */
extern "C" {
int flexible_parsing = 0;                       /* -f */
uint32_t match_limit = CR_ROX_LIMIT;            /* -m, roxmain/cr-matcher.c:39 */
/* every front-end defines its container magic (src/main.c:47; "...-comprox" / "...-comprolz" / "...-comprop"):
 * when the executable exports one, it names the codec the shims mirror */
extern const char* cr_magic_header __attribute__((weak));
}

static crgpu_ctx* g_shim;                       /* model-carrying one-slot context: blocks that continue the previous block's models */
static crgpu_ctx* g_fast;                       /* plain context: blocks that start from fresh models go through the batched kernels */
static int g_fresh = 1;                         /* reset_models() was called since the last block (or nothing was coded yet) */
static int g_expect_dependent = 0;              /* crgpu_shim_expect_dependent_blocks: fresh blocks run on the model-carrying context too */
/* A block coded from fresh models leaves no model state behind on the fast context. If the NEXT block arrives without a
 * reset_models() in between (the stock tool's second block of a file, src/main.c:174-206), that state is rebuilt first by
 * running the remembered block through the model-carrying coder (its output is dropped): kind 1 = an lzencode input,
 * 2 = an lzdecode input of `cap` decoded bytes. */
static struct { int kind; uint8_t* data; uint32_t n, cap; } g_replay;
static int g_shim_codec = 0;                    /* 0: not chosen yet */
static int g_shim_device = 0;
static int g_shim_status = CRGPU_OK;
static char g_shim_err[320];
static crgpu_error_fn g_shim_handler;
static void* g_shim_handler_user;

extern "C" void crgpu_shim_set_error_handler(crgpu_error_fn fn, void* user) { g_shim_handler = fn; g_shim_handler_user = user; }
extern "C" int crgpu_shim_status(void) { return g_shim_status; }
extern "C" const char* crgpu_shim_last_error(void) { return g_shim_err; }

static void shim_fail(int code, const char* what, const char* detail) {
    g_shim_status = code;
    snprintf(g_shim_err, sizeof g_shim_err, "crgpu: %s failed (%d)%s%s", what, code, detail && *detail ? ": " : "", detail ? detail : "");
    if (g_shim_handler) { g_shim_handler(code, g_shim_err, g_shim_handler_user); return; }
    fprintf(stderr, "%s\n", g_shim_err);
    exit(EXIT_FAILURE);
}

static int ends_with(const char* s, const char* tail) {
    const size_t a = strlen(s), b = strlen(tail);
    return a >= b && memcmp(s + a - b, tail, b) == 0;
}

static int shim_codec(void) {
    if (!g_shim_codec) {
        g_shim_codec = CRGPU_CODEC_ROP;
        const char* const* magic = &cr_magic_header;            /* NULL when no front-end defines it */
        if (magic && *magic) {
            if (ends_with(*magic, "-comprox")) g_shim_codec = CRGPU_CODEC_ROX;
            else if (ends_with(*magic, "-comprolz")) g_shim_codec = CRGPU_CODEC_ROLZ;
        }
    }
    return g_shim_codec;
}

extern "C" int crgpu_shim_config(int codec, int device) {
    if (codec != CRGPU_CODEC_ROP && codec != CRGPU_CODEC_ROX && codec != CRGPU_CODEC_ROLZ) return CRGPU_E_ARG;
    g_shim_codec = codec;
    g_shim_device = device;
    return CRGPU_OK;
}

extern "C" int crgpu_shim_codec(void) { return shim_codec(); }

/* A caller that knows more blocks will follow WITHOUT reset_models() (the stock block loop on a file of several blocks,
 * src/main.c:174-206) says so: a block that starts from fresh models then runs on the model-carrying context straight away —
 * slower for that block than the batched kernels, but the next block continues from its models instead of rebuilding them by
 * running this block's input through the model-carrying coder a second time (16 MiB: seconds). Bytes are the same either way. */
extern "C" void crgpu_shim_expect_dependent_blocks(int on) { g_expect_dependent = on != 0; }
/* kernel milliseconds of the most recent lzencode / lzdecode shim call (HIP events), -1 if none */
static crgpu_ctx* g_last_shim_ctx;
extern "C" float crgpu_shim_last_kernel_ms(void) { return g_last_shim_ctx ? crgpu_last_kernel_ms(g_last_shim_ctx) : -1.0f; }

static crgpu_ctx* fast_ctx(const char* who);

/* Start the device side of the shims now (HIP runtime, context, stream) instead of at the first lzencode / lzdecode, so
 * that a tool can do it while it is busy elsewhere (comp*-gpu: while the host's dicpick pass runs). */
extern "C" int crgpu_shim_prepare(void) {
    g_shim_status = CRGPU_OK;
    return fast_ctx("crgpu_shim_prepare") ? CRGPU_OK : g_shim_status;
}

/* page-locked host memory for the buffers a caller hands to the host-pointer entry points: copies to and from it run
 * as DMA at the link's rate (pageable memory is staged through the runtime's bounce buffers at a fraction of it) */
extern "C" void* crgpu_host_alloc(size_t bytes) {
    void* p = NULL;
    if (hipHostMalloc(&p, bytes ? bytes : 1u, hipHostMallocDefault) != hipSuccess) return NULL;
    return p;
}
extern "C" void crgpu_host_free(void* p) { if (p) (void)hipHostFree(p); }

static crgpu_ctx* shim_ctx(const char* who) {
    if (!g_shim) {
        int rc = crgpu_create(&g_shim, g_shim_device);
        if (rc != CRGPU_OK) {
            g_shim = NULL;
            shim_fail(rc, who, "no usable gfx950 device; there is no CPU fallback");
            return NULL;
        }
        g_shim->persist = 1;
        g_shim->next_fresh = 1;
    }
    g_shim->rox_limit = match_limit ? match_limit : 1u;
    g_shim->flexible = flexible_parsing != 0;
    return g_shim;
}

extern "C" int crgpu_shim_rox_chain_limit(uint32_t limit) {
    if (limit == 0) return CRGPU_E_ARG;
    match_limit = limit;
    return CRGPU_OK;
}

extern "C" int crgpu_shim_flexible_parsing(int on) {
    flexible_parsing = on != 0;
    return CRGPU_OK;
}

/* reset_models(), src/ropmain/cr-coder.c:73-83 / src/roxmain/cr-coder.c:88-114: the next block starts
 * from freshly initialised models; without it the next block continues with the previous one's. */
static void replay_drop(void) { free(g_replay.data); g_replay.data = NULL; g_replay.kind = 0; g_replay.n = g_replay.cap = 0; }

static void replay_keep(int kind, const uint8_t* data, uint32_t n, uint32_t cap) {
    replay_drop();
    g_replay.data = (uint8_t*)malloc(n ? n : 1u);
    if (!g_replay.data) { g_replay.kind = -1; return; }     /* no copy: a block that continues these models cannot be coded — carried_ctx reports it */
    if (n) memcpy(g_replay.data, data, n);
    g_replay.kind = kind; g_replay.n = n; g_replay.cap = cap;
}

extern "C" void reset_models(void) {
    g_fresh = 1;
    replay_drop();
    if (g_shim) g_shim->next_fresh = 1;        /* a context that does not exist yet starts fresh anyway */
}

static crgpu_ctx* fast_ctx(const char* who) {
    if (!g_fast) {
        int rc = crgpu_create(&g_fast, g_shim_device);
        if (rc != CRGPU_OK) { g_fast = NULL; shim_fail(rc, who, "no usable gfx950 device; there is no CPU fallback"); return NULL; }
    }
    g_fast->rox_limit = match_limit ? match_limit : 1u;
    g_fast->flexible = flexible_parsing != 0;
    return g_fast;
}

/* the models the next block continues from exist on the model-carrying context; returns it (NULL after a reported failure) */
static crgpu_ctx* carried_ctx(const char* who, int codec) {
    crgpu_ctx* c = shim_ctx(who);
    if (!c) return NULL;
    if (g_replay.kind < 0) {                   /* the previous block's input could not be kept (replay_keep): its models cannot be rebuilt */
        shim_fail(CRGPU_E_NOMEM, who, "out of memory keeping the previous block for the model carry-over");
        return NULL;
    }
    if (g_replay.kind) {
        uint64_t zero = 0;
        uint32_t n = g_replay.n, produced = 0, cap = g_replay.kind == 1 ? crgpu_bound(codec, g_replay.n) : g_replay.cap;
        uint8_t* scratch = (uint8_t*)malloc(cap ? cap : 1u);
        int rc = scratch ? CRGPU_OK : CRGPU_E_NOMEM;
        c->next_fresh = 1;
        if (rc == CRGPU_OK) rc = g_replay.kind == 1 ? crgpu_encode_blocks(c, codec, g_replay.data, &zero, &n, 1, scratch, &zero, &produced)
                                                     : crgpu_decode_blocks(c, codec, g_replay.data, &zero, &n, 1, scratch, &zero, &cap, &produced);
        free(scratch);
        replay_drop();
        if (rc != CRGPU_OK) { shim_fail(rc, who, "could not rebuild the previous block's models"); return NULL; }
    }
    return c;
}

static uint32_t shim_header_bytes(int codec) {
    return codec == CRGPU_CODEC_ROX ? CRGPU_ROX_HEADER : codec == CRGPU_CODEC_ROLZ ? CRGPU_ROLZ_HEADER : CRGPU_ROP_HEADER;
}

extern "C" void lzencode(data_block_t* ib, data_block_t* ob, int print_information) {
    (void)print_information;
    const int codec = shim_codec();
    g_shim_status = CRGPU_OK;
    uint32_t n = ib->m_size, produced = 0;
    if (n > CRGPU_MAX_BLOCK + 1u) {
        /* larger than the device tables are laid out for (-b above 16): the block is written in the reference's
         * stored form (a zeroed header + the raw bytes: ropmain/cr-coder.c:222-228, roxmain/cr-coder.c:311-317,
         * rolzmain/cr-coder.c:251-257), which every decoder of the format accepts; the models are left as they are */
        const uint32_t hdr = shim_header_bytes(codec);
        if (n > 0xFFFFFFFFu - hdr) { data_block_resize(ob, 0); shim_fail(CRGPU_E_ARG, "lzencode", "block too large"); return; }
        data_block_resize(ob, hdr + n);
        memset(ob->m_data, 0, hdr);
        memcpy(ob->m_data + hdr, ib->m_data, n);
        return;
    }
    /* fresh models (the first block of a file, the dictionary blob, every file of up to one block): the batched kernel
     * pipeline; a block that continues the previous one's models: the model-carrying one-wave coder */
    const int fresh = g_fresh && !g_expect_dependent;
    crgpu_ctx* c = fresh ? fast_ctx("lzencode") : carried_ctx("lzencode", codec);
    if (!c) { data_block_resize(ob, 0); return; }
    uint64_t zero = 0;
    data_block_resize(ob, crgpu_bound(codec, n));
    static uint8_t dummy;
    int rc = crgpu_encode_blocks(c, codec, n ? ib->m_data : &dummy, &zero, &n, 1, ob->m_data, &zero, &produced);
    if (rc != CRGPU_OK || produced == 0xFFFFFFFFu) { data_block_resize(ob, 0); shim_fail(rc != CRGPU_OK ? rc : CRGPU_E_ARG, "lzencode", crgpu_last_error(c)); return; }
    g_last_shim_ctx = c;
    if (fresh) replay_keep(1, n ? ib->m_data : &dummy, n, 0);
    g_fresh = 0;
    data_block_resize(ob, produced);
}

extern "C" void lzdecode(data_block_t* ib, data_block_t* ob, int print_information) {
    (void)print_information;
    const int codec = shim_codec();
    g_shim_status = CRGPU_OK;
    const uint32_t hdr = shim_header_bytes(codec);
    if (ib->m_size < hdr) { shim_fail(CRGPU_E_CORRUPT, "lzdecode", "truncated block"); return; }
    uint32_t total;
    /* the coded flag is byte 0 of comprop's and comprox's header, byte 1 of comprolz's (rolzmain/cr-coder.c:63-71) */
    const int coded = ib->m_data[codec == CRGPU_CODEC_ROLZ ? 1 : 0];
    if (coded) {
        const uint8_t* p = ib->m_data + 4;
        total = (uint32_t)p[0] | ((uint32_t)p[1] << 8) | ((uint32_t)p[2] << 16) | ((uint32_t)p[3] << 24);
    } else {
        total = ib->m_size - hdr;
    }
    /* ropmain appends a stored block to ob (cr-coder.c:244-246) but restarts ob for a coded one
     * (cr-coder.c:251); roxmain always restarts ob (roxmain/cr-coder.c:430) */
    /* rolzmain appends in its stored branch too (data_block_add, rolzmain/cr-coder.c:303-308) */
    uint32_t base = (codec != CRGPU_CODEC_ROX && !coded) ? ob->m_size : 0u;
    if (total > CRGPU_MAX_BLOCK + 1u) {
        if (coded || total > 0xFFFFFFFFu - base) { shim_fail(CRGPU_E_ARG, "lzdecode", "block larger than CRGPU_MAX_BLOCK + 1"); return; }
        data_block_resize(ob, base + total);                   /* the stored form lzencode writes for oversized blocks */
        memcpy(ob->m_data + base, ib->m_data + hdr, total);
        return;
    }
    const int fresh = g_fresh && !g_expect_dependent;
    crgpu_ctx* c = fresh ? fast_ctx("lzdecode") : carried_ctx("lzdecode", codec);
    if (!c) return;
    uint64_t zero = 0;
    uint32_t n = ib->m_size, produced = 0, cap = total;
    data_block_resize(ob, base + total);
    static uint8_t dummy;
    int rc = crgpu_decode_blocks(c, codec, ib->m_data, &zero, &n, 1, total ? ob->m_data + base : &dummy, &zero, &cap, &produced);
    if (rc != CRGPU_OK) { data_block_resize(ob, base); shim_fail(rc, "lzdecode", rc == CRGPU_E_CORRUPT ? "malformed block" : crgpu_last_error(c)); return; }
    g_last_shim_ctx = c;
    if (fresh) replay_keep(2, ib->m_data, n, total);
    g_fresh = 0;
    data_block_resize(ob, base + produced);
}

/* ------------------------------------------------------------------ dictionary shims (cr-diccode.h:44-48) */

static crgpu_dict* g_shim_dict;

extern "C" int dictionary_load(const char* dicstr, int init_trie) {
    (void)init_trie;                       /* the device copy always carries both the trie and the word table */
    g_shim_status = CRGPU_OK;
    crgpu_ctx* c = shim_ctx("dictionary_load");
    if (!c) return 0;
    if (g_shim_dict) { shim_fail(CRGPU_E_ARG, "dictionary_load", "may be called once per process (as in the reference)"); return 0; }
    int rc = crgpu_dict_create(c, dicstr, &g_shim_dict);
    if (rc != CRGPU_OK) { g_shim_dict = NULL; shim_fail(rc, "dictionary_load", crgpu_last_error(c)); return 0; }
    return crgpu_dict_words(g_shim_dict);
}

extern "C" void dictionary_encode(data_block_t* ib, data_block_t* ob) {
    g_shim_status = CRGPU_OK;
    crgpu_ctx* c = shim_ctx("dictionary_encode");
    if (!c) { data_block_resize(ob, 0); return; }
    if (!g_shim_dict) { data_block_resize(ob, 0); shim_fail(CRGPU_E_ARG, "dictionary_encode", "called before dictionary_load"); return; }
    uint64_t zero = 0;
    uint32_t n = ib->m_size, produced = 0;
    static uint8_t dummy;
    if (n > CRGPU_MAX_BLOCK) {                              /* raw form (cr-diccode.c:208-217): the block + flag 0 */
        if (n == 0xFFFFFFFFu) { data_block_resize(ob, 0); shim_fail(CRGPU_E_ARG, "dictionary_encode", "block too large"); return; }
        data_block_resize(ob, n + 1u);
        memcpy(ob->m_data, ib->m_data, n);
        ob->m_data[n] = 0;
        return;
    }
    data_block_resize(ob, n + 1u);
    int rc = crgpu_dict_encode_blocks(c, g_shim_dict, n ? ib->m_data : &dummy, &zero, &n, 1, ob->m_data, &zero, &produced);
    if (rc != CRGPU_OK) { data_block_resize(ob, 0); shim_fail(rc, "dictionary_encode", crgpu_last_error(c)); return; }
    data_block_resize(ob, produced);
}

extern "C" void dictionary_decode(data_block_t* ib, data_block_t* ob, FILE* fpout_sync) {
    g_shim_status = CRGPU_OK;
    const uint8_t* s = ib->m_data;
    const uint32_t n = ib->m_size;
    if (n == 0) return;
    if (s[n - 1] == 0) {                                   /* cr-diccode.c:238-242: raw form replaces ob */
        data_block_resize(ob, n - 1u);
        memcpy(ob->m_data, s, n - 1u);
        return;
    }
    crgpu_ctx* c = shim_ctx("dictionary_decode");
    if (!c) return;
    if (!g_shim_dict) { shim_fail(CRGPU_E_ARG, "dictionary_decode", "called before dictionary_load"); return; }
    uint64_t total = 0;                                    /* sum of the pieces' recorded sizes */
    for (uint32_t pos = 0; (uint64_t)pos + 11u < n; ) {
        if (pos + 8u > n) { shim_fail(CRGPU_E_CORRUPT, "dictionary_decode", "truncated block"); return; }
        uint32_t a, b2;
        memcpy(&a, s + pos, 4); memcpy(&b2, s + pos + 4, 4);
        pos += 8u;
        if ((uint64_t)pos + a + b2 + 11u > n || a < 4u || b2 < 4u) { shim_fail(CRGPU_E_CORRUPT, "dictionary_decode", "malformed block"); return; }
        uint32_t t1, t2;
        memcpy(&t1, s + pos + a - 4u, 4); memcpy(&t2, s + pos + a + b2 - 4u, 4);
        total += (uint64_t)t1 + t2;
        pos += a + b2;
    }
    if (total > CRGPU_MAX_BLOCK) { shim_fail(CRGPU_E_ARG, "dictionary_decode", "block larger than CRGPU_MAX_BLOCK"); return; }
    uint64_t zero = 0;
    uint32_t nin = n, cap = (uint32_t)total, produced = 0;
    const uint32_t base = ob->m_size;                      /* cr-diccode.c:266: appends */
    data_block_resize(ob, base + cap);
    static uint8_t dummy;
    int rc = crgpu_dict_decode_blocks(c, g_shim_dict, s, &zero, &nin, 1, cap ? ob->m_data + base : &dummy, &zero, &cap, &produced);
    if (rc != CRGPU_OK) { data_block_resize(ob, base); shim_fail(rc, "dictionary_decode", rc == CRGPU_E_CORRUPT ? "malformed block" : crgpu_last_error(c)); return; }
    data_block_resize(ob, base + produced);
    if (fpout_sync) {                                      /* cr-diccode.c:274-277 */
        fwrite(ob->m_data, 1, ob->m_size, fpout_sync);
        data_block_resize(ob, 0);
    }
}

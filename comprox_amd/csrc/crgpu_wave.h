/*
 * comprox_amd/csrc/crgpu_wave.h — wave64 primitives for the gfx950 block codec.
 *
 * Everything here assumes blockDim.x == 64 (one wavefront per workgroup) and that all 64 lanes
 * are active at the call site. Cross-lane sums and scans use DPP row shifts / row broadcasts
 * (6 VALU instructions for a 64-lane inclusive scan), not LDS.
 */
#ifndef CRGPU_WAVE_H
#define CRGPU_WAVE_H

#include "crgpu_device.h"

#define CR_DEV __device__ __forceinline__

typedef uint32_t __attribute__((aligned(1))) cr_u32u;   /* unaligned views of byte streams */
typedef u64      __attribute__((aligned(1))) cr_u64u;

CR_DEV uint32_t cr_lane() { return threadIdx.x & 63u; }
CR_DEV uint32_t cr_wave_id() { return threadIdx.x >> 6; }
CR_DEV uint32_t cr_uni(uint32_t v) { return (uint32_t)__builtin_amdgcn_readfirstlane((int)v); }
CR_DEV uint32_t cr_lane_get(uint32_t v, uint32_t l) { return (uint32_t)__builtin_amdgcn_readlane((int)v, (int)l); }
CR_DEV u64 cr_lane_get64(u64 v, uint32_t l) {
    return ((u64)cr_lane_get((uint32_t)(v >> 32), l) << 32) | cr_lane_get((uint32_t)v, l);
}
CR_DEV u64 cr_ballot(bool p) { return __ballot(p); }

/* inclusive prefix sum over the 64 lanes */
CR_DEV uint32_t cr_scan_incl(uint32_t v) {
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x111, 0xf, 0xf, false);   /* row_shr:1 */
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x112, 0xf, 0xf, false);   /* row_shr:2 */
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x114, 0xf, 0xf, false);   /* row_shr:4 */
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x118, 0xf, 0xf, false);   /* row_shr:8 */
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x142, 0xa, 0xf, false);   /* row_bcast:15 */
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x143, 0xc, 0xf, false);   /* row_bcast:31 */
    return v;
}
/* sum over the 64 lanes, returned uniformly */
CR_DEV uint32_t cr_sum(uint32_t v) { return cr_lane_get(cr_scan_incl(v), 63); }

/* sum of the four bytes packed in w */
CR_DEV uint32_t cr_bytesum(uint32_t w) { return __builtin_amdgcn_sad_u8(w, 0u, 0u); }

/* mask selecting the bytes of lane `lane`'s word (bytes 4*lane .. 4*lane+3 of a 256-entry table)
 * whose table index is < limit */
CR_DEV uint32_t cr_mask_below(uint32_t lane, uint32_t limit) {
    int k = (int)limit - (int)(lane * 4u);
    if (k <= 0) return 0u;
    if (k >= 4) return 0xFFFFFFFFu;
    return (1u << (8 * k)) - 1u;
}

/* byte `idx` (uniform) of a 256-entry table held one word per lane */
CR_DEV uint32_t cr_table_byte(uint32_t w, uint32_t idx) {
    return (cr_lane_get(w, idx >> 2) >> ((idx & 3u) * 8u)) & 0xffu;
}

/* 0x01 in every byte of w that equals 1, then counted */
CR_DEV uint32_t cr_count_ones_bytes(uint32_t w) {
    return (uint32_t)((w & 0xffu) == 1u) + (uint32_t)(((w >> 8) & 0xffu) == 1u) +
           (uint32_t)(((w >> 16) & 0xffu) == 1u) + (uint32_t)((w >> 24) == 1u);
}

/* wave-wide fill of `bytes` (multiple of 16, 16-byte aligned) with a 32-bit pattern */
CR_DEV void cr_fill(uint8_t* dst, u64 bytes, uint32_t pattern) {
    uint4 v = make_uint4(pattern, pattern, pattern, pattern);
    for (u64 i = (u64)cr_lane() * 16u; i < bytes; i += 16u * CRGPU_WAVE)
        *reinterpret_cast<uint4*>(dst + i) = v;
}

/* previous lane's value (lane 0 receives `fill`): DPP wave_shr:1 */
CR_DEV uint32_t cr_shift_up1(uint32_t v, uint32_t fill) {
    return (uint32_t)__builtin_amdgcn_update_dpp((int)fill, (int)v, 0x138, 0xf, 0xf, false);
}

/* next lane's value (lane 63 receives `fill`): DPP wave_shl:1 */
CR_DEV uint32_t cr_shift_down1(uint32_t v, uint32_t fill) {
    return (uint32_t)__builtin_amdgcn_update_dpp((int)fill, (int)v, 0x130, 0xf, 0xf, false);
}

CR_DEV u64 cr_ld64(const u64* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
CR_DEV uint32_t cr_ld32(const uint32_t* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
CR_DEV void cr_st64(u64* p, u64 v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
CR_DEV void cr_st32(uint32_t* p, uint32_t v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

/* order the wave's own global/LDS traffic: earlier stores and atomics are performed before later
 * loads issue. Lanes of one wave run in lockstep, so this is a counter wait (s_waitcnt), never an
 * s_barrier — safe inside wave-specialised code of a multi-wave workgroup. */
CR_DEV void cr_wave_sync() {
    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
    __builtin_amdgcn_wave_barrier();
}

/* Wait for every outstanding vector-memory operation of this wave, as a real S_WAITCNT that the
 * compiler's wait-count pass sees. Used at the end of RARE refill branches so that the registers
 * they load are not "maybe pending" at the join — otherwise the join gets a vmcnt(0) that would
 * also drain the software-pipelined model loads on every pass through the common path. */
CR_DEV void cr_drain_loads() { __builtin_amdgcn_s_waitcnt(0x0F70); }

/* Two waves that share a SIMD do not share it evenly: at equal priority the OLDER one issues nearly unimpeded and the younger one
 * gets what is left (MI355X_MICROARCH.md, two waves per SIMD; tools/coissue_probe.hip: 4.1 against 8.2 clocks per scalar
 * instruction). The one-wave-per-block kernels run 1 526 waves on 1 024 SIMDs and last as long as their slowest block, i.e. as
 * long as the younger waves of the 502 pairs. Called every so often with a counter that follows the wave's progress, this lets
 * the two take turns: the priority flips between `base + 1` and `base`, in opposite phase for the even and the odd wave slots of
 * a SIMD (HW_ID bit 0). */
template <int BASE>
CR_DEV void cr_take_turns(uint32_t phase) {
    const uint32_t slot = (uint32_t)__builtin_amdgcn_s_getreg(4);        /* hwreg(HW_REG_HW_ID, 0, 1): bit 0 of the wave's slot */
    if ((slot ^ phase) & 1u) __builtin_amdgcn_s_setprio(BASE + 1); else __builtin_amdgcn_s_setprio(BASE);
}

CR_DEV uint32_t cr_log2_ceil_pow2(uint32_t want, uint32_t lo, uint32_t hi) {   /* smallest 2^k >= want within [lo,hi] */
    uint32_t c = lo;
    while (c < want && c < hi) c <<= 1;
    return c;
}

#endif

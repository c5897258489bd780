// Co-residency probe for gfx950: what do two (or four) issue-bound waves on ONE SIMD cost each other, by instruction
// mix? One workgroup of W waves on one CU (waves land on the four SIMDs in turn), every wave runs the same chain and
// reports its cycles and the SIMD it ran on (HW_REG_HW_ID). The decoder's step is ~75 % scalar; at 1 526 resident
// blocks half the SIMDs carry two chains.
// build: hipcc --offload-arch=gfx950 -O3 tools/coissue_probe.hip -o /tmp/coissue_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

#define REP(n, body) asm volatile(".rept " #n "\n" body "\n.endr" : "+v"(v), "+v"(u), "+s"(s), "+s"(t) :: "vcc", "scc")
#define PROBE(name, n, body) \
__global__ void name(uint64_t* out, int iters) { \
    uint32_t v = threadIdx.x, u = threadIdx.x * 3u, s = 1, t = 2; \
    __syncthreads(); \
    uint64_t c0 = __builtin_amdgcn_s_memtime(); \
    for (int k = 0; k < iters; k++) { REP(n, body); } \
    uint64_t c1 = __builtin_amdgcn_s_memtime(); \
    uint32_t hw; asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw)); \
    if ((threadIdx.x & 63) == 0) { uint64_t* o = out + (threadIdx.x >> 6) * 4; o[0] = c1 - c0; o[1] = hw; o[2] = v + u + s + t; } \
}
// 256 instructions per repetition in every mix
PROBE(k_salu,  256, "s_add_u32 %2, %2, %3")
PROBE(k_valu,  256, "v_add_u32 %0, %0, %1")
PROBE(k_mix75,  64, "s_add_u32 %2, %2, %3\n s_add_u32 %3, %3, 1\n s_add_u32 %2, %2, %3\n v_add_u32 %0, %0, %1")
PROBE(k_mix50, 128, "s_add_u32 %2, %2, %3\n v_add_u32 %0, %0, %1")
PROBE(k_mix25,  64, "v_add_u32 %0, %0, %1\n v_add_u32 %1, 1, %1\n v_add_u32 %0, %0, %1\n s_add_u32 %2, %2, %3")
// the same with the scalar work moved to the vector unit through a uniform value (v_readfirstlane keeps a scalar consumer fed)
PROBE(k_mix50r, 64, "s_add_u32 %2, %2, %3\n v_add_u32 %0, %0, %1\n v_readfirstlane_b32 %3, %1\n v_add_u32 %1, 1, %1")

template <typename K> static void run(const char* name, K k, uint64_t* d_out, int waves) {
    const int iters = 200;
    hipLaunchKernelGGL(k, dim3(1), dim3(64 * waves), 0, 0, d_out, iters);
    hipLaunchKernelGGL(k, dim3(1), dim3(64 * waves), 0, 0, d_out, iters);
    (void)hipDeviceSynchronize();
    uint64_t h[16 * 4]; (void)hipMemcpy(h, d_out, sizeof(uint64_t) * 4 * waves, hipMemcpyDeviceToHost);
    double n = (double)iters * 256;
    double worst = 0, sum = 0; int per_simd[4] = {0, 0, 0, 0};
    for (int w = 0; w < waves; w++) {
        const double c = h[w * 4] / n; sum += c; if (c > worst) worst = c;
        per_simd[(h[w * 4 + 1] >> 4) & 3]++;
    }
    printf("%-8s waves %2d (per SIMD %d %d %d %d)  cycles/instr per wave: mean %5.2f worst %5.2f\n", name, waves,
           per_simd[0], per_simd[1], per_simd[2], per_simd[3], sum / waves, worst);
}
int main() {
    uint64_t* d_out; (void)hipMalloc(&d_out, sizeof(uint64_t) * 64);
    const int counts[4] = {1, 4, 8, 16};
    for (int i = 0; i < 4; i++) {
        const int w = counts[i];
        run("salu", k_salu, d_out, w);
        run("valu", k_valu, d_out, w);
        run("mix75", k_mix75, d_out, w);
        run("mix50", k_mix50, d_out, w);
        run("mix25", k_mix25, d_out, w);
        run("mix50r", k_mix50r, d_out, w);
    }
    return 0;
}

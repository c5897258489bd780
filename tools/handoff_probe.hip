// What does it cost to hand a value from one wave of a workgroup to another through LDS? (task 1 step A: a
// decoder workgroup with a coder wave and a helper wave lives or dies by this number.)
// Two waves of one workgroup (they land on different SIMDs of the CU) play ping-pong: wave 0 writes 64 dwords (one per lane)
// and a sequence word, wave 1 waits for the word, reads the 64 dwords, writes 64 dwords + its own word back, wave 0 waits
// for that. Time per ONE-WAY hand-off = loop time / (2 x iterations), in ns and shader clocks, for 1 / 256 / 1 526 / 3 052
// such workgroups resident (the decoder's batch is 1 526).
//   mode 0  tight poll: ds_read_b32 / s_waitcnt lgkmcnt(0) / v_cmp / s_cbranch
//   mode 1  the same with s_sleep 1 in the loop (less LDS traffic from the pollers)
//   mode 2  s_barrier instead of the sequence word (the writer waits for its LDS writes, then both pass a barrier)
//   mode 3  tight poll, sequence word only (no 64-dword payload): the floor of a flag hand-off
// build: hipcc --offload-arch=gfx950 -O3 tools/handoff_probe.hip -o tools/bin/handoff_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>

__device__ __forceinline__ void lds_write(uint32_t addr, uint32_t v) { asm volatile("ds_write_b32 %0, %1" :: "v"(addr), "v"(v) : "memory"); }
__device__ __forceinline__ uint32_t lds_read_wait(uint32_t addr) {
    uint32_t v; asm volatile("ds_read_b32 %0, %1\n s_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(addr) : "memory"); return v;
}
template <int SLEEP>
__device__ __forceinline__ void poll(uint32_t addr, uint32_t want) {
    uint32_t v;
    if (SLEEP)
        asm volatile(".Lp_%=:\n ds_read_b32 %0, %1\n s_waitcnt lgkmcnt(0)\n v_cmp_ne_u32 vcc, %0, %2\n s_cbranch_vccz .Ld_%=\n s_sleep 1\n s_branch .Lp_%=\n.Ld_%=:"
                     : "=&v"(v) : "v"(addr), "v"(want) : "vcc", "memory");
    else
        asm volatile(".Lp_%=:\n ds_read_b32 %0, %1\n s_waitcnt lgkmcnt(0)\n v_cmp_ne_u32 vcc, %0, %2\n s_cbranch_vccnz .Lp_%="
                     : "=&v"(v) : "v"(addr), "v"(want) : "vcc", "memory");
}

template <int MODE>
__global__ void k_pingpong(int iters, uint64_t* out) {
    extern __shared__ uint32_t sm[];                   // [0,256) payload 0 -> 1, [256,512) payload 1 -> 0, 512 / 516 the sequence words
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    if (threadIdx.x < 2) sm[128 + threadIdx.x] = 0;
    __syncthreads();
    const uint32_t pay0 = lane * 4u, pay1 = 256u + lane * 4u, seq0 = 512u, seq1 = 516u;
    uint32_t x = lane;
    const uint64_t t0 = wall_clock64();
    const uint64_t c0 = __builtin_amdgcn_s_memtime();
    for (int k = 1; k <= iters; k++) {
        if (wave == 0) {
            if (MODE != 3) lds_write(pay0, x);
            if (MODE == 2) { asm volatile("s_waitcnt lgkmcnt(0)\n s_barrier\n s_barrier" ::: "memory"); }
            else { lds_write(seq0, (uint32_t)k); if (MODE == 1) poll<1>(seq1, (uint32_t)k); else poll<0>(seq1, (uint32_t)k); }
            if (MODE != 3) x = lds_read_wait(pay1) + 1u;
        } else {
            if (MODE == 2) asm volatile("s_barrier" ::: "memory");
            else if (MODE == 1) poll<1>(seq0, (uint32_t)k); else poll<0>(seq0, (uint32_t)k);
            if (MODE != 3) { x = lds_read_wait(pay0) + 1u; lds_write(pay1, x); }
            if (MODE == 2) asm volatile("s_waitcnt lgkmcnt(0)\n s_barrier" ::: "memory");
            else lds_write(seq1, (uint32_t)k);
        }
    }
    const uint64_t c1 = __builtin_amdgcn_s_memtime();
    const uint64_t t1 = wall_clock64();
    if (threadIdx.x == 0) { out[blockIdx.x * 3] = t1 - t0; out[blockIdx.x * 3 + 1] = c1 - c0; out[blockIdx.x * 3 + 2] = x; }
}

typedef void (*kern_t)(int, uint64_t*);
int main() {
    uint64_t* d_out; (void)hipMalloc(&d_out, 4096 * 24);
    const kern_t ks[] = { k_pingpong<0>, k_pingpong<1>, k_pingpong<2>, k_pingpong<3> };
    const char* names[] = { "tight poll + 64 dwords", "poll with s_sleep 1 + 64 dwords", "s_barrier + 64 dwords", "tight poll, flag only" };
    const int counts[] = { 1, 256, 1526, 3052 };
    const int iters = 20000;
    printf("one-way hand-off between two waves of a workgroup through LDS (ping-pong, %d round trips)\n", iters);
    printf("%-36s %6s %10s %10s\n", "mode", "wgs", "ns", "clocks");
    for (int m = 0; m < 4; m++)
        for (int n : counts) {
            for (int rep = 0; rep < 2; rep++) { hipLaunchKernelGGL(ks[m], dim3(n), dim3(128), 1024, 0, iters, d_out); (void)hipDeviceSynchronize(); }
            std::vector<uint64_t> h(n * 3);
            (void)hipMemcpy(h.data(), d_out, n * 24, hipMemcpyDeviceToHost);
            double ns = 0, clk = 0;
            for (int i = 0; i < n; i++) { ns += h[i * 3] * 10.0; clk += (double)h[i * 3 + 1]; }
            printf("%-36s %6d %10.1f %10.0f\n", names[m], n, ns / n / (2.0 * iters), clk / n / (2.0 * iters));
            fflush(stdout);
        }
    return 0;
}

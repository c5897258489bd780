// Latency probes for gfx950: dependent-load chains at several working-set sizes, store acknowledge,
// atomic round trip. One wave; numbers are shader clocks (s_memtime) and ns (wall clock 100 MHz).
// build: hipcc --offload-arch=gfx950 -O3 tools/lat_probe.hip -o /tmp/lat_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
#include <numeric>
#include <random>
#include <algorithm>

__global__ void k_chase(const uint32_t* next, uint32_t start, int iters, uint64_t* out, int mode) {
    uint32_t i = start;
    uint64_t t0 = wall_clock64();
    uint64_t c0 = __builtin_amdgcn_s_memtime();
    for (int k = 0; k < iters; k++) {
        if (mode == 0) i = next[i];
        else i = __hip_atomic_load(next + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    uint64_t c1 = __builtin_amdgcn_s_memtime();
    uint64_t t1 = wall_clock64();
    if (threadIdx.x == 0) { out[0] = c1 - c0; out[1] = t1 - t0; out[2] = i; }
}
__global__ void k_store_ack(uint32_t* p, int iters, uint64_t* out) {
    uint64_t t0 = wall_clock64();
    for (int k = 0; k < iters; k++) {
        p[(k * 4099u) & 0xfffffu] = k;
        __builtin_amdgcn_s_waitcnt(0x0F70);   // vmcnt(0)
    }
    uint64_t t1 = wall_clock64();
    if (threadIdx.x == 0) out[1] = t1 - t0;
}
__global__ void k_atomic(unsigned long long* p, int iters, uint64_t* out) {
    uint64_t t0 = wall_clock64();
    unsigned long long v = 0;
    for (int k = 0; k < iters; k++) v += atomicCAS(p + ((k * 4099u + (uint32_t)v) & 0xfffffu), 0ull, 1ull + k);
    uint64_t t1 = wall_clock64();
    if (threadIdx.x == 0) { out[1] = t1 - t0; out[2] = v; }
}
int main() {
    uint64_t* d_out; hipMalloc(&d_out, 64);
    const size_t sizes[] = {16u << 10, 1u << 20, 3u << 20, 16u << 20, 128u << 20, 1024u << 20};
    for (size_t bytes : sizes) {
        size_t n = bytes / 64;                       // one hop per 64-byte line
        std::vector<uint32_t> perm(n); std::iota(perm.begin(), perm.end(), 0u);
        std::mt19937 rng(1); std::shuffle(perm.begin(), perm.end(), rng);
        std::vector<uint32_t> next(bytes / 4, 0);
        for (size_t k = 0; k < n; k++) next[(size_t)perm[k] * 16] = perm[(k + 1) % n] * 16;
        uint32_t* d; hipMalloc(&d, bytes); hipMemcpy(d, next.data(), bytes, hipMemcpyHostToDevice);
        for (int mode = 0; mode < 2; mode++) {
            int iters = 20000;
            hipLaunchKernelGGL(k_chase, dim3(1), dim3(64), 0, 0, d, perm[0] * 16, iters, d_out, mode);   // warm
            hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
            hipEventRecord(e0, 0);
            hipLaunchKernelGGL(k_chase, dim3(1), dim3(64), 0, 0, d, perm[iters % n] * 16, iters, d_out, mode);
            hipEventRecord(e1, 0); hipEventSynchronize(e1);
            float ms = 0; hipEventElapsedTime(&ms, e0, e1);
            uint64_t h[3]; hipMemcpy(h, d_out, 24, hipMemcpyDeviceToHost);
            printf("chase %8zu KiB %s: %.1f ns/hop  (%.0f memtime ticks/hop; host events %.1f ns/hop)\n", bytes >> 10, mode ? "sc1 " : "plain", h[1] * 10.0 / iters, (double)h[0] / iters, ms * 1e6 / iters);
        }
        hipFree(d);
    }
    uint32_t* p; hipMalloc(&p, 4u << 20); hipMemset(p, 0, 4u << 20);
    hipLaunchKernelGGL(k_store_ack, dim3(1), dim3(64), 0, 0, p, 20000, d_out);
    hipLaunchKernelGGL(k_store_ack, dim3(1), dim3(64), 0, 0, p, 20000, d_out);
    uint64_t h[3]; hipMemcpy(h, d_out, 24, hipMemcpyDeviceToHost);
    printf("store + vmcnt(0): %.1f ns\n", h[1] * 10.0 / 20000);
    unsigned long long* q; hipMalloc(&q, 8u << 20); hipMemset(q, 0, 8u << 20);
    hipLaunchKernelGGL(k_atomic, dim3(1), dim3(64), 0, 0, q, 20000, d_out);
    hipMemcpy(h, d_out, 24, hipMemcpyDeviceToHost);
    printf("atomicCAS round trip: %.1f ns\n", h[1] * 10.0 / 20000);
    return 0;
}

// Does a dependent load get slower because MANY waves are resident, or because they touch many PAGES?
// N single-wave workgroups each chase K dependent random loads inside their own region of R bytes that starts at
// wg * stride (memory is zero: the loaded value is added to the next address, which keeps the chain dependent).
// build: hipcc --offload-arch=gfx950 -O3 tools/tlb_probe.hip -o tools/bin/tlb_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>

__global__ void k_probe(const uint8_t* base, uint64_t stride, uint32_t region_lines, int iters, uint64_t* out) {
    const uint8_t* p = base + (uint64_t)blockIdx.x * stride;
    uint32_t state = blockIdx.x * 2654435761u + 12345u;
    uint32_t v = 0;
    const uint64_t t0 = wall_clock64();
    for (int k = 0; k < iters; k++) {
        state = state * 1664525u + 1013904223u + v;
        const uint32_t line = (uint32_t)(((uint64_t)(state >> 4) * region_lines) >> 28);   // uniform in [0, region_lines)
        v = __hip_atomic_load(reinterpret_cast<const uint32_t*>(p + (uint64_t)line * 128u), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    const uint64_t t1 = wall_clock64();
    if (threadIdx.x == 0) { out[blockIdx.x * 2] = t1 - t0; out[blockIdx.x * 2 + 1] = v; }
}

int main() {
    const uint64_t total = 56ull << 30;
    uint8_t* d; if (hipMalloc(&d, total) != hipSuccess) { printf("alloc failed\n"); return 1; }
    hipMemset(d, 0, total);
    uint64_t* d_out; hipMalloc(&d_out, 8192 * 16);
    struct Case { int n; uint64_t region, stride; const char* what; };
    const Case cases[] = {
        {1,    32ull << 20, 35ull << 20, "one wave, 32 MiB region"},
        {1526, 32ull << 20, 35ull << 20, "1526 waves, 32 MiB regions 35 MiB apart (the decoder's arenas)"},
        {1526,  2ull << 20, 35ull << 20, "1526 waves,  2 MiB regions 35 MiB apart"},
        {1526,  2ull << 20,  2ull << 20, "1526 waves,  2 MiB regions back to back (3 GiB)"},
        {1526, 256ull << 10, 35ull << 20, "1526 waves, 256 KiB regions 35 MiB apart"},
        {1526, 32ull << 20, 0,           "1526 waves, one shared 32 MiB region"},
        {256,  32ull << 20, 35ull << 20, "256 waves, 32 MiB regions 35 MiB apart"},
        {4096, 12ull << 20, 13ull << 20, "4096 waves, 12 MiB regions 13 MiB apart"},
    };
    const int iters = 4000;
    for (const Case& c : cases) {
        for (int rep = 0; rep < 2; rep++) {
            hipLaunchKernelGGL(k_probe, dim3(c.n), dim3(64), 0, 0, d, c.stride, (uint32_t)(c.region / 128), iters, d_out);
            hipDeviceSynchronize();
        }
        std::vector<uint64_t> h(c.n * 2);
        hipMemcpy(h.data(), d_out, c.n * 16, hipMemcpyDeviceToHost);
        double sum = 0, mx = 0;
        for (int i = 0; i < c.n; i++) { double t = h[i * 2] * 10.0 / iters; sum += t; if (t > mx) mx = t; }
        printf("%-66s %7.0f ns per dependent load (slowest wave %7.0f)\n", c.what, sum / c.n, mx);
        fflush(stdout);
    }
    return 0;
}

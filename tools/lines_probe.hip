// What does a decode step's memory shape cost at the decoder's load? N single-wave workgroups, each in its own 32 MiB
// region 35 MiB from the next (the decoder's arenas), chase K dependent hops; a hop LOADS `nl` random lines of its region
// at once (lanes 0..nl-1, one dword each; the hop depends on all of them) and STORES one dword into `ns` other random
// lines (not waited for), like a decode step's model loads and write-backs.
// build: hipcc --offload-arch=gfx950 -O3 tools/lines_probe.hip -o tools/bin/lines_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>

__global__ void k_probe(uint8_t* base, uint64_t stride, uint32_t region_lines, int iters, int nl, int ns, uint64_t* out) {
    uint8_t* p = base + (uint64_t)blockIdx.x * stride;
    const uint32_t lane = threadIdx.x;
    uint32_t state = blockIdx.x * 2654435761u + 12345u;
    uint32_t v = 0;
    const uint64_t t0 = wall_clock64();
    for (int k = 0; k < iters; k++) {
        state = state * 1664525u + 1013904223u + v;
        const uint32_t mine = (state ^ (lane * 0x9e3779b9u)) * 2246822519u;
        const uint32_t line = (uint32_t)(((uint64_t)(mine >> 4) * region_lines) >> 28);
        uint32_t got = 0;
        if ((int)lane < nl) got = __hip_atomic_load(reinterpret_cast<const uint32_t*>(p + (uint64_t)line * 128u), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if ((int)lane >= 32 && (int)lane < 32 + ns) __hip_atomic_store(reinterpret_cast<uint32_t*>(p + (uint64_t)line * 128u + 64u), 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        // the hop depends on every loaded line
        for (int o = 32; o >= 1; o >>= 1) got += __shfl_xor((int)got, o);
        v = got;
    }
    const uint64_t t1 = wall_clock64();
    if (threadIdx.x == 0) { out[blockIdx.x * 2] = t1 - t0; out[blockIdx.x * 2 + 1] = v; }
}

int main() {
    const int n = 1526;
    const uint64_t stride = 35ull << 20, region = 32ull << 20;
    const uint64_t total = (uint64_t)n * stride + region;
    uint8_t* d; if (hipMalloc(&d, total) != hipSuccess) { printf("alloc failed\n"); return 1; }
    (void)hipMemset(d, 0, total);
    uint64_t* d_out; (void)hipMalloc(&d_out, 8192 * 16);
    const int shapes[][2] = {{1, 0}, {2, 0}, {4, 0}, {6, 0}, {1, 1}, {2, 2}, {4, 3}, {6, 4}, {3, 2}, {2, 1}};
    const int counts[] = {1526, 256, 1};
    const int iters = 3000;
    for (int ci = 0; ci < 3; ci++) {
        for (const auto& sh : shapes) {
            for (int rep = 0; rep < 2; rep++) {
                hipLaunchKernelGGL(k_probe, dim3(counts[ci]), dim3(64), 0, 0, d, stride, (uint32_t)(region / 128), iters, sh[0], sh[1], d_out);
                (void)hipDeviceSynchronize();
            }
            std::vector<uint64_t> h(counts[ci] * 2);
            (void)hipMemcpy(h.data(), d_out, counts[ci] * 16, hipMemcpyDeviceToHost);
            double sum = 0, mx = 0;
            for (int i = 0; i < counts[ci]; i++) { double t = h[i * 2] * 10.0 / iters; sum += t; if (t > mx) mx = t; }
            printf("%5d waves, %d lines loaded + %d lines stored per hop: %7.0f ns per hop (slowest wave %7.0f)\n", counts[ci], sh[0], sh[1], sum / counts[ci], mx);
            fflush(stdout);
        }
    }
    return 0;
}

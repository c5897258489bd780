// Which number does a decode step pay for its model fetch: tlb_probe's 393 ns or lines_probe's 573 ns (VERDICT r3, weak #3a)?
// Both chase dependent random loads in 1 526 per-wave regions; they differ in what a HOP does besides the load:
// tlb_probe loads ONE dword at a wave-uniform address and feeds it straight into the next address; lines_probe loads with
// lane 0 only and then reduces over the wave with six dependent __shfl_xor (ds_bpermute round trips) before the next address
// exists. This probe runs both hop shapes and the ones in between, each next to the same hop WITHOUT its load (what the hop
// costs when memory is free), at 1 526 / 256 / 1 waves:
//   shape 0  uniform address, all lanes load the same dword                         (tlb_probe)
//   shape 1  lane 0 loads a dword, six __shfl_xor reduce                            (lines_probe, nl = 1)
//   shape 2  lane 0 loads a dword, v_readfirstlane                                  (lines_probe without the reduction)
//   shape 3  lanes 0..7 / 0..15 / 0..31 load consecutive dwords of one line (32 / 64 / 128 bytes), v_readfirstlane
//   shape 4  64 lanes load a u16 each of one line (the decoder's node line), v_readfirstlane
// and each of 2 / 3 / 4 with the decoder's store mix (a dword into each of 3 other random lines of the region, not waited for).
// build: hipcc --offload-arch=gfx950 -O3 tools/hop_probe.hip -o tools/bin/hop_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>

template <int SHAPE, int WIDTH, int NS, int LOAD>
__global__ void k_hop(uint8_t* base, uint64_t stride, uint32_t region_lines, int iters, uint64_t* out) {
    uint8_t* p = base + (uint64_t)blockIdx.x * stride;
    const uint32_t lane = threadIdx.x;
    uint32_t state = blockIdx.x * 2654435761u + 12345u;
    uint32_t v = 0;
    const uint64_t t0 = wall_clock64();
    for (int k = 0; k < iters; k++) {
        state = state * 1664525u + 1013904223u + v;
        const uint32_t line = (uint32_t)(((uint64_t)(state >> 4) * region_lines) >> 28);       // wave-uniform
        uint8_t* a = p + (uint64_t)line * 128u;
        uint32_t got = 0;
        if (SHAPE == 0) {
            if (LOAD) got = __hip_atomic_load(reinterpret_cast<const uint32_t*>(a), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            v = got;
        } else if (SHAPE == 1) {
            if (LOAD && lane == 0) got = __hip_atomic_load(reinterpret_cast<const uint32_t*>(a), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            for (int o = 32; o >= 1; o >>= 1) got += __shfl_xor((int)got, o);
            v = got;
        } else if (SHAPE == 2 || SHAPE == 3) {
            const uint32_t lanes = SHAPE == 2 ? 1u : (uint32_t)WIDTH / 4u;
            if (LOAD && lane < lanes) got = __hip_atomic_load(reinterpret_cast<const uint32_t*>(a) + lane, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            v = __builtin_amdgcn_readfirstlane(got);
        } else {
            if (LOAD) got = __hip_atomic_load(reinterpret_cast<const uint16_t*>(a) + lane, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            v = __builtin_amdgcn_readfirstlane(got);
        }
        if (NS) {
            const uint32_t mine = (state ^ (lane * 0x9e3779b9u)) * 2246822519u;
            const uint32_t sl = (uint32_t)(((uint64_t)(mine >> 4) * region_lines) >> 28);
            if (lane >= 32 && lane < 32 + NS) __hip_atomic_store(reinterpret_cast<uint32_t*>(p + (uint64_t)sl * 128u + 64u), 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
    const uint64_t t1 = wall_clock64();
    if (threadIdx.x == 0) { out[blockIdx.x * 2] = t1 - t0; out[blockIdx.x * 2 + 1] = v; }
}

typedef void (*kern_t)(uint8_t*, uint64_t, uint32_t, int, uint64_t*);
struct Shape { const char* what; kern_t with, without; };

#define SH(what, S, W, NS) { what, k_hop<S, W, NS, 1>, k_hop<S, W, NS, 0> }

static double run(kern_t k, int n, uint8_t* d, uint64_t stride, uint64_t region, int iters, uint64_t* d_out, double* slowest) {
    for (int rep = 0; rep < 2; rep++) {
        hipLaunchKernelGGL(k, dim3(n), dim3(64), 0, 0, d, stride, (uint32_t)(region / 128), iters, d_out);
        (void)hipDeviceSynchronize();
    }
    std::vector<uint64_t> h(n * 2);
    (void)hipMemcpy(h.data(), d_out, n * 16, hipMemcpyDeviceToHost);
    double sum = 0, mx = 0;
    for (int i = 0; i < n; i++) { double t = h[i * 2] * 10.0 / iters; sum += t; if (t > mx) mx = t; }
    *slowest = mx;
    return sum / n;
}

int main() {
    const int nmax = 1526;
    const uint64_t stride = 35ull << 20, region = 32ull << 20;
    const uint64_t total = (uint64_t)nmax * stride + region;
    uint8_t* d; if (hipMalloc(&d, total) != hipSuccess) { printf("alloc failed\n"); return 1; }
    (void)hipMemset(d, 0, total);
    uint64_t* d_out; (void)hipMalloc(&d_out, 8192 * 16);
    const Shape shapes[] = {
        SH("0 uniform dword (tlb_probe)", 0, 4, 0),
        SH("1 lane-0 dword + 6 shfl_xor (lines_probe)", 1, 4, 0),
        SH("2 lane-0 dword + readfirstlane", 2, 4, 0),
        SH("3 32 B of a line + readfirstlane", 3, 32, 0),
        SH("3 64 B of a line + readfirstlane", 3, 64, 0),
        SH("3 128 B of a line + readfirstlane", 3, 128, 0),
        SH("4 64 x u16 of a line + readfirstlane", 4, 128, 0),
        SH("2 lane-0 dword, 3 lines stored", 2, 4, 3),
        SH("3 64 B of a line, 3 lines stored", 3, 64, 3),
        SH("3 128 B of a line, 3 lines stored", 3, 128, 3),
        SH("4 64 x u16 of a line, 3 lines stored", 4, 128, 3),
    };
    const int counts[] = {1526, 256, 1};
    const int iters = 3000;
    printf("%-48s %6s %10s %10s %10s %10s\n", "hop shape", "waves", "ns/hop", "slowest", "no-load", "memory");
    for (int ci = 0; ci < 3; ci++) {
        for (const Shape& s : shapes) {
            double mx, mx0;
            const double t = run(s.with, counts[ci], d, stride, region, iters, d_out, &mx);
            const double t0 = run(s.without, counts[ci], d, stride, region, iters, d_out, &mx0);
            printf("%-48s %6d %10.0f %10.0f %10.0f %10.0f\n", s.what, counts[ci], t, mx, t0, t - t0);
            fflush(stdout);
        }
    }
    return 0;
}

// Does asking for a line EARLY shorten the dependent load that follows? (round 5: the decoder's successor-line prefetch,
// crgpu_rop5.h -DCR_V5_PF, bought nothing — this probe isolates the memory system's side of that.)
// N single-wave workgroups chase dependent random lines in their own 32 MiB regions (the decoder's arenas), as
// tools/hop_probe.hip does. A hop = [touch] -> s_sleep (the search's ~450 / ~900 clocks) -> the real load (64 x u16 of the
// line, the node line's shape) -> s_waitcnt vmcnt(0). Touch variants:
//   0 none      1 one lane reads a byte of the SAME line (the prefetch)      2 one lane reads a byte of ANOTHER random line
//   3 eight lanes read a byte each of eight random lines, one of them the right one (several candidates)
//   4 the touch comes from a SECOND wave of the workgroup (address handed over through LDS), wave 0 never waits for it
// build: hipcc --offload-arch=gfx950 -O3 tools/touch_probe.hip -o tools/bin/touch_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>

template <int TOUCH, int SLEEP>
__global__ void k_touch(uint8_t* base, uint64_t stride, uint32_t region_lines, int iters, uint64_t* out) {
    uint8_t* p = base + (uint64_t)blockIdx.x * stride;
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    __shared__ volatile uint32_t box[2];
    uint32_t state = blockIdx.x * 2654435761u + 12345u;
    uint32_t v = 0;
    if (threadIdx.x == 0) { box[0] = 0; box[1] = 0; }
    __syncthreads();                                   // (LDS keeps the last launch's stop word)
    if (TOUCH == 4 && wave == 1) {
        // the toucher: polls the mailbox, reads a byte of the line it names; ends on the stop word
        uint32_t seen = 0, dummy = 0;
        for (;;) {
            uint32_t seq = box[0];
            if (seq == 0xffffffffu) break;
            if (seq != seen) {
                seen = seq;
                const uint8_t* a = p + (uint64_t)box[1] * 128u;
                asm volatile("global_load_ubyte %0, %1, off" : "=v"(dummy) : "v"(a) : "memory");
            }
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if (lane == 0) out[blockIdx.x * 2 + 1] = dummy;
        return;
    }
    const uint64_t t0 = wall_clock64();
    for (int k = 0; k < iters; k++) {
        state = state * 1664525u + 1013904223u + v;
        const uint32_t line = (uint32_t)(((uint64_t)(state >> 4) * region_lines) >> 28);       // wave-uniform
        const uint8_t* a = p + (uint64_t)line * 128u;
        const uint16_t* ra = reinterpret_cast<const uint16_t*>(a) + lane;
        uint32_t got, dummy = 0;
        if (TOUCH == 4) {
            if (lane == 0) { box[1] = line; box[0] = (uint32_t)k + 1u; }
            asm volatile("s_sleep %[sl]\n global_load_ushort %[g], %[ra], off\n s_waitcnt vmcnt(0)"
                         : [g] "=&v"(got) : [ra] "v"(ra), [sl] "n"(SLEEP) : "memory");
        } else if (TOUCH == 0) {
            asm volatile("s_sleep %[sl]\n global_load_ushort %[g], %[ra], off\n s_waitcnt vmcnt(0)"
                         : [g] "=&v"(got) : [ra] "v"(ra), [sl] "n"(SLEEP) : "memory");
        } else {
            const uint8_t* ta = a;
            if (TOUCH == 2 || (TOUCH == 3 && lane != 5)) {
                const uint32_t mine = (state ^ ((lane + 1u) * 0x9e3779b9u)) * 2246822519u;
                ta = p + (uint64_t)(uint32_t)(((uint64_t)(mine >> 4) * region_lines) >> 28) * 128u;
            }
            const uint32_t lanes = TOUCH == 3 ? 8u : 1u;
            if (lane < lanes) asm volatile("global_load_ubyte %0, %1, off" : "=v"(dummy) : "v"(ta) : "memory");
            asm volatile("s_sleep %[sl]\n global_load_ushort %[g], %[ra], off\n s_waitcnt vmcnt(0)"
                         : [g] "=&v"(got), [d] "+v"(dummy) : [ra] "v"(ra), [sl] "n"(SLEEP) : "memory");
        }
        v = __builtin_amdgcn_readfirstlane(got) + (dummy & 0u);
    }
    const uint64_t t1 = wall_clock64();
    if (TOUCH == 4 && threadIdx.x == 0) box[0] = 0xffffffffu;
    if (threadIdx.x == 0) { out[blockIdx.x * 2] = t1 - t0; out[blockIdx.x * 2 + 1] = v; }
}

typedef void (*kern_t)(uint8_t*, uint64_t, uint32_t, int, uint64_t*);

static double run(kern_t k, int n, int threads, uint8_t* d, uint64_t stride, uint64_t region, int iters, uint64_t* d_out) {
    for (int rep = 0; rep < 2; rep++) {
        hipLaunchKernelGGL(k, dim3(n), dim3(threads), 0, 0, d, stride, (uint32_t)(region / 128), iters, d_out);
        (void)hipDeviceSynchronize();
    }
    std::vector<uint64_t> h(n * 2);
    (void)hipMemcpy(h.data(), d_out, n * 16, hipMemcpyDeviceToHost);
    double sum = 0;
    for (int i = 0; i < n; i++) sum += h[i * 2] * 10.0 / iters;
    return sum / n;
}

#define ROW(S) { S, { k_touch<0, S>, k_touch<1, S>, k_touch<2, S>, k_touch<3, S>, k_touch<4, S> } }
int main() {
    const int nmax = 1526;
    const uint64_t stride = 35ull << 20, region = 32ull << 20;
    const uint64_t total = (uint64_t)nmax * stride + region;
    uint8_t* d; if (hipMalloc(&d, total) != hipSuccess) { printf("alloc failed\n"); return 1; }
    (void)hipMemset(d, 0, total);
    uint64_t* d_out; (void)hipMalloc(&d_out, 8192 * 16);
    struct { int sleep; kern_t k[5]; } rows[] = { ROW(0), ROW(3), ROW(7), ROW(14), ROW(21) };
    const int counts[] = {1526, 256, 1};
    const int iters = 3000;
    printf("ns per hop: [touch] -> s_sleep n (64 n clocks) -> load 64 x u16 of a random line of the wave's 32 MiB region -> wait\n");
    printf("%6s %8s %10s %12s %12s %14s %16s\n", "waves", "s_sleep", "no touch", "same line", "other line", "8 lines (1 ok)", "second wave");
    for (int ci = 0; ci < 3; ci++)
        for (auto& r : rows) {
            printf("%6d %8d", counts[ci], r.sleep);
            for (int t = 0; t < 5; t++) printf(" %12.0f", run(r.k[t], counts[ci], t == 4 ? 128 : 64, d, stride, region, iters, d_out));
            printf("\n"); fflush(stdout);
        }
    return 0;
}

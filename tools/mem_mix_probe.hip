// What stretches the decoder's model fetch when 1 526 waves are resident: the number of waves, or what they WRITE?
// N single-wave workgroups; each iteration a wave loads one 256-byte line pair (lane * 4) at a random place of its own 8 MiB
// region (the address depends on the loaded value: a chain, like the decoder's context), waits for it (timed with s_memtime),
// then issues the step's other traffic and burns D dependent VALU instructions (the step's ~1 400 clocks of issue):
//   mix bit 0: store the line back (the node's write-back, 256 B)
//   mix bit 1: three one-lane dword stores at random places of a second 1 MiB region (LZP inserts, order-3 entry, output)
//   mix bit 2: three one-lane dword loads at random places of that region (the lookups)
//   mix bit 3: the stores of bit 1 widened to 16 lanes x 4 B (one 64-byte piece each)
// build: hipcc --offload-arch=gfx950 -O3 tools/mem_mix_probe.hip -o tools/bin/mem_mix_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>

__global__ void k_probe(uint8_t* base, uint64_t stride, int iters, int mix, int delay, uint64_t* out) {
    uint8_t* p = base + (uint64_t)blockIdx.x * stride;
    uint8_t* q = p + (9ull << 20);
    const uint32_t lane = threadIdx.x;
    uint32_t state = blockIdx.x * 2654435761u + 12345u;
    uint32_t v = 0, acc = 0;
    uint64_t lat = 0;
    const uint64_t t0 = __builtin_amdgcn_s_memtime();
    for (int k = 0; k < iters; k++) {
        state = state * 1664525u + 1013904223u + (v & 1u);
        const uint32_t line = __builtin_amdgcn_readfirstlane(state >> 16);            // 65 536 lines of 128 B = 8 MiB
        uint32_t* a = reinterpret_cast<uint32_t*>(p + (uint64_t)line * 128u) + lane;
        const uint64_t ta = __builtin_amdgcn_s_memtime();
        asm volatile("global_load_dword %0, %1, off\n s_waitcnt vmcnt(0)" : "=v"(v) : "v"(a) : "memory");     // plain, as the decoder's
        lat += __builtin_amdgcn_s_memtime() - ta;
        if (mix & 1) asm volatile("global_store_dword %0, %1, off" :: "v"(a), "v"(v + 1u) : "memory");
        uint32_t s2 = state;                                                          // (the side loads are waited for with the next line: in order)
#pragma unroll
        for (int j = 0; j < 3; j++) {
            s2 = s2 * 22695477u + 1u;
            uint32_t* b = reinterpret_cast<uint32_t*>(q + (uint64_t)(s2 >> 19) * 128u);   // 8 192 lines = 1 MiB
            if (mix & 8) { if (lane < 16) asm volatile("global_store_dword %0, %1, off" :: "v"(b + lane), "v"(s2) : "memory"); }
            else if (mix & 2) { if (lane == 0) asm volatile("global_store_dword %0, %1, off" :: "v"(b), "v"(s2) : "memory"); }
            if (mix & 4) { if (lane == 0) asm volatile("global_load_dword v200, %0, off" :: "v"(b + 16) : "memory", "v200"); }   // lands in a register nothing else uses
        }
        uint32_t d = v;
        for (int j = 0; j < delay; j++) asm volatile(".rept 20\n v_add_u32 %0, %0, 1\n .endr" : "+v"(d));   // 20 instructions a turn
        v ^= d & 2u;
    }
    const uint64_t t1 = __builtin_amdgcn_s_memtime();
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory", "v200");
    if (threadIdx.x == 0) { out[blockIdx.x * 4] = t1 - t0; out[blockIdx.x * 4 + 1] = lat; out[blockIdx.x * 4 + 2] = v + acc; }
}

int main(int argc, char** argv) {
    const uint64_t stride = 35ull << 20;
    const int nmax = 3052;
    uint8_t* d; if (hipMalloc(&d, stride * nmax) != hipSuccess) { printf("alloc failed\n"); return 1; }
    hipMemset(d, 0, stride * nmax);
    uint64_t* d_out; hipMalloc(&d_out, nmax * 32);
    const int iters = 3000;
    const int ns[] = {1, 256, 1024, 1526, 3052};
    const int mixes[] = {0, 1, 2, 4, 3, 7, 9, 13};
    const int delays[] = {0, 8, 16};      // x 20 instructions (+ the loop's own)
    printf("clocks per iteration / of which the timed load (mean over waves)\n");
    for (int delay : delays) for (int n : ns) {
        printf("delay %3d x 20 instr, %4d waves:", delay, n);
        for (int mix : mixes) {
            for (int rep = 0; rep < 2; rep++) {
                hipLaunchKernelGGL(k_probe, dim3(n), dim3(64), 0, 0, d, stride, iters, mix, delay, d_out);
                hipDeviceSynchronize();
            }
            std::vector<uint64_t> h(n * 4);
            hipMemcpy(h.data(), d_out, n * 32, hipMemcpyDeviceToHost);
            double it = 0, la = 0;
            for (int i = 0; i < n; i++) { it += (double)h[i * 4] / iters; la += (double)h[i * 4 + 1] / iters; }
            printf("  mix %2d: %5.0f /%5.0f", mix, it / n, la / n);
            fflush(stdout);
        }
        printf("\n");
    }
    return 0;
}

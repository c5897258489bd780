// Exactness and cost of the decoder's range / total division done in double precision (round 5, crgpu_rop5.h c5_vdiv):
//   q = trunc((range + 0.5) * r),  r = v_rcp_f64(total) after ONE Newton step
// against the u32 division the reference performs (cr-rangecoder.c:101-104: range /= sum). total < 2^20 (an order-1 row
// sums to at most 256 x 2 025), range any u32 (the coder keeps it >= 2^24).
// (range + 0.5) / total = q + (rem + 0.5) / total lies at least 0.5 / total from an integer, and the computed product is off
// by at most (range / total) x eps: exact as long as eps < 0.5 / range, i.e. 2^-33. The probe checks every total in
// [1, 2^20) against ranges chosen at the quotient boundaries (k x total - 1, k x total, k x total + 1 for random k, the
// extremes) and random ones, also WITHOUT the Newton step (how good is v_rcp_f64 alone?).
// build: hipcc --offload-arch=gfx950 -O3 tools/div_probe.hip -o tools/bin/div_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

template <int NEWTON>
__device__ __forceinline__ uint32_t div_f64(uint32_t range, uint32_t tot) {
    uint32_t q;
    double dd, dr, de, dn;
    if (NEWTON)
        asm volatile("v_cvt_f64_u32 %[dd], %[tot]\n v_rcp_f64 %[dr], %[dd]\n v_cvt_f64_u32 %[dn], %[rng]\n"
                     "v_fma_f64 %[de], -%[dd], %[dr], 1.0\n v_fma_f64 %[dr], %[de], %[dr], %[dr]\n"
                     "v_add_f64 %[dn], %[dn], 0.5\n v_mul_f64 %[dn], %[dn], %[dr]\n v_cvt_u32_f64 %[q], %[dn]"
                     : [q] "=&v"(q), [dd] "=&v"(dd), [dr] "=&v"(dr), [de] "=&v"(de), [dn] "=&v"(dn) : [tot] "v"(tot), [rng] "v"(range));
    else
        asm volatile("v_cvt_f64_u32 %[dd], %[tot]\n v_rcp_f64 %[dr], %[dd]\n v_cvt_f64_u32 %[dn], %[rng]\n"
                     "v_add_f64 %[dn], %[dn], 0.5\n s_nop 0\n v_mul_f64 %[dn], %[dn], %[dr]\n v_cvt_u32_f64 %[q], %[dn]"
                     : [q] "=&v"(q), [dd] "=&v"(dd), [dr] "=&v"(dr), [de] "=&v"(de), [dn] "=&v"(dn) : [tot] "v"(tot), [rng] "v"(range));
    return q;
}

template <int NEWTON>
__global__ void k_check(uint64_t* out, uint32_t tot_hi) {
    const uint32_t tot = blockIdx.x * blockDim.x + threadIdx.x + 1u;
    if (tot >= tot_hi) return;
    uint64_t bad = 0, n = 0;
    uint32_t s = tot * 2654435761u + 1u;
    auto test = [&](uint32_t range) { n++; if (div_f64<NEWTON>(range, tot) != range / tot) bad++; };
    test(0xffffffffu); test(0xfffffffeu); test(0x01000000u); test(0x00ffffffu); test(tot); test(tot - 1u); test(0u); test(1u);
    const uint32_t qmax = 0xffffffffu / tot;
    for (int i = 0; i < 96; i++) {
        s = s * 1664525u + 1013904223u;
        const uint32_t k = (uint32_t)(((uint64_t)s * qmax) >> 32) + 1u;      // 1 .. qmax
        const uint32_t b = k * tot;
        test(b); test(b - 1u); if (b != 0xffffffffu) test(b + 1u);
        s = s * 1664525u + 1013904223u;
        test(s); test(s | 0x01000000u);
    }
    test(qmax * tot); test(qmax * tot - 1u);
    atomicAdd((unsigned long long*)&out[0], (unsigned long long)bad);
    atomicAdd((unsigned long long*)&out[1], (unsigned long long)n);
}

// cost: a dependent chain of divisions, the round-4 u32 sequence (21 instructions) against the f64 one
__global__ void k_cost(uint64_t* out, int iters, int which) {
    uint32_t range = 0x9e3779b9u + threadIdx.x * 0u, tot = 31337u, acc = 0;
    const uint64_t c0 = __builtin_amdgcn_s_memtime();
    for (int k = 0; k < iters; k++) {
        uint32_t q;
        if (which == 0) {
            uint32_t dm, dneg, dq1, dr, dr1;
            asm volatile("v_cvt_f32_u32 %[dm], %[tot]\n v_rcp_iflag_f32 %[dm], %[dm]\n v_sub_u32 %[dneg], 0, %[tot]\n v_mul_f32 %[dm], 0x4f7ffffe, %[dm]\n"
                         "v_cvt_u32_f32 %[dm], %[dm]\n v_mul_lo_u32 %[dneg], %[dneg], %[dm]\n v_mul_hi_u32 %[dneg], %[dm], %[dneg]\n v_add_u32 %[dm], %[dm], %[dneg]\n"
                         "v_mul_hi_u32 %[q], %[rng], %[dm]\n v_mul_lo_u32 %[dr], %[q], %[tot]\n v_sub_u32 %[dr], %[rng], %[dr]\n v_cmp_ge_u32 vcc, %[dr], %[tot]\n"
                         "v_add_u32 %[dq1], 1, %[q]\n v_sub_u32 %[dr1], %[dr], %[tot]\n v_cndmask_b32 %[q], %[q], %[dq1], vcc\n v_cndmask_b32 %[dr], %[dr], %[dr1], vcc\n"
                         "v_add_u32 %[dq1], 1, %[q]\n v_cmp_ge_u32 vcc, %[dr], %[tot]\n s_nop 0\n s_nop 0\n v_cndmask_b32 %[q], %[q], %[dq1], vcc"
                         : [q] "=&v"(q), [dm] "=&v"(dm), [dneg] "=&v"(dneg), [dq1] "=&v"(dq1), [dr] "=&v"(dr), [dr1] "=&v"(dr1)
                         : [tot] "v"(tot), [rng] "v"(range) : "vcc");
        } else q = div_f64<1>(range, tot);
        acc += q;
        range = (range ^ q) | 0x01000000u;
        tot = (q & 0xffffu) + 2u;
    }
    const uint64_t c1 = __builtin_amdgcn_s_memtime();
    if (threadIdx.x == 0) { out[2 + which] = c1 - c0; out[4 + which] = acc; }
}

int main() {
    uint64_t* d; (void)hipMalloc(&d, 64);
    for (int newton = 1; newton >= 0; newton--) {
        (void)hipMemset(d, 0, 64);
        const uint32_t hi = 1u << 20;
        if (newton) hipLaunchKernelGGL(k_check<1>, dim3(hi / 256), dim3(256), 0, 0, d, hi);
        else hipLaunchKernelGGL(k_check<0>, dim3(hi / 256), dim3(256), 0, 0, d, hi);
        (void)hipDeviceSynchronize();
        uint64_t h[2]; (void)hipMemcpy(h, d, 16, hipMemcpyDeviceToHost);
        printf("%-28s totals 1 .. 2^20 - 1, %llu divisions checked against u32 division: %llu wrong\n",
               newton ? "rcp_f64 + one Newton step:" : "rcp_f64 alone:", (unsigned long long)h[1], (unsigned long long)h[0]);
    }
    const int iters = 100000;
    for (int w = 0; w < 2; w++) { hipLaunchKernelGGL(k_cost, dim3(1), dim3(64), 0, 0, d, iters, w); hipLaunchKernelGGL(k_cost, dim3(1), dim3(64), 0, 0, d, iters, w); }
    (void)hipDeviceSynchronize();
    uint64_t h[6]; (void)hipMemcpy(h, d, 48, hipMemcpyDeviceToHost);
    printf("one wave, dependent divisions (+ 4 instructions of loop): u32 sequence %.1f clocks each, f64 sequence %.1f clocks each\n",
           (double)h[2] / iters, (double)h[3] / iters);
    return 0;
}

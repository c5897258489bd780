// Issue-rate probe for a lone wave on gfx950: cycles per instruction for dependent / independent VALU and
// SALU chains, VALU<->SALU hops, DPP, readlane, taken branches, and the shader clock the chip holds.
// build: hipcc --offload-arch=gfx950 -O3 tools/issue_probe.hip -o /tmp/issue_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

#define REP(body) asm volatile(".rept 256\n" body "\n.endr" : "+v"(v), "+v"(u), "+s"(s), "+s"(t) :: "vcc", "scc", "v100", "v101", "v102", "v103", "s90", "s91")
#define PROBE(name, body) \
__global__ void name(uint64_t* out, int iters) { \
    uint32_t v = threadIdx.x, u = threadIdx.x * 3u, s = 1, t = 2; \
    uint64_t c0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime(); \
    for (int k = 0; k < iters; k++) { REP(body); } \
    uint64_t c1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime(); \
    if (threadIdx.x == 0) { out[0] = c1 - c0; out[1] = r1 - r0; out[2] = v + u + s + t; } \
}
PROBE(k_valu_dep,   "v_add_u32 %0, %0, %1")
PROBE(k_valu_ind,   "v_add_u32 %0, %1, %1\n v_add_u32 %1, 1, %1")
PROBE(k_salu_dep,   "s_add_u32 %2, %2, %3")
PROBE(k_salu_ind,   "s_add_u32 %2, %3, %3\n s_add_u32 %3, %3, 1")
PROBE(k_v2s_hop,    "v_readfirstlane_b32 %2, %0\n v_add_u32 %0, %2, %0")
PROBE(k_mix_ind,    "v_add_u32 %0, %0, %1\n s_add_u32 %2, %2, %3")
PROBE(k_dpp,        "v_add_u32_dpp %0, %0, %0 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1")
PROBE(k_mul_lo,     "v_mul_lo_u32 %0, %0, %1")
PROBE(k_smul,       "s_mul_i32 %2, %2, %3")
PROBE(k_cmp_sel,    "s_cmp_lt_u32 %2, %3\n s_cselect_b32 %2, %3, %2")
PROBE(k_vcmp_cnd,   "v_cmp_lt_u32 vcc, %0, %1\n v_cndmask_b32 %0, %0, %1, vcc")
PROBE(k_branch,     "s_cmp_eq_u32 %2, %2\n s_cbranch_scc1 1f\n s_nop 0\n1:")
PROBE(k_nop,        "s_nop 0")
PROBE(k_rcp,        "v_rcp_f32 %0, %0")
// round 5: what the decoder's division and search are made of
PROBE(k_mul_hi,     "v_mul_hi_u32 %0, %0, %1")
PROBE(k_mul_u24,    "v_mul_u32_u24 %0, %0, %1")
PROBE(k_mad_u24,    "v_mad_u32_u24 %0, %0, %1, %1")
PROBE(k_mad_u64,    "v_mad_u64_u32 v[100:101], vcc, %0, %1, v[100:101]\n v_mov_b32 %0, v100")
PROBE(k_cvt_fu,     "v_cvt_f32_u32 %0, %0")
PROBE(k_cvt_uf,     "v_cvt_u32_f32 %0, %0")
PROBE(k_mul_f32,    "v_mul_f32 %0, %0, %1")
PROBE(k_fma_f32,    "v_fma_f32 %0, %0, %1, %1")
PROBE(k_rcp_f64,    "v_rcp_f64 v[100:101], v[100:101]")
PROBE(k_fma_f64,    "v_fma_f64 v[100:101], v[100:101], v[102:103], v[102:103]")
PROBE(k_cvt_f64u,   "v_cvt_f64_u32 v[100:101], %0\n v_cvt_u32_f64 %0, v[100:101]")
PROBE(k_lshl64,     "v_lshlrev_b64 v[100:101], %1, v[100:101]")
PROBE(k_ffbh,       "v_ffbh_u32 %0, %0")
PROBE(k_rdlane_hop, "v_readlane_b32 %2, %0, 63\n v_add_u32 %0, %2, %0")
PROBE(k_rdlane_nop, "v_add_u32 %0, %1, %0\n s_nop 0\n v_readlane_b32 %2, %0, 63\n s_add_u32 %3, %2, %3")
PROBE(k_smul_hi,    "s_mul_hi_u32 %2, %2, %3")
PROBE(k_bcnt,       "v_cmp_ge_u32 vcc, %0, %1\n s_bcnt1_i32_b64 %2, vcc\n v_add_u32 %0, %2, %0")
PROBE(k_snop1,      "s_nop 1")
PROBE(k_snop3,      "s_nop 3")
PROBE(k_sdwa,       "v_add_u32_sdwa %0, %0, %1 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_1")
PROBE(k_ballot,     "v_cmp_ge_u32_e64 s[90:91], %0, %1\n s_bcnt1_i32_b64 %2, s[90:91]")

// vector-memory issue cost: N instructions back to back (same line, L2-resident), one wait at the end of each group of 8
#define VPROBE(name, body) \
__global__ void name(uint64_t* out, int iters, uint32_t* buf) { \
    uint32_t v = threadIdx.x * 4u, u = 0, w = 1, z = 2; \
    uint64_t c0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime(); \
    for (int k = 0; k < iters; k++) { \
        asm volatile(".rept 32\n" body "\n.endr\n s_waitcnt vmcnt(0)" : "+v"(u), "+v"(w), "+v"(z) : "v"(v), "s"(buf) : "memory"); \
    } \
    uint64_t c1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime(); \
    if (threadIdx.x == 0) { out[0] = c1 - c0; out[1] = r1 - r0; out[2] = u + w + z; } \
}
VPROBE(k_vload64,   "global_load_dword %0, %3, %4")
VPROBE(k_vstore64,  "global_store_dword %3, %1, %4")
VPROBE(k_vstore1,   "s_mov_b64 exec, 1\n global_store_dword %3, %1, %4\n s_mov_b64 exec, -1")
VPROBE(k_vstoreb,   "global_store_byte %3, %1, %4")
VPROBE(k_vmix,      "global_load_dword %0, %3, %4\n global_store_dword %3, %1, %4 offset:1024")
template <typename K> static void vrun(const char* name, K k, uint64_t* d_out, uint32_t* buf, int per_iter) {
    const int iters = 400;
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d_out, iters, buf);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d_out, iters, buf);
    (void)hipDeviceSynchronize();
    uint64_t h[3]; (void)hipMemcpy(h, d_out, 24, hipMemcpyDeviceToHost);
    double n = (double)iters * 32 * per_iter;
    printf("%-12s %6.2f cycles/instr incl. one drain per 32 (%.0f cycles per group)\n", name, h[0] / n, h[0] / (double)iters);
}

template <typename K> static void run(const char* name, K k, uint64_t* d_out, int per_iter) {
    const int iters = 200;
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d_out, iters);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d_out, iters);
    hipDeviceSynchronize();
    uint64_t h[3]; hipMemcpy(h, d_out, 24, hipMemcpyDeviceToHost);
    double n = (double)iters * 256 * per_iter;
    printf("%-12s %6.2f cycles/instr  %6.2f ns/instr  clock %.2f GHz\n", name, h[0] / n, h[1] * 10.0 / n, (double)h[0] / (h[1] * 10.0));
}
int main() {
    uint64_t* d_out; hipMalloc(&d_out, 64);
    run("valu_dep", k_valu_dep, d_out, 1);
    run("valu_ind", k_valu_ind, d_out, 2);
    run("salu_dep", k_salu_dep, d_out, 1);
    run("salu_ind", k_salu_ind, d_out, 2);
    run("v2s_hop", k_v2s_hop, d_out, 2);
    run("mix_ind", k_mix_ind, d_out, 2);
    run("dpp_dep", k_dpp, d_out, 1);
    run("mul_lo_dep", k_mul_lo, d_out, 1);
    run("s_mul_dep", k_smul, d_out, 1);
    run("s_cmp_csel", k_cmp_sel, d_out, 2);
    run("v_cmp_cnd", k_vcmp_cnd, d_out, 2);
    run("branch_tkn", k_branch, d_out, 2);
    run("s_nop", k_nop, d_out, 1);
    run("v_rcp_dep", k_rcp, d_out, 1);
    run("mul_hi_dep", k_mul_hi, d_out, 1);
    run("mul_u24_dep", k_mul_u24, d_out, 1);
    run("mad_u24_dep", k_mad_u24, d_out, 1);
    run("mad_u64+mov", k_mad_u64, d_out, 2);
    run("cvt_f32_u32", k_cvt_fu, d_out, 1);
    run("cvt_u32_f32", k_cvt_uf, d_out, 1);
    run("mul_f32_dep", k_mul_f32, d_out, 1);
    run("fma_f32_dep", k_fma_f32, d_out, 1);
    run("rcp_f64_dep", k_rcp_f64, d_out, 1);
    run("fma_f64_dep", k_fma_f64, d_out, 1);
    run("cvt f64 pair", k_cvt_f64u, d_out, 2);
    run("lshl_b64_dep", k_lshl64, d_out, 1);
    run("ffbh_dep", k_ffbh, d_out, 1);
    run("rdlane_hop", k_rdlane_hop, d_out, 2);
    run("add,nop,rdl,sadd", k_rdlane_nop, d_out, 4);
    run("s_mul_hi_dep", k_smul_hi, d_out, 1);
    run("cmp,bcnt,add", k_bcnt, d_out, 3);
    run("s_nop 1", k_snop1, d_out, 1);
    run("s_nop 3", k_snop3, d_out, 1);
    run("sdwa_add_dep", k_sdwa, d_out, 1);
    run("cmp_e64,bcnt", k_ballot, d_out, 2);
    // round 5: the same dependent chains with 8 and 16 waves on the CU (two and four per SIMD): which multiplication holds the vector pipe?
    for (int threads : {64, 512, 1024}) {
        struct { const char* n; void (*k)(uint64_t*, int); int per; } ks[] = {
            {"valu_dep", k_valu_dep, 1}, {"mul_lo_dep", k_mul_lo, 1}, {"mul_hi_dep", k_mul_hi, 1}, {"mul_u24_dep", k_mul_u24, 1}, {"mad_u64+mov", k_mad_u64, 2}, {"fma_f64_dep", k_fma_f64, 1}, {"s_mul_dep", k_smul, 1}};
        for (auto& e : ks) {
            hipLaunchKernelGGL(e.k, dim3(1), dim3(threads), 0, 0, d_out, 200);
            hipLaunchKernelGGL(e.k, dim3(1), dim3(threads), 0, 0, d_out, 200);
            (void)hipDeviceSynchronize();
            uint64_t h[3]; (void)hipMemcpy(h, d_out, 24, hipMemcpyDeviceToHost);
            printf("%-12s %4d threads on one CU: %6.2f cycles/instr (wave 0)\n", e.n, threads, h[0] / (200.0 * 256 * e.per));
        }
    }
    uint32_t* buf; (void)hipMalloc(&buf, 1 << 20); (void)hipMemset(buf, 0, 1 << 20);
    vrun("vload x64", k_vload64, d_out, buf, 1);
    vrun("vstore x64", k_vstore64, d_out, buf, 1);
    vrun("vstore x1", k_vstore1, d_out, buf, 3);
    vrun("vstore byte", k_vstoreb, d_out, buf, 1);
    vrun("load+store", k_vmix, d_out, buf, 2);
    return 0;
}

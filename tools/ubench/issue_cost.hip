// Micro-benchmark (diagnostic, not product): cycles per wave-instruction on one SIMD for the instruction
// classes the march and frequency kernels are made of, at 1..4 waves per SIMD.
//   hipcc --offload-arch=gfx950 -O3 -o issue_cost issue_cost.hip && ./issue_cost
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <string>

#define REP16(x) x x x x x x x x x x x x x x x x
#define BODY(NAME, ASM)                                                                              \
    __global__ void __launch_bounds__(1024) k_##NAME(unsigned long long *out, int iters)             \
    {                                                                                                \
        float a = threadIdx.x * 1e-3f + 1.0f, b = 1.0001f, c = 0.5f, d = a + 1.0f;                   \
        double x = a, y = 1.0001, z = 0.5, w = x + 1.0;                                              \
        int i0 = (threadIdx.x & 63) * 16, i1 = 0xfff;                                                                \
        unsigned long long t0 = __builtin_readcyclecounter();                                        \
        for (int i = 0; i < iters; i++) {                                                            \
            REP16(asm volatile(ASM ASM ASM ASM : "+v"(a), "+v"(d), "+v"(x), "+v"(w), "+v"(i0) : "v"(b), "v"(c), "v"(y), "v"(z), "v"(i1) : "vcc", "scc", "s10", "s11", "s12", "s13", "v20","v21","v22","v23","v24","v25","v26","v27");) \
        }                                                                                            \
        unsigned long long t1 = __builtin_readcyclecounter();                                        \
        if ((threadIdx.x & 63) == 0)                                                                 \
            out[blockIdx.x * 16 + (threadIdx.x >> 6)] = t1 - t0;                                     \
        if (a == 123.456f || x == 1.5 || i0 == -77 || d == 3.25f || w == 9.75)                       \
            out[0] = 0;                                                                              \
    }

// operands: %0 a(f32) %1 d(f32) %2 x(f64) %3 w(f64) %4 i0 ; %5 b %6 c %7 y %8 z %9 i1
BODY(fma32_dep, "v_fma_f32 %0, %0, %5, %6\n")
BODY(fma32_ind, "v_fma_f32 %0, %0, %5, %6\n v_fma_f32 %1, %1, %5, %6\n")
BODY(mul32, "v_mul_f32 %0, %0, %5\n v_mul_f32 %1, %1, %5\n")
BODY(cndmask, "v_cndmask_b32 %0, %0, %5, vcc\n v_cndmask_b32 %1, %1, %6, vcc\n")
BODY(cmp32, "v_cmp_lt_f32 vcc, %0, %5\n v_cmp_lt_f32 vcc, %1, %6\n")
BODY(mov, "v_mov_b32 %0, %5\n v_mov_b32 %1, %6\n")
BODY(rcp32, "v_rcp_f32 %0, %0\n v_rcp_f32 %1, %1\n")
BODY(rsq32, "v_rsq_f32 %0, %0\n v_rsq_f32 %1, %1\n")
BODY(fma64, "v_fma_f64 %2, %2, %7, %8\n v_fma_f64 %3, %3, %7, %8\n")
BODY(mul64, "v_mul_f64 %2, %2, %7\n v_mul_f64 %3, %3, %7\n")
BODY(add64, "v_add_f64 %2, %2, %7\n v_add_f64 %3, %3, %7\n")
BODY(cvt64_32, "v_cvt_f64_f32 %2, %0\n v_cvt_f64_f32 %3, %1\n")
BODY(cvt32_64, "v_cvt_f32_f64 %0, %2\n v_cvt_f32_f64 %1, %3\n")
BODY(cmp64, "v_cmp_lt_f64 vcc, %2, %7\n v_cmp_lt_f64 vcc, %3, %8\n")
BODY(mullo, "v_mul_lo_u32 %4, %4, %9\n v_mul_lo_u32 %4, %4, %9\n")
BODY(addu, "v_add_u32 %4, %4, %9\n v_add_u32 %4, %4, %9\n")
BODY(divscale, "v_div_scale_f32 %0, vcc, %0, %5, %0\n v_div_scale_f32 %1, vcc, %1, %5, %1\n")
BODY(divfix, "v_div_fixup_f32 %0, %0, %5, %6\n v_div_fixup_f32 %1, %1, %5, %6\n")
BODY(divfmas, "v_div_fmas_f32 %0, %0, %5, %6\n v_div_fmas_f32 %1, %1, %5, %6\n")
BODY(salu, "s_and_b64 vcc, vcc, exec\n s_or_b64 vcc, vcc, exec\n")
BODY(mix_fs, "v_fma_f32 %0, %0, %5, %6\n s_and_b64 vcc, vcc, exec\n")
BODY(min3, "v_min3_f32 %0, %0, %5, %6\n v_min3_f32 %1, %1, %5, %6\n")
BODY(bfi, "v_bfi_b32 %4, %9, %4, %9\n v_bfi_b32 %4, %9, %4, %9\n")
BODY(pkfma, "v_pk_fma_f32 %2, %2, %7, %8\n v_pk_fma_f32 %3, %3, %7, %8\n")

BODY(cnd_e64, "v_cndmask_b32_e64 %0, %0, %5, s[10:11]\n v_cndmask_b32_e64 %1, %1, %6, s[10:11]\n")
BODY(cnd_ind, "v_cndmask_b32 %0, %5, %6, vcc\n v_cndmask_b32 %1, %6, %5, vcc\n")
BODY(cnd_ind64, "v_cndmask_b32_e64 %0, %5, %6, s[10:11]\n v_cndmask_b32_e64 %1, %6, %5, s[10:11]\n")
BODY(snop, "s_nop 0\n s_nop 0\n")
BODY(swait, "s_waitcnt lgkmcnt(0)\n s_waitcnt vmcnt(0)\n")
BODY(saveexec, "v_cmp_lt_f32 vcc, %0, %5\n s_and_saveexec_b64 s[10:11], vcc\n s_or_b64 exec, exec, s[10:11]\n v_fma_f32 %0, %0, %5, %6\n")
BODY(br_nt, "s_cbranch_execz 1f\n v_fma_f32 %0, %0, %5, %6\n 1:\n s_cbranch_execz 2f\n v_fma_f32 %0, %0, %5, %6\n 2:\n s_cbranch_execz 3f\n v_fma_f32 %0, %0, %5, %6\n 3:\n s_cbranch_execz 4f\n v_fma_f32 %0, %0, %5, %6\n 4:\n")
BODY(br_tk, "s_branch 1f\n v_fma_f32 %0, %0, %5, %6\n 1:\n v_fma_f32 %1, %1, %5, %6\n")
BODY(cmp_cnd, "v_cmp_lt_f32 vcc, %0, %5\n v_cndmask_b32 %1, %5, %6, vcc\n")
BODY(cmp_e64, "v_cmp_lt_f32_e64 s[10:11], %0, %5\n v_cmp_lt_f32_e64 s[12:13], %1, %6\n")
BODY(and_b32, "v_and_b32 %4, %4, %9\n v_and_b32 %4, %4, %9\n")
BODY(fma_sgpr, "v_fma_f32 %0, %0, s10, %6\n v_fma_f32 %1, %1, s11, %6\n")
BODY(absfma, "v_fma_f32 %0, |%0|, %5, %6\n v_fma_f32 %1, -%1, %5, %6\n")
BODY(dsread, "ds_read_b128 v[20:23], %4\n ds_read_b128 v[24:27], %4 offset:16\n")
BODY(dsread32, "ds_read_b32 v20, %4\n ds_read_b32 v24, %4 offset:16\n")
BODY(dsread64, "ds_read_b64 v[20:21], %4\n ds_read_b64 v[24:25], %4 offset:16\n")
BODY(dsread96, "ds_read_b96 v[20:22], %4\n ds_read_b96 v[24:26], %4 offset:16\n")
BODY(min32, "v_min_f32 %0, %0, %5\n v_min_f32 %1, %1, %6\n")
BODY(cmp_nop_cnd, "v_cmp_lt_f32 vcc, %0, %5\n s_nop 1\n v_cndmask_b32 %1, %5, %6, vcc\n")
BODY(readlane, "v_readlane_b32 s10, %0, 3\n v_readlane_b32 s11, %1, 5\n")
BODY(fma_inl, "v_fma_f32 %0, %0, %5, 1.0\n v_fma_f32 %1, %1, %5, 0.5\n")
BODY(fmaak, "v_fmaak_f32 %0, %0, %5, 0x3f800000\n v_fmaak_f32 %1, %1, %5, 0x3f000000\n")
BODY(fmamk, "v_fmamk_f32 %0, %0, 0x3e2aaaab, %5\n v_fmamk_f32 %1, %1, 0x3e2aaaab, %6\n")
BODY(mul_s, "v_mul_f32 %0, s10, %0\n v_mul_f32 %1, s11, %1\n")
BODY(add_lit, "v_add_f32 %0, 0x40400000, %0\n v_add_f32 %1, 0x40400000, %1\n")
BODY(add_inl, "v_add_f32 %0, 1.0, %0\n v_add_f32 %1, 2.0, %1\n")
BODY(fma_abs_s, "v_fma_f32 %0, -|%0|, %5, s10\n v_fma_f32 %1, -|%1|, %5, s11\n")
BODY(add64_inl, "v_add_f64 %2, %2, -1.0\n v_add_f64 %3, %3, -1.0\n")
BODY(fmac64, "v_fmac_f64 %2, %7, %8\n v_fmac_f64 %3, %7, %8\n")
BODY(fmac32, "v_fmac_f32 %0, %5, %6\n v_fmac_f32 %1, %5, %6\n")
BODY(fmac32_lit, "v_fmac_f32 %0, 0xbb317218, %5\n v_fmac_f32 %1, 0x2d02e308, %6\n")
BODY(mix_64_32, "v_fma_f64 %2, %2, %7, %8\n v_fma_f32 %0, %0, %5, %6\n")
BODY(mix_64_32x2, "v_fma_f64 %2, %2, %7, %8\n v_fma_f32 %0, %0, %5, %6\n v_fma_f32 %1, %1, %5, %6\n")
BODY(mix_64_pk, "v_fma_f64 %2, %2, %7, %8\n v_pk_fma_f32 %3, %3, %7, %8\n")
BODY(pkfma_dep, "v_pk_fma_f32 %2, %2, %7, %8\n")
BODY(fma64_dep, "v_fma_f64 %2, %2, %7, %8\n")
BODY(mix_64_ds, "v_fma_f64 %2, %2, %7, %8\n ds_read_b64 v[20:21], %4\n")
BODY(mix_32_ds, "v_fma_f32 %0, %0, %5, %6\n ds_read_b64 v[20:21], %4\n")
BODY(sdwa, "v_lshlrev_b32_sdwa %4, %9, %4 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_0\n v_lshlrev_b32_sdwa %4, %9, %4 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_0\n")
BODY(lshladd, "v_lshl_add_u32 %4, %4, 12, %9\n v_lshl_add_u32 %4, %4, 12, %9\n")
BODY(mov_dpp, "v_mov_b32_dpp %0, %1 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n v_mov_b32_dpp %1, %0 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n")

struct K { const char *name; void (*fn)(unsigned long long *, int); int per_rep; };
int main()
{
    std::vector<K> ks = {
#define E(n, c) { #n, k_##n, c }
        E(fma32_dep, 1), E(fma32_ind, 2), E(mul32, 2), E(cndmask, 2), E(cmp32, 2), E(mov, 2), E(min3, 2), E(bfi, 2), E(addu, 2),
        E(mullo, 2), E(rcp32, 2), E(rsq32, 2), E(divscale, 2), E(divfix, 2), E(divfmas, 2), E(fma64, 2), E(mul64, 2), E(add64, 2),
        E(cvt64_32, 2), E(cvt32_64, 2), E(cmp64, 2), E(pkfma, 2), E(salu, 2), E(mix_fs, 2), E(cnd_e64,2), E(cnd_ind,2), E(cnd_ind64,2), E(snop,2), E(swait,2), E(saveexec,4), E(br_nt,8), E(br_tk,2), E(cmp_cnd,2), E(cmp_e64,2), E(and_b32,2), E(fma_sgpr,2), E(absfma,2), E(dsread,2), E(dsread32,2), E(dsread64,2), E(dsread96,2), E(min32,2), E(cmp_nop_cnd,3), E(readlane,2), E(fma_inl,2), E(fmaak,2), E(fmamk,2), E(mul_s,2), E(add_lit,2), E(add_inl,2), E(fma_abs_s,2), E(add64_inl,2), E(fmac64,2), E(fmac32,2), E(fmac32_lit,2), E(mix_64_32,2), E(mix_64_32x2,3), E(mix_64_pk,2), E(pkfma_dep,1), E(fma64_dep,1), E(mix_64_ds,2), E(mix_32_ds,2), E(sdwa,2), E(lshladd,2), E(mov_dpp,2) };
    unsigned long long *d;
    hipMalloc(&d, 4096 * 16 * 8);
    const int iters = 500;
    printf("%-12s %8s %8s %8s %8s   (cycles per wave-instruction per SIMD; waves/SIMD = 1,2,3,4)\n", "instr", "w1", "w2", "w3", "w4");
    for (auto &k : ks) {
        printf("%-12s", k.name);
        for (int wps = 1; wps <= 4; wps++) {
            const int threads = 256 * wps; // 4 SIMDs x wps waves
            hipMemset(d, 0, 4096 * 16 * 8);
            hipLaunchKernelGGL(k.fn, dim3(256), dim3(threads), 0, 0, d, iters);
            hipLaunchKernelGGL(k.fn, dim3(256), dim3(threads), 0, 0, d, iters);
            hipDeviceSynchronize();
            std::vector<unsigned long long> h(256 * 16);
            hipMemcpy(h.data(), d, h.size() * 8, hipMemcpyDeviceToHost);
            double sum = 0; int n = 0;
            for (int b = 0; b < 256; b++) for (int w = 0; w < 4 * wps; w++) { sum += (double) h[b * 16 + w]; n++; }
            const double cyc_wave = sum / n;                       // cycles one wave needed
            const double instr_wave = (double) iters * 16 * k.per_rep * 4;
            // SIMD throughput: wps waves ran concurrently on the SIMD
            printf(" %8.2f", cyc_wave / (instr_wave * wps));
        }
        printf("\n"); fflush(stdout);
    }
    return 0;
}

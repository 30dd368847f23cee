// VALU issue-rate microbenchmark for gfx950: cycles per wave64 instruction per SIMD for a
// few instruction kinds, with W waves per SIMD. Diagnostic only (roofline).
//   hipcc -O3 --offload-arch=gfx950 tools/ubench/valu_rate.hip -o tools/ubench/valu_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

#define REP 64      // instructions per accumulator group and loop iteration
#define ITERS 2000

template <int KIND>
__global__ __launch_bounds__(256) void k(float *out, float a, float b)
{
    float x0 = threadIdx.x, x1 = x0 + 1, x2 = x0 + 2, x3 = x0 + 3, x4 = x0 + 4, x5 = x0 + 5, x6 = x0 + 6, x7 = x0 + 7;
    typedef float v2 __attribute__((ext_vector_type(2)));
    v2 p0{x0, x1}, p1{x2, x3}, p2{x4, x5}, p3{x6, x7}, pa{a, a}, pb{b, b};
    double d0 = x0, d1 = x1, d2 = x2, d3 = x3, da = a, db = b;
    for (int it = 0; it < ITERS; ++it) {
#pragma unroll
        for (int r = 0; r < REP / 8; ++r) {
            if (KIND == 0) {   // v_fma_f32
                asm volatile("v_fma_f32 %0, %0, %8, %9\n v_fma_f32 %1, %1, %8, %9\n v_fma_f32 %2, %2, %8, %9\n v_fma_f32 %3, %3, %8, %9\n"
                             "v_fma_f32 %4, %4, %8, %9\n v_fma_f32 %5, %5, %8, %9\n v_fma_f32 %6, %6, %8, %9\n v_fma_f32 %7, %7, %8, %9\n"
                             : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7) : "v"(a), "v"(b));
            } else if (KIND == 1) {   // v_mul_f32
                asm volatile("v_mul_f32 %0, %0, %8\n v_mul_f32 %1, %1, %8\n v_mul_f32 %2, %2, %8\n v_mul_f32 %3, %3, %8\n"
                             "v_mul_f32 %4, %4, %8\n v_mul_f32 %5, %5, %8\n v_mul_f32 %6, %6, %8\n v_mul_f32 %7, %7, %8\n"
                             : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7) : "v"(a));
            } else if (KIND == 2) {   // v_add_f32
                asm volatile("v_add_f32 %0, %0, %8\n v_add_f32 %1, %1, %8\n v_add_f32 %2, %2, %8\n v_add_f32 %3, %3, %8\n"
                             "v_add_f32 %4, %4, %8\n v_add_f32 %5, %5, %8\n v_add_f32 %6, %6, %8\n v_add_f32 %7, %7, %8\n"
                             : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7) : "v"(a));
            } else if (KIND == 3) {   // v_pk_fma_f32 (4 of them = 8 flops-lanes)
                asm volatile("v_pk_fma_f32 %0, %0, %4, %5\n v_pk_fma_f32 %1, %1, %4, %5\n v_pk_fma_f32 %2, %2, %4, %5\n v_pk_fma_f32 %3, %3, %4, %5\n"
                             "v_pk_fma_f32 %0, %0, %4, %5\n v_pk_fma_f32 %1, %1, %4, %5\n v_pk_fma_f32 %2, %2, %4, %5\n v_pk_fma_f32 %3, %3, %4, %5\n"
                             : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3) : "v"(pa), "v"(pb));
            } else if (KIND == 4) {   // v_pk_mul_f32
                asm volatile("v_pk_mul_f32 %0, %0, %4\n v_pk_mul_f32 %1, %1, %4\n v_pk_mul_f32 %2, %2, %4\n v_pk_mul_f32 %3, %3, %4\n"
                             "v_pk_mul_f32 %0, %0, %4\n v_pk_mul_f32 %1, %1, %4\n v_pk_mul_f32 %2, %2, %4\n v_pk_mul_f32 %3, %3, %4\n"
                             : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3) : "v"(pa));
            } else if (KIND == 5) {   // v_pk_add_f32
                asm volatile("v_pk_add_f32 %0, %0, %4\n v_pk_add_f32 %1, %1, %4\n v_pk_add_f32 %2, %2, %4\n v_pk_add_f32 %3, %3, %4\n"
                             "v_pk_add_f32 %0, %0, %4\n v_pk_add_f32 %1, %1, %4\n v_pk_add_f32 %2, %2, %4\n v_pk_add_f32 %3, %3, %4\n"
                             : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3) : "v"(pa));
            } else if (KIND == 6) {   // v_fma_f64
                asm volatile("v_fma_f64 %0, %0, %4, %5\n v_fma_f64 %1, %1, %4, %5\n v_fma_f64 %2, %2, %4, %5\n v_fma_f64 %3, %3, %4, %5\n"
                             "v_fma_f64 %0, %0, %4, %5\n v_fma_f64 %1, %1, %4, %5\n v_fma_f64 %2, %2, %4, %5\n v_fma_f64 %3, %3, %4, %5\n"
                             : "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3) : "v"(da), "v"(db));
            } else if (KIND == 7) {   // v_rcp_f32 (transcendental)
                asm volatile("v_rcp_f32 %0, %0\n v_rcp_f32 %1, %1\n v_rcp_f32 %2, %2\n v_rcp_f32 %3, %3\n"
                             "v_rcp_f32 %4, %4\n v_rcp_f32 %5, %5\n v_rcp_f32 %6, %6\n v_rcp_f32 %7, %7\n"
                             : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7));
            } else if (KIND == 8) {   // v_cndmask_b32 (vcc)
                asm volatile("v_cndmask_b32 %0, %0, %8, vcc\n v_cndmask_b32 %1, %1, %8, vcc\n v_cndmask_b32 %2, %2, %8, vcc\n v_cndmask_b32 %3, %3, %8, vcc\n"
                             "v_cndmask_b32 %4, %4, %8, vcc\n v_cndmask_b32 %5, %5, %8, vcc\n v_cndmask_b32 %6, %6, %8, vcc\n v_cndmask_b32 %7, %7, %8, vcc\n"
                             : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7) : "v"(a) : "vcc");
            } else if (KIND == 9) {   // v_mul_f64
                asm volatile("v_mul_f64 %0, %0, %4\n v_mul_f64 %1, %1, %4\n v_mul_f64 %2, %2, %4\n v_mul_f64 %3, %3, %4\n"
                             "v_mul_f64 %0, %0, %4\n v_mul_f64 %1, %1, %4\n v_mul_f64 %2, %2, %4\n v_mul_f64 %3, %3, %4\n"
                             : "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3) : "v"(da));
            } else if (KIND == 10) {  // v_sqrt_f32
                asm volatile("v_sqrt_f32 %0, %0\n v_sqrt_f32 %1, %1\n v_sqrt_f32 %2, %2\n v_sqrt_f32 %3, %3\n"
                             "v_sqrt_f32 %4, %4\n v_sqrt_f32 %5, %5\n v_sqrt_f32 %6, %6\n v_sqrt_f32 %7, %7\n"
                             : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7));
            } else if (KIND == 11) {  // v_max_f32 with DPP row_shr:1
                asm volatile("v_max_f32_dpp %0, %0, %0 row_shr:1 row_mask:0xf bank_mask:0xf\n v_max_f32_dpp %1, %1, %1 row_shr:1 row_mask:0xf bank_mask:0xf\n"
                             "v_max_f32_dpp %2, %2, %2 row_shr:1 row_mask:0xf bank_mask:0xf\n v_max_f32_dpp %3, %3, %3 row_shr:1 row_mask:0xf bank_mask:0xf\n"
                             "v_max_f32_dpp %4, %4, %4 row_shr:1 row_mask:0xf bank_mask:0xf\n v_max_f32_dpp %5, %5, %5 row_shr:1 row_mask:0xf bank_mask:0xf\n"
                             "v_max_f32_dpp %6, %6, %6 row_shr:1 row_mask:0xf bank_mask:0xf\n v_max_f32_dpp %7, %7, %7 row_shr:1 row_mask:0xf bank_mask:0xf\n"
                             : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7));
            }
        }
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = x0 + x1 + x2 + x3 + x4 + x5 + x6 + x7 + p0.x + p0.y + p1.x + p1.y + p2.x + p2.y +
                                                 p3.x + p3.y + (float)(d0 + d1 + d2 + d3);
}


// ---- second batch: one instruction kind per kernel through a macro (8 independent destinations) ----
#define K8(NAME, TEXT)                                                                                              \
    __global__ __launch_bounds__(256) void NAME(float *out, float a, float b, unsigned long long *clk)              \
    {                                                                                                                 \
        float x0 = threadIdx.x, x1 = x0 + 1, x2 = x0 + 2, x3 = x0 + 3, x4 = x0 + 4, x5 = x0 + 5, x6 = x0 + 6, x7 = x0 + 7; \
        float sa = __builtin_amdgcn_readfirstlane(a), sb = __builtin_amdgcn_readfirstlane(b);                         \
        const unsigned long long t0 = __builtin_amdgcn_s_memtime();                                                  \
        for (int it = 0; it < ITERS; ++it) {                                                                          \
            _Pragma("unroll") for (int r = 0; r < REP / 8; ++r) {                                                     \
                asm volatile(TEXT : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7)   \
                             : "v"(a), "v"(b), "s"(sa), "s"(sb) : "vcc", "s40", "s41", "v200", "v201", "v202", "v203", "v204", "v205", "v206", "v207");                                   \
            }                                                                                                         \
        }                                                                                                             \
        const unsigned long long t1 = __builtin_amdgcn_s_memtime();                                                  \
        if (blockIdx.x == 0 && threadIdx.x == 0) *clk = t1 - t0;                                                      \
        out[blockIdx.x * blockDim.x + threadIdx.x] = x0 + x1 + x2 + x3 + x4 + x5 + x6 + x7;                            \
    }
#define R8(I) I(0) I(1) I(2) I(3) I(4) I(5) I(6) I(7)
K8(k_mul, "v_mul_f32 %0, %0, %8\n v_mul_f32 %1, %1, %8\n v_mul_f32 %2, %2, %8\n v_mul_f32 %3, %3, %8\n v_mul_f32 %4, %4, %8\n v_mul_f32 %5, %5, %8\n v_mul_f32 %6, %6, %8\n v_mul_f32 %7, %7, %8\n")
K8(k_mul_s, "v_mul_f32 %0, %10, %0\n v_mul_f32 %1, %10, %1\n v_mul_f32 %2, %10, %2\n v_mul_f32 %3, %10, %3\n v_mul_f32 %4, %10, %4\n v_mul_f32 %5, %10, %5\n v_mul_f32 %6, %10, %6\n v_mul_f32 %7, %10, %7\n")
K8(k_sub, "v_sub_f32 %0, %0, %8\n v_sub_f32 %1, %1, %8\n v_sub_f32 %2, %2, %8\n v_sub_f32 %3, %3, %8\n v_sub_f32 %4, %4, %8\n v_sub_f32 %5, %5, %8\n v_sub_f32 %6, %6, %8\n v_sub_f32 %7, %7, %8\n")
K8(k_fmac, "v_fmac_f32 %0, %8, %9\n v_fmac_f32 %1, %8, %9\n v_fmac_f32 %2, %8, %9\n v_fmac_f32 %3, %8, %9\n v_fmac_f32 %4, %8, %9\n v_fmac_f32 %5, %8, %9\n v_fmac_f32 %6, %8, %9\n v_fmac_f32 %7, %8, %9\n")
K8(k_fma_s, "v_fma_f32 %0, %0, %10, %11\n v_fma_f32 %1, %1, %10, %11\n v_fma_f32 %2, %2, %10, %11\n v_fma_f32 %3, %3, %10, %11\n v_fma_f32 %4, %4, %10, %11\n v_fma_f32 %5, %5, %10, %11\n v_fma_f32 %6, %6, %10, %11\n v_fma_f32 %7, %7, %10, %11\n")
K8(k_fma_c, "v_fma_f32 %0, %0, %8, 1.0\n v_fma_f32 %1, %1, %8, 1.0\n v_fma_f32 %2, %2, %8, 1.0\n v_fma_f32 %3, %3, %8, 1.0\n v_fma_f32 %4, %4, %8, 1.0\n v_fma_f32 %5, %5, %8, 1.0\n v_fma_f32 %6, %6, %8, 1.0\n v_fma_f32 %7, %7, %8, 1.0\n")
K8(k_mov, "v_mov_b32 %0, %8\n v_mov_b32 %1, %8\n v_mov_b32 %2, %8\n v_mov_b32 %3, %8\n v_mov_b32 %4, %8\n v_mov_b32 %5, %8\n v_mov_b32 %6, %8\n v_mov_b32 %7, %8\n")
K8(k_and, "v_and_b32 %0, %0, %8\n v_and_b32 %1, %1, %8\n v_and_b32 %2, %2, %8\n v_and_b32 %3, %3, %8\n v_and_b32 %4, %4, %8\n v_and_b32 %5, %5, %8\n v_and_b32 %6, %6, %8\n v_and_b32 %7, %7, %8\n")
K8(k_cmp, "v_cmp_gt_f32 vcc, %0, %8\n v_cmp_gt_f32 vcc, %1, %8\n v_cmp_gt_f32 vcc, %2, %8\n v_cmp_gt_f32 vcc, %3, %8\n v_cmp_gt_f32 vcc, %4, %8\n v_cmp_gt_f32 vcc, %5, %8\n v_cmp_gt_f32 vcc, %6, %8\n v_cmp_gt_f32 vcc, %7, %8\n")
K8(k_cmp_e64, "v_cmp_gt_f32 s[40:41], %0, %8\n v_cmp_gt_f32 s[40:41], %1, %8\n v_cmp_gt_f32 s[40:41], %2, %8\n v_cmp_gt_f32 s[40:41], %3, %8\n v_cmp_gt_f32 s[40:41], %4, %8\n v_cmp_gt_f32 s[40:41], %5, %8\n v_cmp_gt_f32 s[40:41], %6, %8\n v_cmp_gt_f32 s[40:41], %7, %8\n")
K8(k_cnd, "v_cndmask_b32 %0, %0, %8, vcc\n v_cndmask_b32 %1, %1, %8, vcc\n v_cndmask_b32 %2, %2, %8, vcc\n v_cndmask_b32 %3, %3, %8, vcc\n v_cndmask_b32 %4, %4, %8, vcc\n v_cndmask_b32 %5, %5, %8, vcc\n v_cndmask_b32 %6, %6, %8, vcc\n v_cndmask_b32 %7, %7, %8, vcc\n")
K8(k_cnd_e64, "v_cndmask_b32 %0, %0, %8, s[40:41]\n v_cndmask_b32 %1, %1, %8, s[40:41]\n v_cndmask_b32 %2, %2, %8, s[40:41]\n v_cndmask_b32 %3, %3, %8, s[40:41]\n v_cndmask_b32 %4, %4, %8, s[40:41]\n v_cndmask_b32 %5, %5, %8, s[40:41]\n v_cndmask_b32 %6, %6, %8, s[40:41]\n v_cndmask_b32 %7, %7, %8, s[40:41]\n")
K8(k_cmp_cnd, "v_cmp_gt_f32 vcc, %0, %8\n v_cndmask_b32 %0, %0, %9, vcc\n v_cmp_gt_f32 vcc, %1, %8\n v_cndmask_b32 %1, %1, %9, vcc\n v_cmp_gt_f32 vcc, %2, %8\n v_cndmask_b32 %2, %2, %9, vcc\n v_cmp_gt_f32 vcc, %3, %8\n v_cndmask_b32 %3, %3, %9, vcc\n")
K8(k_divscale, "v_div_scale_f32 %0, vcc, %0, %8, %0\n v_div_scale_f32 %1, vcc, %1, %8, %1\n v_div_scale_f32 %2, vcc, %2, %8, %2\n v_div_scale_f32 %3, vcc, %3, %8, %3\n v_div_scale_f32 %4, vcc, %4, %8, %4\n v_div_scale_f32 %5, vcc, %5, %8, %5\n v_div_scale_f32 %6, vcc, %6, %8, %6\n v_div_scale_f32 %7, vcc, %7, %8, %7\n")
K8(k_divfmas, "v_div_fmas_f32 %0, %0, %8, %9\n v_div_fmas_f32 %1, %1, %8, %9\n v_div_fmas_f32 %2, %2, %8, %9\n v_div_fmas_f32 %3, %3, %8, %9\n v_div_fmas_f32 %4, %4, %8, %9\n v_div_fmas_f32 %5, %5, %8, %9\n v_div_fmas_f32 %6, %6, %8, %9\n v_div_fmas_f32 %7, %7, %8, %9\n")
K8(k_divfixup, "v_div_fixup_f32 %0, %0, %8, %9\n v_div_fixup_f32 %1, %1, %8, %9\n v_div_fixup_f32 %2, %2, %8, %9\n v_div_fixup_f32 %3, %3, %8, %9\n v_div_fixup_f32 %4, %4, %8, %9\n v_div_fixup_f32 %5, %5, %8, %9\n v_div_fixup_f32 %6, %6, %8, %9\n v_div_fixup_f32 %7, %7, %8, %9\n")
K8(k_readlane, "v_readlane_b32 s40, %0, 3\n v_readlane_b32 s40, %1, 3\n v_readlane_b32 s40, %2, 3\n v_readlane_b32 s40, %3, 3\n v_readlane_b32 s40, %4, 3\n v_readlane_b32 s40, %5, 3\n v_readlane_b32 s40, %6, 3\n v_readlane_b32 s40, %7, 3\n")
K8(k_salu, "s_add_u32 s40, s40, 1\n s_add_u32 s40, s40, 1\n s_add_u32 s40, s40, 1\n s_add_u32 s40, s40, 1\n s_add_u32 s40, s40, 1\n s_add_u32 s40, s40, 1\n s_add_u32 s40, s40, 1\n s_add_u32 s40, s40, 1\n")
K8(k_mix_vs, "v_mul_f32 %0, %0, %8\n s_add_u32 s40, s40, 1\n v_mul_f32 %1, %1, %8\n s_add_u32 s41, s41, 1\n v_mul_f32 %2, %2, %8\n s_add_u32 s40, s40, 1\n v_mul_f32 %3, %3, %8\n s_add_u32 s41, s41, 1\n")
K8(k_rsq, "v_rsq_f32 %0, %0\n v_rsq_f32 %1, %1\n v_rsq_f32 %2, %2\n v_rsq_f32 %3, %3\n v_rsq_f32 %4, %4\n v_rsq_f32 %5, %5\n v_rsq_f32 %6, %6\n v_rsq_f32 %7, %7\n")
K8(k_cvt, "v_cvt_f64_f32 v[200:201], %0\n v_cvt_f64_f32 v[202:203], %1\n v_cvt_f64_f32 v[204:205], %2\n v_cvt_f64_f32 v[206:207], %3\n v_cvt_f64_f32 v[200:201], %4\n v_cvt_f64_f32 v[202:203], %5\n v_cvt_f64_f32 v[204:205], %6\n v_cvt_f64_f32 v[206:207], %7\n")

typedef void (*kfn)(float *, float, float, unsigned long long *);
static void run2(const char *name, kfn f, int per_block, int waves_per_simd, float *out, unsigned long long *clk)
{
    hipDeviceProp_t p; (void)hipGetDeviceProperties(&p, 0);
    const int blocks = p.multiProcessorCount * waves_per_simd;
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    hipLaunchKernelGGL(f, dim3(blocks), dim3(256), 0, 0, out, 1.0001f, 0.5f, clk);
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL(f, dim3(blocks), dim3(256), 0, 0, out, 1.0001f, 0.5f, clk);
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    unsigned long long c; (void)hipMemcpy(&c, clk, 8, hipMemcpyDeviceToHost);
    const double insts = (double)waves_per_simd * ITERS * REP * per_block / 8.0;
    printf("%-16s w/SIMD %d: %.3f ms  memtime ticks/wave %llu (%.1f MHz)  %.2f ns per inst per SIMD  %.2f ticks per inst per SIMD\n",
           name, waves_per_simd, ms, c, c / (ms * 1e3), ms * 1e6 / insts, (double)c * waves_per_simd / insts / waves_per_simd * 1.0);
}

template <int KIND>
static void run(const char *name, int waves_per_simd, float *out, double ghz)
{
    hipDeviceProp_t p; hipGetDeviceProperties(&p, 0);
    const int cus = p.multiProcessorCount;
    const int blocks = cus * waves_per_simd;   // 256 threads = 4 waves = one per SIMD
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k<KIND>, dim3(blocks), dim3(256), 0, 0, out, 1.0001f, 0.5f);
    hipEventRecord(e0);
    hipLaunchKernelGGL(k<KIND>, dim3(blocks), dim3(256), 0, 0, out, 1.0001f, 0.5f);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double insts_per_simd = (double)waves_per_simd * ITERS * REP;
    printf("%-14s waves/SIMD %d: %.3f ms, %.2f ns per instruction per SIMD = %.2f cycles at %.2f GHz\n", name, waves_per_simd, ms,
           ms * 1e6 / insts_per_simd, ms * 1e6 / insts_per_simd * ghz, ghz);
}

int main(int argc, char **argv)
{
    const double ghz = argc > 1 ? atof(argv[1]) : 2.4;
    float *out; hipMalloc(&out, 256 * 8 * 256 * sizeof(float) * 4);
    for (int w : {1, 2, 4, 8}) {
        run<0>("v_fma_f32", w, out, ghz); run<1>("v_mul_f32", w, out, ghz); run<2>("v_add_f32", w, out, ghz);
        run<3>("v_pk_fma_f32", w, out, ghz); run<4>("v_pk_mul_f32", w, out, ghz); run<5>("v_pk_add_f32", w, out, ghz);
        run<6>("v_fma_f64", w, out, ghz); run<9>("v_mul_f64", w, out, ghz); run<7>("v_rcp_f32", w, out, ghz);
        run<10>("v_sqrt_f32", w, out, ghz); run<8>("v_cndmask_b32", w, out, ghz); run<11>("v_max_f32_dpp", w, out, ghz);
    }
    unsigned long long *clk; (void)hipMalloc(&clk, 8);
    for (int w : {4, 8}) {
#define RUN2(K, N) run2(#K, K, N, w, out, clk)
        RUN2(k_mul, 8); RUN2(k_mul_s, 8); RUN2(k_sub, 8); RUN2(k_fmac, 8); RUN2(k_fma_s, 8); RUN2(k_fma_c, 8); RUN2(k_mov, 8); RUN2(k_and, 8);
        RUN2(k_cmp, 8); RUN2(k_cmp_e64, 8); RUN2(k_cnd, 8); RUN2(k_cnd_e64, 8); RUN2(k_cmp_cnd, 8); RUN2(k_divscale, 8); RUN2(k_divfmas, 8);
        RUN2(k_divfixup, 8); RUN2(k_readlane, 8); RUN2(k_salu, 8); RUN2(k_mix_vs, 8); RUN2(k_rsq, 8); RUN2(k_cvt, 8);
    }
    return 0;
}

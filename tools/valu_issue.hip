// Issue cost of VALU instruction classes on gfx950, measured in shader cycles
// (s_memtime) so that clock changes under load do not matter.  W waves per SIMD
// run the same independent-instruction stream; the figure printed is
// SIMD cycles per wave-instruction = loop cycles / (instructions per wave * waves per SIMD).
// Build: hipcc -O3 --offload-arch=gfx950 tools/valu_issue.hip -o tools/valu_issue
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>

#define REP8(x) x x x x x x x x
#define KERNEL(name, body)                                                                   \
    __global__ __launch_bounds__(256) void name(float *out, unsigned long long *cyc, int iters, float a) \
    {                                                                                        \
        float v0 = threadIdx.x, v1 = v0 + 1, v2 = v0 + 2, v3 = v0 + 3, v4 = v0 + 4, v5 = v0 + 5, v6 = v0 + 6, v7 = v0 + 7; \
        unsigned long long t0 = __builtin_amdgcn_s_memtime();                                \
        for (int i = 0; i < iters; i++) {                                                    \
            asm volatile(REP8(body) : "+v"(v0), "+v"(v1), "+v"(v2), "+v"(v3), "+v"(v4), "+v"(v5), "+v"(v6), "+v"(v7) : "s"(a) : "vcc", "s40", "s41"); \
        }                                                                                    \
        unsigned long long t1 = __builtin_amdgcn_s_memtime();                                \
        out[blockIdx.x * blockDim.x + threadIdx.x] = v0 + v1 + v2 + v3 + v4 + v5 + v6 + v7;  \
        if ((threadIdx.x & 63) == 0) cyc[(blockIdx.x * blockDim.x + threadIdx.x) >> 6] = t1 - t0; \
    }

typedef float f2v __attribute__((ext_vector_type(2)));
#define KERNEL64(name, body)                                                                 \
    __global__ __launch_bounds__(256) void name(float *out, unsigned long long *cyc, int iters, float a) \
    {                                                                                        \
        f2v v0 = {(float)threadIdx.x, 1.f}, v1 = v0 + 1.f, v2 = v0 + 2.f, v3 = v0 + 3.f, v4 = v0 + 4.f, v5 = v0 + 5.f, v6 = v0 + 6.f, v7 = v0 + 7.f; \
        unsigned long long t0 = __builtin_amdgcn_s_memtime();                                \
        for (int i = 0; i < iters; i++) {                                                    \
            asm volatile(REP8(body) : "+v"(v0), "+v"(v1), "+v"(v2), "+v"(v3), "+v"(v4), "+v"(v5), "+v"(v6), "+v"(v7) : "s"(a) : "vcc", "s40", "s41"); \
        }                                                                                    \
        unsigned long long t1 = __builtin_amdgcn_s_memtime();                                \
        f2v r = v0 + v1 + v2 + v3 + v4 + v5 + v6 + v7;                                       \
        out[blockIdx.x * blockDim.x + threadIdx.x] = r.x + r.y;                              \
        if ((threadIdx.x & 63) == 0) cyc[(blockIdx.x * blockDim.x + threadIdx.x) >> 6] = t1 - t0; \
    }

// each body = 4 instructions on independent registers; %8 is an SGPR
KERNEL(k_fma_v,  "v_fma_f32 %0, %0, %1, %0\n v_fma_f32 %2, %2, %3, %2\n v_fma_f32 %4, %4, %5, %4\n v_fma_f32 %6, %6, %7, %6\n")
KERNEL(k_fma_s,  "v_fma_f32 %0, %0, %8, %0\n v_fma_f32 %2, %2, %8, %2\n v_fma_f32 %4, %4, %8, %4\n v_fma_f32 %6, %6, %8, %6\n")
KERNEL(k_mul_v,  "v_mul_f32 %0, %1, %0\n v_mul_f32 %2, %3, %2\n v_mul_f32 %4, %5, %4\n v_mul_f32 %6, %7, %6\n")
KERNEL(k_mul_s,  "v_mul_f32 %0, %8, %0\n v_mul_f32 %2, %8, %2\n v_mul_f32 %4, %8, %4\n v_mul_f32 %6, %8, %6\n")
KERNEL(k_fmac,   "v_fmac_f32 %0, %1, %2\n v_fmac_f32 %2, %3, %4\n v_fmac_f32 %4, %5, %6\n v_fmac_f32 %6, %7, %0\n")
KERNEL64(k_pkfma_v,"v_pk_fma_f32 %0, %0, %1, %0\n v_pk_fma_f32 %2, %2, %3, %2\n v_pk_fma_f32 %4, %4, %5, %4\n v_pk_fma_f32 %6, %6, %7, %6\n")
KERNEL64(k_pkmul_v,"v_pk_mul_f32 %0, %0, %1\n v_pk_mul_f32 %2, %2, %3\n v_pk_mul_f32 %4, %4, %5\n v_pk_mul_f32 %6, %6, %7\n")
KERNEL64(k_pkadd_v,"v_pk_add_f32 %0, %0, %1\n v_pk_add_f32 %2, %2, %3\n v_pk_add_f32 %4, %4, %5\n v_pk_add_f32 %6, %6, %7\n")
KERNEL(k_cmp,    "v_cmp_gt_f32 vcc, %8, %0\n v_cmp_gt_f32 vcc, %8, %1\n v_cmp_gt_f32 vcc, %8, %2\n v_cmp_gt_f32 vcc, %8, %3\n")
KERNEL(k_cnd,    "v_cndmask_b32 %0, %0, %1, vcc\n v_cndmask_b32 %2, %2, %3, vcc\n v_cndmask_b32 %4, %4, %5, vcc\n v_cndmask_b32 %6, %6, %7, vcc\n")
KERNEL(k_cnd_s,  "v_cndmask_b32 %0, %0, %1, s[40:41]\n v_cndmask_b32 %2, %2, %3, s[40:41]\n v_cndmask_b32 %4, %4, %5, s[40:41]\n v_cndmask_b32 %6, %6, %7, s[40:41]\n")
KERNEL(k_cnd_ind,"v_cndmask_b32 %0, %1, %2, vcc\n v_cndmask_b32 %3, %4, %5, vcc\n v_cndmask_b32 %6, %7, %1, vcc\n v_cndmask_b32 %2, %4, %5, vcc\n")
KERNEL(k_cmpcnd_v, "v_cmp_gt_f32 vcc, %0, %1\n v_cndmask_b32 %2, %2, %3, vcc\n v_cmp_gt_f32 vcc, %4, %5\n v_cndmask_b32 %6, %6, %7, vcc\n")
KERNEL(k_cmpcnd_s, "v_cmp_gt_f32 s[40:41], %0, %1\n v_cndmask_b32 %2, %2, %3, s[40:41]\n v_cmp_gt_f32 s[40:41], %4, %5\n v_cndmask_b32 %6, %6, %7, s[40:41]\n")
KERNEL(k_cmpaddc,  "v_cmp_gt_f32 vcc, %0, %1\n v_addc_co_u32 %2, vcc, %2, %2, vcc\n v_cmp_gt_f32 vcc, %4, %5\n v_addc_co_u32 %6, vcc, %6, %6, vcc\n")
KERNEL(k_min3,     "v_min3_f32 %0, %0, %1, %2\n v_min3_f32 %3, %3, %4, %5\n v_min3_f32 %6, %6, %7, %1\n v_min3_f32 %2, %2, %4, %5\n")
KERNEL(k_cmp_e64,"v_cmp_gt_f32 s[40:41], %0, %1\n v_cmp_gt_f32 s[40:41], %2, %3\n v_cmp_gt_f32 s[40:41], %4, %5\n v_cmp_gt_f32 s[40:41], %6, %7\n")
KERNEL(k_cmp_vv, "v_cmp_gt_f32 vcc, %0, %1\n v_cmp_gt_f32 vcc, %2, %3\n v_cmp_gt_f32 vcc, %4, %5\n v_cmp_gt_f32 vcc, %6, %7\n")
KERNEL(k_movs,   "v_mov_b32 %0, %8\n v_mov_b32 %2, %8\n v_mov_b32 %4, %8\n v_mov_b32 %6, %8\n")
KERNEL(k_sub_s,  "v_sub_f32 %0, %8, %0\n v_sub_f32 %2, %8, %2\n v_sub_f32 %4, %8, %4\n v_sub_f32 %6, %8, %6\n")
KERNEL(k_fma_c,  "v_fma_f32 %0, %0, 2.0, %0\n v_fma_f32 %2, %2, 2.0, %2\n v_fma_f32 %4, %4, 2.0, %4\n v_fma_f32 %6, %6, 2.0, %6\n")
KERNEL(k_fmaak,  "v_fmaak_f32 %0, %0, %1, 0x3f000001\n v_fmaak_f32 %2, %2, %3, 0x3f000001\n v_fmaak_f32 %4, %4, %5, 0x3f000001\n v_fmaak_f32 %6, %6, %7, 0x3f000001\n")
KERNEL(k_mov,    "v_mov_b32 %0, %1\n v_mov_b32 %2, %3\n v_mov_b32 %4, %5\n v_mov_b32 %6, %7\n")
KERNEL(k_rcp,    "v_rcp_f32 %0, %0\n v_rcp_f32 %1, %1\n v_rcp_f32 %2, %2\n v_rcp_f32 %3, %3\n")
KERNEL(k_dscale, "v_div_scale_f32 %0, vcc, %0, %1, %0\n v_div_scale_f32 %2, vcc, %2, %3, %2\n v_div_scale_f32 %4, vcc, %4, %5, %4\n v_div_scale_f32 %6, vcc, %6, %7, %6\n")
KERNEL(k_dfmas,  "v_div_fmas_f32 %0, %0, %1, %2\n v_div_fmas_f32 %2, %2, %3, %4\n v_div_fmas_f32 %4, %4, %5, %6\n v_div_fmas_f32 %6, %6, %7, %0\n")
KERNEL(k_rdlane, "v_readlane_b32 s40, %0, 3\n v_readlane_b32 s40, %1, 3\n v_readlane_b32 s40, %2, 3\n v_readlane_b32 s40, %3, 3\n")
KERNEL(k_dep,    "v_fma_f32 %0, %0, %1, %0\n v_fma_f32 %0, %0, %1, %0\n v_fma_f32 %0, %0, %1, %0\n v_fma_f32 %0, %0, %1, %0\n")
KERNEL(k_mulo,   "v_mul_lo_u32 %0, %0, %1\n v_mul_lo_u32 %2, %2, %3\n v_mul_lo_u32 %4, %4, %5\n v_mul_lo_u32 %6, %6, %7\n")
KERNEL64(k_pkfma_s,"v_pk_fma_f32 %0, %0, s[40:41], %0 op_sel_hi:[1,0,1]\n v_pk_fma_f32 %2, %2, s[40:41], %2 op_sel_hi:[1,0,1]\n v_pk_fma_f32 %4, %4, s[40:41], %4 op_sel_hi:[1,0,1]\n v_pk_fma_f32 %6, %6, s[40:41], %6 op_sel_hi:[1,0,1]\n")
KERNEL64(k_f64fma, "v_fma_f64 %0, %0, %1, %0\n v_fma_f64 %2, %2, %3, %2\n v_fma_f64 %4, %4, %5, %4\n v_fma_f64 %6, %6, %7, %6\n")

template <class Kn>
void run(const char *name, Kn kern, float *out, unsigned long long *cyc, int cus, int waves_per_simd)
{
    const int iters = 4000, blocks = cus * waves_per_simd;   /* 256 threads = 4 waves = one per SIMD */
    kern<<<blocks, 256>>>(out, cyc, iters, 1.0000001f);
    (void)hipDeviceSynchronize();
    kern<<<blocks, 256>>>(out, cyc, iters, 1.0000001f);
    (void)hipDeviceSynchronize();
    std::vector<unsigned long long> h(blocks * 4);
    (void)hipMemcpy(h.data(), cyc, h.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost);
    std::sort(h.begin(), h.end());
    double med = (double)h[h.size() / 2];
    printf("%-12s waves/SIMD=%d: %.2f SIMD cycles per wave-instruction\n", name, waves_per_simd, med / (32.0 * iters * waves_per_simd));
}

int main()
{
    hipDeviceProp_t p;
    (void)hipGetDeviceProperties(&p, 0);
    const int cus = p.multiProcessorCount;
    float *out;
    unsigned long long *cyc;
    (void)hipMalloc(&out, (size_t)cus * 8 * 256 * sizeof(float) * 2);
    (void)hipMalloc(&cyc, (size_t)cus * 8 * 4 * sizeof(unsigned long long));
    for (int w : {1, 2, 4, 8}) {
        run("fma vgpr", k_fma_v, out, cyc, cus, w);
        run("fma sgpr", k_fma_s, out, cyc, cus, w);
        run("pk_fma", k_pkfma_v, out, cyc, cus, w);
    }
    const int w = 4;
    run("mul vgpr", k_mul_v, out, cyc, cus, w);
    run("mul sgpr", k_mul_s, out, cyc, cus, w);
    run("fmac", k_fmac, out, cyc, cus, w);
    run("pk_mul", k_pkmul_v, out, cyc, cus, w);
    run("pk_add", k_pkadd_v, out, cyc, cus, w);
    run("cmp", k_cmp, out, cyc, cus, w);
    run("cndmask", k_cnd, out, cyc, cus, w);
    run("cndmask sgpr", k_cnd_s, out, cyc, cus, w);
    run("cndmask indep", k_cnd_ind, out, cyc, cus, w);
    run("cmp+cndmask via vcc (per instr)", k_cmpcnd_v, out, cyc, cus, w);
    run("cmp+cndmask via sgpr pair (per instr)", k_cmpcnd_s, out, cyc, cus, w);
    run("cmp+addc via vcc (per instr)", k_cmpaddc, out, cyc, cus, w);
    run("min3", k_min3, out, cyc, cus, w);
    run("cmp e64 sgpr", k_cmp_e64, out, cyc, cus, w);
    run("cmp vcc v,v", k_cmp_vv, out, cyc, cus, w);
    run("mov from sgpr", k_movs, out, cyc, cus, w);
    run("sub sgpr", k_sub_s, out, cyc, cus, w);
    run("fma inline c", k_fma_c, out, cyc, cus, w);
    run("fmaak literal", k_fmaak, out, cyc, cus, w);
    run("pk_fma sgpr", k_pkfma_s, out, cyc, cus, w);
    run("mov", k_mov, out, cyc, cus, w);
    run("rcp", k_rcp, out, cyc, cus, w);
    run("div_scale", k_dscale, out, cyc, cus, w);
    run("div_fmas", k_dfmas, out, cyc, cus, w);
    run("readlane", k_rdlane, out, cyc, cus, w);
    run("fma dep", k_dep, out, cyc, cus, w);
    run("mul_lo_u32", k_mulo, out, cyc, cus, w);
    run("fma_f64", k_f64fma, out, cyc, cus, w);
    return 0;
}

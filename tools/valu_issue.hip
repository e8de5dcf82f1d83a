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

// explicit registers: which VGPR banks (register number mod 4) the three sources of an fma come from
#define KERNEL_X(name, body)                                                                 \
    __global__ __launch_bounds__(256) void name(float *out, unsigned long long *cyc, int iters, float a) \
    {                                                                                        \
        float r = threadIdx.x;                                                               \
        asm volatile("v_mov_b32 v20, %0\n v_mov_b32 v21, %0\n v_mov_b32 v22, %0\n v_mov_b32 v23, %0\n v_mov_b32 v24, %0\n v_mov_b32 v25, %0\n" \
                     "v_mov_b32 v26, %0\n v_mov_b32 v27, %0\n v_mov_b32 v28, %0\n v_mov_b32 v29, %0\n v_mov_b32 v30, %0\n v_mov_b32 v31, %0\n" \
                     "v_mov_b32 v32, %0\n v_mov_b32 v33, %0\n v_mov_b32 v34, %0\n v_mov_b32 v35, %0\n v_mov_b32 v36, %0\n v_mov_b32 v37, %0\n v_mov_b32 v38, %0\n v_mov_b32 v39, %0\n" \
                     :: "v"(r) : "v20","v21","v22","v23","v24","v25","v26","v27","v28","v29","v30","v31","v32","v33","v34","v35","v36","v37","v38","v39"); \
        unsigned long long t0 = __builtin_amdgcn_s_memtime();                                \
        for (int i = 0; i < iters; i++) {                                                    \
            asm volatile(REP8(body) ::: "v20","v21","v22","v23","v24","v25","v26","v27","v28","v29","v30","v31","v32","v33","v34","v35","v36","v37","v38","v39"); \
        }                                                                                    \
        unsigned long long t1 = __builtin_amdgcn_s_memtime();                                \
        asm volatile("v_add_f32 %0, v20, v21\n v_add_f32 %0, %0, v22\n v_add_f32 %0, %0, v23\n" : "=v"(r) :: "v20","v21","v22","v23"); \
        out[blockIdx.x * blockDim.x + threadIdx.x] = r;                                      \
        if ((threadIdx.x & 63) == 0) cyc[(blockIdx.x * blockDim.x + threadIdx.x) >> 6] = t1 - t0; \
    }
KERNEL_X(k_x_banks123, "v_fma_f32 v20, v25, v26, v27\n v_fma_f32 v21, v29, v30, v31\n v_fma_f32 v22, v33, v34, v35\n v_fma_f32 v23, v37, v38, v39\n")
KERNEL_X(k_x_two_same, "v_fma_f32 v20, v25, v29, v27\n v_fma_f32 v21, v30, v34, v31\n v_fma_f32 v22, v33, v37, v35\n v_fma_f32 v23, v26, v38, v39\n")
KERNEL_X(k_x_all_same, "v_fma_f32 v20, v24, v28, v32\n v_fma_f32 v21, v25, v29, v33\n v_fma_f32 v22, v26, v30, v34\n v_fma_f32 v23, v27, v31, v35\n")
KERNEL_X(k_x_fmac,     "v_fmac_f32 v20, v25, v26\n v_fmac_f32 v21, v29, v30\n v_fmac_f32 v22, v33, v34\n v_fmac_f32 v23, v37, v38\n")
KERNEL_X(k_x_fmac_same,"v_fmac_f32 v20, v24, v28\n v_fmac_f32 v21, v25, v29\n v_fmac_f32 v22, v26, v30\n v_fmac_f32 v23, v27, v31\n")
// four fma with scalar instructions between them: are the scalar ones issued beside the vector ones or instead of them?
KERNEL_X(k_x_fma_salu, "v_fma_f32 v20, v25, v26, v27\n s_add_u32 s40, s40, 1\n v_fma_f32 v21, v29, v30, v31\n s_add_u32 s41, s41, 1\n v_fma_f32 v22, v33, v34, v35\n s_add_u32 s40, s40, 1\n v_fma_f32 v23, v37, v38, v39\n s_add_u32 s41, s41, 1\n")
KERNEL_X(k_x_fma_2salu, "v_fma_f32 v20, v25, v26, v27\n s_add_u32 s40, s40, 1\n s_add_u32 s41, s41, 1\n v_fma_f32 v21, v29, v30, v31\n s_add_u32 s40, s40, 1\n s_add_u32 s41, s41, 1\n v_fma_f32 v22, v33, v34, v35\n s_add_u32 s40, s40, 1\n s_add_u32 s41, s41, 1\n v_fma_f32 v23, v37, v38, v39\n s_add_u32 s40, s40, 1\n s_add_u32 s41, s41, 1\n")
KERNEL_X(k_x_fma_nop, "v_fma_f32 v20, v25, v26, v27\n s_nop 0\n v_fma_f32 v21, v29, v30, v31\n s_nop 0\n v_fma_f32 v22, v33, v34, v35\n s_nop 0\n v_fma_f32 v23, v37, v38, v39\n s_nop 0\n")
KERNEL_X(k_x_fma_wait, "v_fma_f32 v20, v25, v26, v27\n s_waitcnt lgkmcnt(0)\n v_fma_f32 v21, v29, v30, v31\n s_waitcnt lgkmcnt(0)\n v_fma_f32 v22, v33, v34, v35\n s_waitcnt lgkmcnt(0)\n v_fma_f32 v23, v37, v38, v39\n s_waitcnt lgkmcnt(0)\n")
KERNEL_X(k_x_fma_br, "v_fma_f32 v20, v25, v26, v27\n v_fma_f32 v21, v29, v30, v31\n s_cbranch_scc1 1f\n1:\n v_fma_f32 v22, v33, v34, v35\n v_fma_f32 v23, v37, v38, v39\n s_cbranch_vccz 2f\n2:\n")
KERNEL_X(k_x_fma_lds, "v_mov_b32 v24, 0\n v_fma_f32 v21, v29, v30, v31\n v_fma_f32 v22, v33, v34, v35\n v_fma_f32 v23, v37, v38, v39\n v_fma_f32 v20, v25, v26, v27\n v_fma_f32 v21, v29, v30, v31\n v_fma_f32 v22, v33, v34, v35\n ds_read_b128 v[36:39], v24\n s_waitcnt lgkmcnt(0)\n")

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
    printf("%-32s waves/SIMD=%d: %.2f SIMD cycles per wave-instruction\n", name, waves_per_simd, med / (32.0 * iters * waves_per_simd));
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
    for (int w : {1, 2, 4, 6, 8}) {
        run("fma 3 src, banks 1 2 3", k_x_banks123, out, cyc, cus, w);
        run("fma 3 src, two in one bank", k_x_two_same, out, cyc, cus, w);
        run("fma 3 src, all in one bank", k_x_all_same, out, cyc, cus, w);
        run("fmac 2 src, distinct banks", k_x_fmac, out, cyc, cus, w);
        run("fmac 2 src, dst+src one bank", k_x_fmac_same, out, cyc, cus, w);
    }
    for (int w : {1, 4, 6, 8}) {   /* figures are per fma: 4 per body (7 in the last) */
        run("fma (+1 s_add each), per pair", k_x_fma_salu, out, cyc, cus, w);
        run("fma (+2 s_add each), per triple", k_x_fma_2salu, out, cyc, cus, w);
        run("fma (+s_nop each), per pair", k_x_fma_nop, out, cyc, cus, w);
        run("fma (+s_waitcnt each), per pair", k_x_fma_wait, out, cyc, cus, w);
        run("2 fma + untaken branch, per 3", k_x_fma_br, out, cyc, cus, w);
        run("7 valu + ds_read_b128 + wait, per 4", k_x_fma_lds, out, cyc, cus, w);
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

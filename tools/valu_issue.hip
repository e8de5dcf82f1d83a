// Issue cost of individual VALU instruction classes on one SIMD, 8 waves per SIMD,
// independent registers (inline asm so the instruction stream is exactly what is written).
// Build: hipcc -O3 --offload-arch=gfx950 tools/valu_issue.hip -o tools/valu_issue
#include <hip/hip_runtime.h>
#include <cstdio>

#define REP8(x) x x x x x x x x
#define KERNEL(name, body)                                                             \
    __global__ __launch_bounds__(256) void name(float *out, int iters, float a)        \
    {                                                                                  \
        float v0 = threadIdx.x, v1 = v0 + 1, v2 = v0 + 2, v3 = v0 + 3;                 \
        for (int i = 0; i < iters; i++) {                                              \
            asm volatile(REP8(body) : "+v"(v0), "+v"(v1), "+v"(v2), "+v"(v3) : "s"(a) : "vcc", "s40", "s41"); \
        }                                                                              \
        out[blockIdx.x * blockDim.x + threadIdx.x] = v0 + v1 + v2 + v3;                \
    }

// each body = 4 instructions on 4 independent registers
KERNEL(k_fma,   "v_fma_f32 %0, %0, %4, %0\n v_fma_f32 %1, %1, %4, %1\n v_fma_f32 %2, %2, %4, %2\n v_fma_f32 %3, %3, %4, %3\n")
KERNEL(k_mul,   "v_mul_f32 %0, %4, %0\n v_mul_f32 %1, %4, %1\n v_mul_f32 %2, %4, %2\n v_mul_f32 %3, %4, %3\n")
KERNEL(k_fmac,  "v_fmac_f32 %0, %4, %1\n v_fmac_f32 %1, %4, %2\n v_fmac_f32 %2, %4, %3\n v_fmac_f32 %3, %4, %0\n")
KERNEL(k_cmp,   "v_cmp_gt_f32 vcc, %4, %0\n v_cmp_gt_f32 vcc, %4, %1\n v_cmp_gt_f32 vcc, %4, %2\n v_cmp_gt_f32 vcc, %4, %3\n")
KERNEL(k_cmp_s, "v_cmp_gt_f32 s[40:41], %4, %0\n v_cmp_gt_f32 s[40:41], %4, %1\n v_cmp_gt_f32 s[40:41], %4, %2\n v_cmp_gt_f32 s[40:41], %4, %3\n")
KERNEL(k_cnd,   "v_cndmask_b32 %0, %0, %1, vcc\n v_cndmask_b32 %1, %1, %2, vcc\n v_cndmask_b32 %2, %2, %3, vcc\n v_cndmask_b32 %3, %3, %0, vcc\n")
KERNEL(k_cmpcnd,"v_cmp_gt_f32 vcc, %4, %0\n s_nop 1\n v_cndmask_b32 %1, %1, %2, vcc\n v_cmp_gt_f32 vcc, %4, %2\n s_nop 1\n v_cndmask_b32 %3, %3, %0, vcc\n")
KERNEL(k_rcp,   "v_rcp_f32 %0, %0\n v_rcp_f32 %1, %1\n v_rcp_f32 %2, %2\n v_rcp_f32 %3, %3\n")
KERNEL(k_sqrt,  "v_sqrt_f32 %0, %0\n v_sqrt_f32 %1, %1\n v_sqrt_f32 %2, %2\n v_sqrt_f32 %3, %3\n")
KERNEL(k_dscale,"v_div_scale_f32 %0, vcc, %0, %1, %0\n v_div_scale_f32 %1, vcc, %1, %2, %1\n v_div_scale_f32 %2, vcc, %2, %3, %2\n v_div_scale_f32 %3, vcc, %3, %0, %3\n")
KERNEL(k_dfixup,"v_div_fixup_f32 %0, %0, %1, %2\n v_div_fixup_f32 %1, %1, %2, %3\n v_div_fixup_f32 %2, %2, %3, %0\n v_div_fixup_f32 %3, %3, %0, %1\n")
KERNEL(k_dfmas, "v_div_fmas_f32 %0, %0, %1, %2\n v_div_fmas_f32 %1, %1, %2, %3\n v_div_fmas_f32 %2, %2, %3, %0\n v_div_fmas_f32 %3, %3, %0, %1\n")
KERNEL(k_min3,  "v_min3_f32 %0, %0, %1, %2\n v_min3_f32 %1, %1, %2, %3\n v_min3_f32 %2, %2, %3, %0\n v_min3_f32 %3, %3, %0, %1\n")
KERNEL(k_rdlane,"v_readlane_b32 s40, %0, 3\n v_readlane_b32 s40, %1, 3\n v_readlane_b32 s40, %2, 3\n v_readlane_b32 s40, %3, 3\n")
KERNEL(k_dep,   "v_fma_f32 %0, %0, %4, %0\n v_fma_f32 %0, %0, %4, %0\n v_fma_f32 %0, %0, %4, %0\n v_fma_f32 %0, %0, %4, %0\n")
KERNEL(k_saveex,"v_cmp_gt_f32 vcc, %4, %0\n s_and_saveexec_b64 s[40:41], vcc\n v_fma_f32 %1, %1, %4, %1\n s_or_b64 exec, exec, s[40:41]\n")
KERNEL(k_salu,  "v_fma_f32 %0, %0, %4, %0\n s_mov_b32 s40, s41\n v_fma_f32 %1, %1, %4, %1\n s_mov_b32 s41, s40\n")

template <class K>
void run(const char *name, K kern, float *out, int blocks, int instr_per_iter)
{
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    const int iters = 20000;
    float best = 1e30f;
    for (int rep = 0; rep < 3; rep++) {
        (void)hipEventRecord(e0);
        kern<<<blocks, 256>>>(out, iters, 1.0000001f);
        (void)hipEventRecord(e1);
        (void)hipEventSynchronize(e1);
        float ms;
        (void)hipEventElapsedTime(&ms, e0, e1);
        best = ms < best ? ms : best;
    }
    // waves per SIMD = 8; cycles per wave-instruction per SIMD at 2.4 GHz
    double wave_instr_per_simd = 8.0 * iters * instr_per_iter;
    printf("%-10s %8.3f ms  %.2f cycles per wave-instruction per SIMD (at 2.4 GHz)\n", name, best, best * 1e-3 * 2.4e9 / wave_instr_per_simd);
}

int main()
{
    hipDeviceProp_t p;
    (void)hipGetDeviceProperties(&p, 0);
    const int blocks = p.multiProcessorCount * 8;
    float *out;
    (void)hipMalloc(&out, (size_t)blocks * 256 * sizeof(float));
    run("fma", k_fma, out, blocks, 32);
    run("mul", k_mul, out, blocks, 32);
    run("fmac", k_fmac, out, blocks, 32);
    run("cmp vcc", k_cmp, out, blocks, 32);
    run("cmp sgpr", k_cmp_s, out, blocks, 32);
    run("cndmask", k_cnd, out, blocks, 32);
    run("cmp+cnd", k_cmpcnd, out, blocks, 32);   /* 4 VALU + 2 s_nop per body */
    run("rcp", k_rcp, out, blocks, 32);
    run("sqrt", k_sqrt, out, blocks, 32);
    run("div_scale", k_dscale, out, blocks, 32);
    run("div_fixup", k_dfixup, out, blocks, 32);
    run("div_fmas", k_dfmas, out, blocks, 32);
    run("min3", k_min3, out, blocks, 32);
    run("readlane", k_rdlane, out, blocks, 32);
    run("fma dep", k_dep, out, blocks, 32);
    run("cmp+saveexec+fma+or (2 VALU)", k_saveex, out, blocks, 16);
    run("fma+salu (2 VALU)", k_salu, out, blocks, 16);
    return 0;
}

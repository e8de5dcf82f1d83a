// Measures the FP32 vector issue rate of the GPU: independent v_fma_f32 chains and
// independent v_pk_fma_f32 chains, 8 waves per SIMD, no memory traffic in the loop.
// Build: hipcc -O3 --offload-arch=gfx950 tools/valu_peak.hip -o tools/valu_peak
#include <hip/hip_runtime.h>
#include <cstdio>

typedef float v2f __attribute__((ext_vector_type(2)));

__global__ __launch_bounds__(256) void fma_loop(float *out, int iters, float a, float b)
{
    float x0 = threadIdx.x, x1 = x0 + 1, x2 = x0 + 2, x3 = x0 + 3, x4 = x0 + 4, x5 = x0 + 5, x6 = x0 + 6, x7 = x0 + 7;
    for (int i = 0; i < iters; i++) {
#pragma unroll
        for (int k = 0; k < 8; k++) {
            x0 = __builtin_fmaf(x0, a, b); x1 = __builtin_fmaf(x1, a, b); x2 = __builtin_fmaf(x2, a, b); x3 = __builtin_fmaf(x3, a, b);
            x4 = __builtin_fmaf(x4, a, b); x5 = __builtin_fmaf(x5, a, b); x6 = __builtin_fmaf(x6, a, b); x7 = __builtin_fmaf(x7, a, b);
        }
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = x0 + x1 + x2 + x3 + x4 + x5 + x6 + x7;
}

__global__ __launch_bounds__(256) void pkfma_loop(float *out, int iters, float a, float b)
{
    v2f va = {a, a}, vb = {b, b};
    v2f x0 = {(float)threadIdx.x, 1.f}, x1 = x0 + 1.f, x2 = x0 + 2.f, x3 = x0 + 3.f, x4 = x0 + 4.f, x5 = x0 + 5.f, x6 = x0 + 6.f, x7 = x0 + 7.f;
    for (int i = 0; i < iters; i++) {
#pragma unroll
        for (int k = 0; k < 8; k++) {
            x0 = __builtin_elementwise_fma(x0, va, vb); x1 = __builtin_elementwise_fma(x1, va, vb);
            x2 = __builtin_elementwise_fma(x2, va, vb); x3 = __builtin_elementwise_fma(x3, va, vb);
            x4 = __builtin_elementwise_fma(x4, va, vb); x5 = __builtin_elementwise_fma(x5, va, vb);
            x6 = __builtin_elementwise_fma(x6, va, vb); x7 = __builtin_elementwise_fma(x7, va, vb);
        }
    }
    v2f s = x0 + x1 + x2 + x3 + x4 + x5 + x6 + x7;
    out[blockIdx.x * blockDim.x + threadIdx.x] = s.x + s.y;
}

__global__ __launch_bounds__(256) void mixed_loop(float *out, int iters, float a, float b)
{
    // mul / add / fma / sub mix with a compare+select, like a primitive test
    float x0 = threadIdx.x, x1 = x0 + 1, x2 = x0 + 2, x3 = x0 + 3;
    for (int i = 0; i < iters; i++) {
#pragma unroll
        for (int k = 0; k < 16; k++) {
            x0 = x0 * a; x1 = x1 + b; x2 = __builtin_fmaf(x2, a, b); x3 = x3 - x0;
            x0 = x0 > 1e30f ? b : x0;
        }
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = x0 + x1 + x2 + x3;
}

int main()
{
    hipDeviceProp_t p;
    hipGetDeviceProperties(&p, 0);
    printf("%s: %d CUs, %d MHz\n", p.name, p.multiProcessorCount, p.clockRate / 1000);
    const int blocks = p.multiProcessorCount * 8, threads = 256, iters = 20000;
    float *out;
    hipMalloc(&out, (size_t)blocks * threads * sizeof(float));
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    for (int which = 0; which < 3; which++) {
        for (int rep = 0; rep < 3; rep++) {
            hipEventRecord(e0);
            if (which == 0) fma_loop<<<blocks, threads>>>(out, iters, 1.0000001f, 1e-9f);
            else if (which == 1) pkfma_loop<<<blocks, threads>>>(out, iters, 1.0000001f, 1e-9f);
            else mixed_loop<<<blocks, threads>>>(out, iters, 1.0000001f, 1e-9f);
            hipEventRecord(e1);
            hipEventSynchronize(e1);
            float ms;
            hipEventElapsedTime(&ms, e0, e1);
            double lanes = (double)blocks * threads;
            if (which == 0) printf("v_fma_f32   : %.3f ms, %.1f TFLOP/s, %.2f G wave-instr/s\n", ms, lanes * iters * 64 * 2 / ms / 1e9, lanes / 64 * iters * 64 / ms / 1e6);
            else if (which == 1) printf("v_pk_fma_f32: %.3f ms, %.1f TFLOP/s, %.2f G wave-instr/s\n", ms, lanes * iters * 64 * 4 / ms / 1e9, lanes / 64 * iters * 64 / ms / 1e6);
            else printf("mixed       : %.3f ms, %.2f G wave-instr/s (5 VALU per step incl. cmp+cndmask as 2)\n", ms, lanes / 64 * iters * 16 * 6 / ms / 1e6);
        }
    }
    return 0;
}

/*
 * bdpt_slice_probe.hip -- not part of the library.  Three kernels that hold the hand-off functions of bdpt_slice.h
 * and nothing else, so that tools/check_slice_isa.py can read the instructions they compile to: the writer side
 * (state stores of 4, 8 and 16 bytes, the wait, the barrier, the tile's word), and the reader side (the poll, the
 * acquire, the barrier).
 */
#include "bdpt_slice.h"

using namespace bdpt;

extern "C" __global__ void bdpt_probe_state_stores(uint32_t *a, uint2 *b, uint4 *c, float4 *d, uint32_t v)
{
    state_store(true, a + threadIdx.x, v);
    state_store(true, b + threadIdx.x, make_uint2(v, v + 1u));
    state_store(true, c + threadIdx.x, make_uint4(v, v + 1u, v + 2u, v + 3u));
    state_store(true, d + threadIdx.x, make_float4(1.0f, 2.0f, 3.0f, (float)v));
}

extern "C" __global__ void bdpt_probe_slice_end(bdpt_kargs K, uint4 *c)
{
    Slice sl;
    sl.shared = true; sl.abandoned = false; sl.tile = (int)blockIdx.x; sl.run = 0; sl.s0 = 0; sl.s1 = 1;
    state_store(true, c + blockIdx.x * blockDim.x + threadIdx.x, make_uint4(1u, 2u, 3u, threadIdx.x));
    slice_end(K, sl);
}

extern "C" __global__ void bdpt_probe_slice_begin(bdpt_kargs K, const uint4 *c, uint4 *out)
{
    const Slice sl = slice_begin(K);
    if (sl.abandoned) return;
    out[sl.tile * blockDim.x + threadIdx.x] = c[sl.tile * blockDim.x + threadIdx.x];
}

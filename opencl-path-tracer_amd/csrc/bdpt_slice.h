/*
 * bdpt_slice.h -- the run-to-run hand-off of a sliced launch: which run of which tile a workgroup takes, how it waits
 * for the run before it, and how pixel state is stored so that a workgroup on another die reads it.
 * Included by bdpt_kernels.hip (the kernels) and by bdpt_slice_probe.hip (the build-time check of the instructions
 * these functions compile to).
 */
#ifndef BDPT_SLICE_H
#define BDPT_SLICE_H

#include "bdpt_layout.h"

#ifndef BDPT_DEV
#define BDPT_DEV __device__ __forceinline__
#endif

namespace bdpt {

/*
 * A launch in slices.
 *
 * A lane keeps its pixel for the whole launch, so a workgroup's run time grows with the sample
 * count, and a launch ends with the machine draining: the last workgroups run on a chip that is
 * mostly idle, for about as long as one workgroup takes (13.6 of 90 ms for 512 samples of the
 * Cornell box at 1080p).  A sliced launch cuts the samples of every workgroup's tile into
 * n_chunks runs of chunk_spp samples and starts n_chunks times as many workgroups, each of which
 * takes one run of one tile: the drain shrinks with the run.
 *
 * Which run of which tile a workgroup takes is decided by a ticket drawn from a counter when it
 * starts (ticket t: run t / n_tiles of tile t % n_tiles), not by its number in the grid, so the
 * run before it on the same tile belongs to a workgroup that drew its ticket earlier and is
 * therefore running or done, whatever order the hardware starts workgroups in.  It waits for that
 * workgroup's "tile done up to run r" word before it reads the pixel state the run left in memory
 * (accumulator, generator word, sample cursor, parity buffers), and publishes its own when it has
 * written them.  A run is to the samples exactly what a launch of its own would be, except that
 * the sample cursor carries over, so the order of the samples of a launch -- and with it every
 * bit of the result -- is that of the unsliced launch.
 */
struct Slice {
    int tile;               /* workgroup-sized tile of the launch's grid, row by row */
    int run;
    int s0, s1;             /* the samples s0 <= s < s1 of the launch's order, for every pixel of the tile */
    bool shared;            /* the launch is sliced: pixel state written here is read by another workgroup of this launch */
    bool abandoned;         /* the run before this one never completed: this workgroup must touch nothing (see slice_begin) */
};

typedef __attribute__((address_space(1))) uint32_t slice_u32;
typedef __attribute__((address_space(1))) unsigned long long slice_u64;
#define BDPT_SLICE_HEADER 4                 /* words in front of the per-tile words: ticket counter, give-up mark, padding */

/*
 * Waiting for the run before.  The holder of ticket t - n_tiles is resident (it has drawn its ticket) and waits only
 * for lower tickets, so the wait ends.  The guard against a fault elsewhere is a limit in wall-clock time that the host
 * derives from how long the launch is expected to take (K.slice_wait_ticks, steady-counter ticks): a workgroup that
 * waits longer gives up.  It then raises the launch's give-up mark (slice_sync[1], which every waiting workgroup also
 * watches, so the rest of the launch drains at once) and the context's sticky mark in host memory (K.slice_abort, which
 * the host checks whenever it has waited for the stream), and ends WITHOUT loading, tracing or storing anything: state
 * that is not ready is never read and nothing stale is written over the tile.  The image is then incomplete, and the
 * host says so; it is never silently wrong.
 */
BDPT_DEV Slice slice_begin(const bdpt_kargs &K)
{
    Slice sl;
    sl.shared = K.n_chunks > 1;
    sl.abandoned = false;
    if (!sl.shared) {
        sl.tile = (int)(blockIdx.y * gridDim.x + blockIdx.x);
        sl.run = 0;
        sl.s0 = 0;
        sl.s1 = K.spp;
        return sl;
    }
    __shared__ uint32_t ticket, gave_up;
    if (threadIdx.x == 0) {
        slice_u32 *sync = (slice_u32 *)(uintptr_t)K.slice_sync;
        const uint32_t t = __hip_atomic_fetch_add(sync, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const uint32_t run = t / (uint32_t)K.n_tiles, tile = t % (uint32_t)K.n_tiles;
        uint32_t polls = 0, stop = 0;
        const unsigned long long t_begin = wall_clock64();
        while (__hip_atomic_load(sync + BDPT_SLICE_HEADER + tile, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < run) {
            if ((polls++ & 63u) == 0u) {                    /* at once, then every 64 polls (some 100 microseconds): the mark and the clock */
                if (__hip_atomic_load(sync + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u) { stop = 1u; break; }
                if (wall_clock64() - t_begin > K.slice_wait_ticks) {
                    __hip_atomic_store(sync + 1, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    __hip_atomic_store(K.slice_abort, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                    stop = 1u;
                    break;
                }
            }
            __builtin_amdgcn_s_sleep(32);
        }
        /* one invalidate of this compute unit's vector cache, after the word matched; the wait holds the barrier
         * below until it has completed, and then every wave of the workgroup may load the state */
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        ticket = t;
        gave_up = stop;
    }
    __syncthreads();
    const uint32_t t = ticket;
    sl.abandoned = gave_up != 0u;
    sl.tile = (int)(t % (uint32_t)K.n_tiles);
    sl.run = (int)(t / (uint32_t)K.n_tiles);
    sl.s0 = sl.run * K.chunk_spp;
    sl.s1 = min(K.spp, sl.s0 + K.chunk_spp);
    return sl;
}

/* Every thread of the workgroup, after its last store of pixel state.  The state of a sliced launch is stored
 * write-through (state_store below), so it needs no write-back of the second-level cache: every wave waits for its
 * own stores, the workgroup meets, one lane raises the tile's word. */
BDPT_DEV void slice_end(const bdpt_kargs &K, const Slice &sl)
{
    if (!sl.shared) return;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (threadIdx.x == 0)
        __hip_atomic_store((slice_u32 *)(uintptr_t)K.slice_sync + BDPT_SLICE_HEADER + sl.tile, (uint32_t)sl.run + 1u,
                           __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

/*
 * Stores of pixel state (accumulator, generator word, cursor, parity buffers).  shared: another workgroup of this
 * launch reads the bytes, on whichever of the eight dies (each with its own second-level cache) it runs.
 *
 * What the hand-off rests on, instruction by instruction (gfx942-class memory model; tools/slice_handoff.hip runs the
 * protocol below with each store form on 16 200 tiles x 16 runs and counts stale reads):
 *   writer  every state store carries the sc1 cache-policy bit: global_store_dword / _dwordx2 ... sc1, which is what a
 *           relaxed agent-scope atomic store compiles to (16 bytes go as two of 8: the compiler has no wider
 *           atomic; one global_store_dwordx4 ... sc1 written out as assembly is just as safe in the tool and bills
 *           14 % fewer bytes written, but the statement keeps the compiler from scheduling around it and a 1024-
 *           sample launch of the Cornell box takes 100.5 instead of 99.2 ms: not taken).  sc1 makes the store write through the die's second-level
 *           cache to memory; a plain store leaves a dirty line there that another die does not see (42 % of the reads
 *           stale in the tool), and the nt bit -- aux value 2 of the buffer-store builtins, where sc1 is 16 -- does
 *           not write through either (the same 42 %: the stale reads recorded for "16-byte stores with the sc1
 *           policy" in an earlier build were that operand).
 *           Then s_waitcnt vmcnt(0) in every wave: the stores have been acknowledged by memory.  Then the barrier,
 *           then ONE relaxed agent-scope store of the tile's word (sc1 again).  This is the documented code sequence
 *           of a release store at agent scope minus its buffer_wbl2 sc1, the write-back of the whole second-level
 *           cache that exists for plain stores; with nothing but write-through stores to publish there is nothing
 *           for it to write back (as a release store the tool's launch takes 6.6 ms instead of 3.1).
 *   reader  relaxed agent-scope loads of the word (sc1: they bypass the caches), then one agent-scope acquire fence
 *           = buffer_inv sc1, which drops every line of this die's caches that another die may have rewritten,
 *           s_waitcnt vmcnt(0), the barrier; after that plain loads.  A relaxed load that reads the value followed
 *           by an acquire fence is the memory model's own reader side.
 * tools/check_slice_isa.py compiles bdpt_slice_probe.hip (these functions and nothing else) with the library's flags
 * and fails the build when an instruction of the sequence above is missing or has other policy bits.
 */
BDPT_DEV void state_store(bool shared, uint32_t *p, uint32_t v)
{
    if (shared) __hip_atomic_store((slice_u32 *)(uintptr_t)p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    else *p = v;
}
BDPT_DEV void state_store(bool shared, uint2 *p, uint2 v)
{
    if (shared) __hip_atomic_store((slice_u64 *)(uintptr_t)p, ((unsigned long long)v.y << 32) | v.x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    else *p = v;
}
BDPT_DEV void state_store(bool shared, uint4 *p, uint4 v)
{
    if (shared) {
        state_store(true, (uint2 *)p, make_uint2(v.x, v.y));
        state_store(true, (uint2 *)p + 1, make_uint2(v.z, v.w));
    } else *p = v;
}
BDPT_DEV void state_store(bool shared, float4 *p, float4 v)
{
    state_store(shared, (uint4 *)p, make_uint4(__float_as_uint(v.x), __float_as_uint(v.y), __float_as_uint(v.z), __float_as_uint(v.w)));
}

}  // namespace bdpt

#endif

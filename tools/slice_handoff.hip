// Which store forms make the run-to-run hand-off of a sliced launch (slice_begin / slice_end / state_store in
// csrc/bdpt_kernels.hip) safe on gfx950, and what each costs.
//
// The protocol, as in the library: a launch of n_tiles * n_runs one-wavefront workgroups; a workgroup draws a ticket
// (run = t / n_tiles, tile = t % n_tiles), polls the tile's word with relaxed agent-scope loads until it reaches
// `run`, issues one agent-scope acquire fence, loads the tile's state (one uint4 per lane, plain loads), checks that
// it is what run - 1 stored, stores its own stamp with the store form under test, waits for its stores
// (s_waitcnt vmcnt(0)), meets at the barrier, and raises the tile's word with a relaxed agent-scope store.
// Consecutive runs of a tile land on whichever of the eight dies the hardware picks, each with its own L2.
//
// Printed per store form: stale or torn reads seen (0 = the form is safe under this protocol) and kernel time.
// Build: hipcc -O3 --offload-arch=gfx950 tools/slice_handoff.hip -o tools/slice_handoff
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <vector>

typedef __attribute__((address_space(1))) uint32_t g_u32;
typedef __attribute__((address_space(1))) unsigned long long g_u64;
typedef uint32_t v4u __attribute__((ext_vector_type(4)));

enum {
    ST_ATOMIC_X2 = 0,      // two 8-byte relaxed agent-scope atomic stores (global_store_dwordx2 ... sc1): what the library uses
    ST_ASM_X4_SC1 = 1,     // one global_store_dwordx4 ... sc1 written as inline assembly
    ST_PLAIN_X4 = 2,       // one plain 16-byte store, relaxed flag: expected to fail (dirty lines stay in the writer's L2)
    ST_PLAIN_X4_RELEASE = 3, // plain 16-byte store, flag raised with a release store at agent scope (buffer_wbl2 sc1): the form the memory model defines
    ST_BUFFER_X4_AUX16 = 4,  // raw buffer store builtin, aux = 16 (the sc1 bit of gfx940-class cache policy operands)
    ST_BUFFER_X4_AUX2 = 5,   // raw buffer store builtin, aux = 2 (that bit is nt, not sc1)
    ST_ATOMIC_X4_PAIR_SC0SC1 = 6, // asm x4 store with sc0 sc1 (system scope)
    N_FORMS = 7
};

__device__ __forceinline__ uint4 stamp(uint32_t pix, uint32_t run, uint32_t salt)
{
    const uint32_t a = pix * 2654435761u + run * 40503u + salt;
    return make_uint4(a, a ^ 0x9E3779B9u, a * 3u + run, ~a);
}

template <int FORM>
__device__ __forceinline__ void store_state(uint4 *p, uint4 v)
{
    if (FORM == ST_ATOMIC_X2) {
        __hip_atomic_store((g_u64 *)(uintptr_t)p, ((unsigned long long)v.y << 32) | v.x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store((g_u64 *)(uintptr_t)p + 1, ((unsigned long long)v.w << 32) | v.z, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    } else if (FORM == ST_ASM_X4_SC1) {
        const v4u x = {v.x, v.y, v.z, v.w};
        asm volatile("global_store_dwordx4 %0, %1, off sc1" ::"v"(p), "v"(x) : "memory");
    } else if (FORM == ST_ATOMIC_X4_PAIR_SC0SC1) {
        const v4u x = {v.x, v.y, v.z, v.w};
        asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1" ::"v"(p), "v"(x) : "memory");
    } else if (FORM == ST_BUFFER_X4_AUX16 || FORM == ST_BUFFER_X4_AUX2) {
        const v4u x = {v.x, v.y, v.z, v.w};
        // a buffer resource over the whole address space from p's base: base = p, stride 0, num_records = 16, flags for raw access
        const uint64_t base = (uint64_t)(uintptr_t)p;
        const __attribute__((ext_vector_type(4))) int rsrc = {(int)(uint32_t)base, (int)(uint32_t)(base >> 32) & 0xffff, 16, 0x00020000};
        __builtin_amdgcn_raw_buffer_store_b128(x, __builtin_amdgcn_make_buffer_rsrc((void *)p, 0, 16, 0x00020000), 0, 0, FORM == ST_BUFFER_X4_AUX16 ? 16 : 2);
        (void)rsrc;
    } else {
        *p = v;
    }
}

template <int FORM>
__global__ __launch_bounds__(64) void handoff(uint32_t *sync, uint4 *state, int n_tiles, uint32_t salt, unsigned long long *errors,
                                              uint32_t *first_bad)
{
    __shared__ uint32_t ticket;
    if (threadIdx.x == 0) {
        const uint32_t t = __hip_atomic_fetch_add((g_u32 *)(uintptr_t)sync, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const uint32_t run = t / (uint32_t)n_tiles, tile = t % (uint32_t)n_tiles;
        uint32_t polls = 0;
        while (__hip_atomic_load((g_u32 *)(uintptr_t)sync + 4 + tile, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < run) {
            if (++polls > (1u << 22)) { atomicAdd(errors + 1, 1ull); break; }
            __builtin_amdgcn_s_sleep(32);
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        ticket = t;
    }
    __syncthreads();
    const uint32_t t = ticket, run = t / (uint32_t)n_tiles, tile = t % (uint32_t)n_tiles;
    const uint32_t pix = tile * 64u + threadIdx.x;
    const uint4 got = state[pix];
    const uint4 want = run == 0 ? make_uint4(pix, 0u, 0u, 0u) : stamp(pix, run - 1, salt);
    if (got.x != want.x || got.y != want.y || got.z != want.z || got.w != want.w) {
        if (atomicAdd(errors, 1ull) == 0ull) { first_bad[0] = tile; first_bad[1] = run; first_bad[2] = threadIdx.x; first_bad[3] = got.x; first_bad[4] = want.x; }
    }
    // some work of varying length between the load and the store, so that runs overlap in every way
    uint32_t h = got.x;
    for (uint32_t i = 0; i < 64u + (tile & 63u) * 8u; i++) h = h * 1664525u + 1013904223u;
    uint4 mine = stamp(pix, run, salt);
    if (h == 0xdeadbeefu) mine.x ^= 1u;                  // keeps the loop alive; practically never taken
    store_state<FORM>(&state[pix], mine);
    if (FORM == ST_PLAIN_X4_RELEASE) {
        __syncthreads();
        if (threadIdx.x == 0)
            __hip_atomic_store((g_u32 *)(uintptr_t)sync + 4 + tile, run + 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
    } else {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (threadIdx.x == 0)
            __hip_atomic_store((g_u32 *)(uintptr_t)sync + 4 + tile, run + 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}

__global__ void init_state(uint4 *state, int n) { const int i = blockIdx.x * blockDim.x + threadIdx.x; if (i < n) state[i] = make_uint4((uint32_t)i, 0u, 0u, 0u); }

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { std::printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

template <int FORM>
static int run_form(const char *name, int n_tiles, int n_runs, int reps)
{
    uint32_t *sync, *first_bad;
    uint4 *state;
    unsigned long long *errors;
    CHECK(hipMalloc((void **)&sync, (4 + n_tiles) * sizeof(uint32_t)));
    CHECK(hipMalloc((void **)&state, (size_t)n_tiles * 64 * sizeof(uint4)));
    CHECK(hipMalloc((void **)&errors, 2 * sizeof(unsigned long long)));
    CHECK(hipMalloc((void **)&first_bad, 8 * sizeof(uint32_t)));
    CHECK(hipMemset(errors, 0, 2 * sizeof(unsigned long long)));
    CHECK(hipMemset(first_bad, 0, 8 * sizeof(uint32_t)));
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0));
    CHECK(hipEventCreate(&e1));
    float total_ms = 0.f;
    for (int r = 0; r < reps; r++) {
        CHECK(hipMemset(sync, 0, (4 + n_tiles) * sizeof(uint32_t)));
        hipLaunchKernelGGL(init_state, dim3((n_tiles * 64 + 255) / 256), dim3(256), 0, 0, state, n_tiles * 64);
        CHECK(hipEventRecord(e0));
        hipLaunchKernelGGL((handoff<FORM>), dim3((unsigned)(n_tiles * n_runs)), dim3(64), 0, 0, sync, state, n_tiles, (uint32_t)(r * 7919 + 1), errors, first_bad);
        CHECK(hipEventRecord(e1));
        CHECK(hipDeviceSynchronize());
        float ms = 0.f;
        CHECK(hipEventElapsedTime(&ms, e0, e1));
        total_ms += ms;
    }
    unsigned long long err[2];
    uint32_t bad[8];
    CHECK(hipMemcpy(err, errors, sizeof err, hipMemcpyDeviceToHost));
    CHECK(hipMemcpy(bad, first_bad, sizeof bad, hipMemcpyDeviceToHost));
    std::printf("%-34s %9llu bad reads of %llu   gave up %llu   %.3f ms per launch", name, err[0],
                (unsigned long long)n_tiles * 64ull * n_runs * reps, err[1], total_ms / reps);
    if (err[0]) std::printf("   first: tile %u run %u lane %u got %08x want %08x", bad[0], bad[1], bad[2], bad[3], bad[4]);
    std::printf("\n");
    (void)hipFree(sync); (void)hipFree(state); (void)hipFree(errors); (void)hipFree(first_bad);
    return 0;
}

int main(int argc, char **argv)
{
    const int n_tiles = argc > 1 ? std::atoi(argv[1]) : 16200, n_runs = argc > 2 ? std::atoi(argv[2]) : 16, reps = argc > 3 ? std::atoi(argv[3]) : 5;
    std::printf("hand-off of %d tiles x %d runs (one wavefront each), %d launches per store form\n", n_tiles, n_runs, reps);
    if (run_form<ST_ATOMIC_X2>("2 x 8-byte atomic store, sc1", n_tiles, n_runs, reps)) return 1;
    if (run_form<ST_ASM_X4_SC1>("16-byte store, sc1 (asm)", n_tiles, n_runs, reps)) return 1;
    if (run_form<ST_ATOMIC_X4_PAIR_SC0SC1>("16-byte store, sc0 sc1 (asm)", n_tiles, n_runs, reps)) return 1;
    if (run_form<ST_BUFFER_X4_AUX16>("16-byte buffer store, aux 16", n_tiles, n_runs, reps)) return 1;
    if (run_form<ST_BUFFER_X4_AUX2>("16-byte buffer store, aux 2", n_tiles, n_runs, reps)) return 1;
    if (run_form<ST_PLAIN_X4_RELEASE>("plain 16-byte store, release flag", n_tiles, n_runs, reps)) return 1;
    if (run_form<ST_PLAIN_X4>("plain 16-byte store, relaxed flag", n_tiles, n_runs, reps)) return 1;
    return 0;
}

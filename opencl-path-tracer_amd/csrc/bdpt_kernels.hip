/*
 * bdpt_kernels.hip -- gfx950 kernels of the BDPT sample path.
 *
 * The unit of work is one path = one execution of the reference's
 * bi_directional_path_trace (kernel0.cl:571-864) for one pixel.  A lane owns
 * one pixel for the whole launch and runs `spp` consecutive samples of it with
 * generator state, running accumulator and path state in registers; the
 * accumulator is read and written once per launch (one float4 per pixel).
 *
 * Lane -> pixel: a wavefront covers an 8x8 pixel tile (lane & 7 = x, lane >> 3
 * = y), a workgroup one such tile or two side by side (bdpt_kargs.block), so that
 * the eye rays of a wavefront stay coherent and silhouette edges cut through as
 * few wavefronts as possible.
 *
 * Built with -ffp-contract=off: the only fused operations are the ones the
 * numeric contract (include/bdpt_clmath.h) writes out.
 */
#include "bdpt_device.h"
#include "bdpt_slice.h"
#include "bdpt_kernels.h"
#include "bdpt_bvh.h"

#include <algorithm>

/* Workgroup: one wavefront (one 8x8 pixel tile) or two (two tiles side by side), chosen per launch (launch_block).
 * One is faster where it fits (1080p Cornell box, 1024 spp: 129.9 against 133.0 ms; 100 000 triangles 915 against
 * 944 ms per 4 samples): a workgroup's slot is free as soon as its wavefront ends, and the barriers of a sliced launch
 * are no-ops.  The LDS kernels copy the scene once per workgroup, so larger scenes take two wavefronts per copy. */
#define BDPT_BLOCK_MAX 128

namespace bdpt {

/* the sample cursor of a pixel between two runs of a sliced launch */
BDPT_DEV void cursor_load(SampleCursor &c, const bdpt_kargs &K, size_t pix)
{
    const uint4 v = K.slice_cursor[pix];
    c.cls = (int)v.x; c.head = (int)v.y; c.st_head = v.z; c.pending = v.w;
}
BDPT_DEV void cursor_store(const SampleCursor &c, const bdpt_kargs &K, size_t pix)
{
    state_store(true, &K.slice_cursor[pix], make_uint4((uint32_t)c.cls, (uint32_t)c.head, c.st_head, c.pending));
}

/*
 * Class order from a table.  Finding a pixel's samples of one class by classifying its sample indices (cursor_rounds)
 * looks at every index once per class, and a look is five integer multiplications: 9 % of a 1024-sample launch of the
 * Cornell box.  Here a pixel's indices are classified once, before its first sample, 32 at a time, and kept as one
 * bit mask per class and 32 indices (K.class_masks); taking the next sample is then a count of trailing zeros, and
 * a load when a word is used up.  The samples come in the same order, class by class and index ascending.
 *
 * plane[k] collects, index by index, whether roulette draw k stops the walk; class c < max_l is "draw c - 1 stops and
 * none before it", class max_l is "none of the first max_l - 1 stops".  N as in roulette_class: N + 1 >= max_l.
 */
constexpr uint32_t kWordStride = pow_u32_c(kSampleStride, 32);

template <int N>
BDPT_DEV void cursor_fill_masks(const bdpt_kargs &K, bool shared, size_t pix, uint32_t seed)
{
    const bool five = sampler_draws_four(K.sampler);
    const uint32_t step = five ? lcg_pow(5) : lcg_pow(4), lead = five ? lcg_pow(10) : lcg_pow(9);
    const size_t npix = (size_t)K.width * K.height;
    uint32_t st = cursor_first_state(K, seed);
    for (int w = 0; w < K.mask_words; w++) {
        uint32_t plane[N];
#pragma unroll
        for (int k = 0; k < N; k++) plane[k] = 0u;
        for (int b = 0; b < 32; b++) {
            uint32_t x = st * lead;
#pragma unroll
            for (int k = 0; k < N; k++) {
                plane[k] = (plane[k] >> 1) | (x > K.fate_above ? 0x80000000u : 0u);      /* kernel0.cl:788 */
                x *= step;
            }
            st *= kSampleStride;
        }
        const int left = K.spp - w * 32;                                       /* at least 1 */
        uint32_t open = left >= 32 ? ~0u : (1u << left) - 1u;                  /* indices of the launch that no earlier draw has stopped */
#pragma unroll
        for (int c = 1; c <= N + 1; c++) {
            if (c <= K.max_l) {
                const uint32_t stops_here = c <= N ? plane[c <= N ? c - 1 : 0] : 0u;
                const uint32_t m = c == K.max_l ? open : open & stops_here;
                state_store(shared, &K.class_masks[((size_t)(c - 1) * K.mask_words + w) * npix + pix], m);
                open &= ~stops_here;
            }
        }
    }
}

/* cursor_next for a launch with the table: c.head = next word to load, c.st_head = start state of sample 32 * head,
 * c.pending = what is left of word head - 1 */
BDPT_DEV bool cursor_next_masks(SampleCursor &c, const bdpt_kargs &K, size_t pix, const uint32_t *seed_w, uint32_t &start)
{
    const size_t npix = (size_t)K.width * K.height;
    while (__builtin_amdgcn_ballot_w64(c.pending == 0u && c.cls >= 1) != 0) {
        const bool empty = c.pending == 0u && c.cls >= 1;
        const bool used_up = empty && c.head >= K.mask_words;                 /* no more samples of this class */
        if (__builtin_amdgcn_ballot_w64(used_up) != 0) {
            asm volatile("");                                                 /* a real branch: a handful of times per launch */
            if (used_up) {
                c.cls--;
                c.head = 0;
                c.st_head = cursor_first_state(K, *seed_w);
            }
            continue;                                                         /* the lane may have run out of classes */
        }
        uint32_t word = 0u;
        if (empty) word = K.class_masks[((size_t)(c.cls - 1) * K.mask_words + c.head) * npix + pix];
        c.pending = empty ? word : c.pending;
        c.head += empty ? 1 : 0;
        c.st_head = empty ? c.st_head * kWordStride : c.st_head;
    }
    if (c.cls < 1) return false;
    const int b = __builtin_ctz(c.pending);
    c.pending &= c.pending - 1u;
    start = c.st_head * kStrideInvPow.v[32 - b];             /* back from sample 32 * head to sample 32 * (head - 1) + b */
    return true;
}

/* the next sample of a pixel, whichever way the launch keeps its order */
BDPT_DEV bool sample_next(SampleCursor &c, const bdpt_kargs &K, size_t pix, uint32_t &start)
{
    if (K.class_masks && c.cls >= 0) return cursor_next_masks(c, K, pix, &K.seed_w[pix], start);
    return cursor_next(c, K, &K.seed_w[pix], start);
}

/* what a run leaves in memory for the next run or the next launch */
template <bool AOV>
BDPT_DEV void pixel_state_store(const bdpt_kargs &K, const Slice &sl, size_t pix, const float acc[4], uint32_t w, const SampleCursor &cur,
                                const int eye_id[2], uint32_t hash, const float col[4], uint32_t draws, uint32_t rays)
{
    state_store(sl.shared, &K.accum[pix], make_float4(acc[0], acc[1], acc[2], acc[3]));   /* kernel0.cl:863 */
    if (K.rng_mode == BDPT_RNG_STREAM) state_store(sl.shared, &K.seed_w[pix], w);        /* :861 */
    if (sl.s1 < K.spp) cursor_store(cur, K, pix);
    if (AOV) {
        state_store(sl.shared, (uint2 *)&K.aov_eye_hit[pix * 2], make_uint2((uint32_t)eye_id[0], (uint32_t)eye_id[1]));
        state_store(sl.shared, &K.aov_hit_hash[pix], hash);
        state_store(sl.shared, &K.aov_radiance[pix], make_float4(col[0], col[1], col[2], col[3]));
        state_store(sl.shared, &K.aov_rng_draws[pix], draws);
        state_store(sl.shared, &K.aov_rays[pix], rays);
    }
}

/* Pixel of this lane in tile `tile`.  Whole region: the tiles are a workgroup's pixels (8x8 or 16x8), wg_x per row.  Interleaved tile
 * split: the tiles enumerate, split tile after split tile, the workgroups of the split tiles this launch owns. */
BDPT_DEV void lane_pixel(const bdpt_kargs &K, int tile, int &x, int &y)
{
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    int bx, by, ox = K.rx0, oy = K.ry0;
    if (K.tile_stride > 1) {
        const int bw = K.tile_w / (K.block >> 3), per_tile = bw * (K.tile_h / 8);
        const int owned = tile / per_tile, local = tile % per_tile;
        const int t = K.tile_phase + owned * K.tile_stride;
        ox += (t % K.tiles_x) * K.tile_w;
        oy += (t / K.tiles_x) * K.tile_h;
        bx = local % bw;
        by = local / bw;
    } else if (K.n_chunks <= 1) {
        bx = (int)blockIdx.x;
        by = (int)blockIdx.y;
    } else {
        bx = tile % K.wg_x;
        by = tile / K.wg_x;
    }
    x = ox + bx * (K.block >> 3) + wv * 8 + (lane & 7);
    y = oy + by * 8 + (lane >> 3);
}

/* ------------------------------------------------------------------------- */
/* Direct variant: every lane walks the primitive arrays itself.  All lanes of
 * a wavefront read the same record at the same time, so the reads are
 * wave-uniform (scalar loads served by the constant cache).                    */
/* ------------------------------------------------------------------------- */

BDPT_DEV Hit closest_hit_direct(const bdpt_dev_scene &S, clm_f3 o, clm_f3 d, int otype, int oidx)
{
    Best b;
    best_init(b);                                                         /* kernel0.cl:374, :384-385 */
    const int skip_s = otype == 0 ? oidx : -1;                            /* :390 */
    const int skip_p = otype == 1 ? oidx : -1;                            /* :415 */
    const int skip_t = otype == 2 ? oidx : -1;                            /* :436 */
    for (int i = 0; i < S.n_spheres; i++) {
        float4 s = uload(S.sph, i);
        if (i != skip_s) test_sphere(b, s, i, o, d);
    }
    for (int i = 0; i < S.n_planes; i++) {
        float4 q0 = uload(S.pln, (size_t)i * BDPT_PLN_F4 + 0), q1 = uload(S.pln, (size_t)i * BDPT_PLN_F4 + 1);
        if (i != skip_p) test_plane(b, q0, q1, i, o, d);
    }
    for (int i = 0; i < S.n_triangles; i++) {
        const size_t r = (size_t)i * BDPT_TRI_F4;
        float4 q0 = uload(S.tri, r), q1 = uload(S.tri, r + 1), q2 = uload(S.tri, r + 2), q3 = uload(S.tri, r + 3),
               q4 = uload(S.tri, r + 4);
        float e2z = uload(S.tri, r + 5).x;
        /* pin the whole record here: otherwise the compiler sinks the second half of the load
         * into the edge tests and every triangle pays two scalar-load round trips instead of one */
        asm volatile("" ::"s"(q4.x), "s"(e2z));
        if (i != skip_t) test_triangle(b, q0, q1, q2, q3, q4, e2z, i, o, d);
    }
    return finish_hit(S, b, d);
}

/* ------------------------------------------------------------------------- */
/* LDS-resident variant.  The workgroup copies the pre-laid-out scene records
 * into LDS once (coalesced float4 loads), and every lane then walks them with
 * wave-uniform ds_read_b128 (all lanes read one address: a broadcast, no bank
 * conflicts).  The records land in VGPRs, so the arithmetic of the primitive
 * tests has no scalar-register operand: on gfx950 a VALU instruction with an
 * SGPR source issues in about 3.2 cycles per SIMD, one with VGPR sources only in
 * 2.0 (tools/valu_issue.hip).                                                  */
/* ------------------------------------------------------------------------- */

extern __shared__ float4 bdpt_lds[];   /* [n_triangles*6 | (two-stage kernel: n_triangles*5) | n_spheres | n_planes*2 | n_surfaces*2 | surface ids] */

/* offsets, in float4, of the parts that follow the triangle records */
struct LdsLayout {
    int cull, sph, pln, surf, ids;  /* cull records, spheres, planes, surface table, surface ids (uint32: triangles, spheres, planes) */
    int n_ids;                      /* number of surface ids */
};
template <bool CULL>
BDPT_DEV LdsLayout lds_layout(const bdpt_dev_scene &S)
{
    LdsLayout L;
    L.cull = S.n_triangles * BDPT_TRI_F4;
    L.sph = L.cull + (CULL ? S.cull_f4 : 0);
    L.pln = L.sph + S.n_spheres;
    L.surf = L.pln + S.n_planes * BDPT_PLN_F4;
    L.ids = L.surf + S.n_surfaces * 2;
    L.n_ids = (S.n_triangles + S.n_spheres + S.n_planes + 3) & ~3;   /* rounded up: what follows stays 16-byte aligned */
    return L;
}

template <bool CULL>
BDPT_DEV void lds_fill_scene(const bdpt_dev_scene &S)
{
    const LdsLayout L = lds_layout<CULL>(S);
    const int n_tri = S.n_triangles * BDPT_TRI_F4, n_pln = S.n_planes * BDPT_PLN_F4, n_surf = S.n_surfaces * 2;
    for (int i = threadIdx.x; i < n_tri; i += blockDim.x) bdpt_lds[i] = S.tri[i];
    if (CULL)
        for (int i = threadIdx.x; i < S.cull_f4; i += blockDim.x) bdpt_lds[L.cull + i] = S.tri_cull[i];
    for (int i = threadIdx.x; i < S.n_spheres; i += blockDim.x) bdpt_lds[L.sph + i] = S.sph[i];
    for (int i = threadIdx.x; i < n_pln; i += blockDim.x) bdpt_lds[L.pln + i] = S.pln[i];
    for (int i = threadIdx.x; i < n_surf; i += blockDim.x) bdpt_lds[L.surf + i] = S.surf[i];
    uint32_t *ids = (uint32_t *)(bdpt_lds + L.ids);
    for (int i = threadIdx.x; i < S.n_triangles; i += blockDim.x) ids[i] = S.tri_surf[i];
    for (int i = threadIdx.x; i < S.n_spheres; i += blockDim.x) ids[S.n_triangles + i] = S.sph_surf[i];
    for (int i = threadIdx.x; i < S.n_planes; i += blockDim.x) ids[S.n_triangles + S.n_spheres + i] = S.pln_surf[i];
    __syncthreads();
}

/* finish_hit (bdpt_device.h) reading the winner's record and surface id from LDS */
template <bool CULL>
BDPT_DEV Hit finish_hit_lds(const bdpt_dev_scene &S, const Best &b, clm_f3 d)
{
    const LdsLayout L = lds_layout<CULL>(S);
    const uint32_t *ids = (const uint32_t *)(bdpt_lds + L.ids);
    Hit h;
    h.type = b.type;
    h.idx = b.idx;
    h.pos = b.pos;
    h.n = clm_make3(0.0f, 0.0f, 0.0f);
    h.surf = 0;
    if (b.type == 0) {
        const float4 s = bdpt_lds[L.sph + b.idx];
        h.n = clm_normalize3(clm_sub3(clm_make3(s.x, s.y, s.z), b.pos));             /* :462 */
        h.surf = ids[S.n_triangles + b.idx];
    } else if (b.type == 1) {
        const float4 q = bdpt_lds[L.pln + b.idx * BDPT_PLN_F4];
        h.n = clm_make3(q.x, q.y, q.z);                                              /* :464 */
        h.surf = ids[S.n_triangles + S.n_spheres + b.idx];
    } else if (b.type == 2) {
        const float4 q = bdpt_lds[b.idx * BDPT_TRI_F4 + 5];
        h.n = clm_make3(q.y, q.z, q.w);                                              /* :467 */
        h.surf = ids[b.idx];
    }
    if (clm_dot3(d, h.n) > 0) h.n = clm_neg3(h.n);                                   /* :471 */
    return h;
}

template <bool SCATTERED>
BDPT_DEV Hit closest_hit_lds(const bdpt_dev_scene &S, clm_f3 o, clm_f3 d, int otype, int oidx)
{
    Best b;
    best_init(b);                                                         /* kernel0.cl:374, :384-385 */
    const int skip_s = otype == 0 ? oidx : -1;                            /* :390 */
    const int skip_p = otype == 1 ? oidx : -1;                            /* :415 */
    const int skip_t = otype == 2 ? oidx : -1;                            /* :436 */
    const LdsLayout L = lds_layout<false>(S);
    const int sph0 = L.sph, pln0 = L.pln;
    for (int i = 0; i < S.n_spheres; i++) {
        const float4 s = bdpt_lds[sph0 + i];
        if (i != skip_s) test_sphere(b, s, i, o, d);
    }
    for (int i = 0; i < S.n_planes; i++) {
        const float4 q0 = bdpt_lds[pln0 + i * BDPT_PLN_F4], q1 = bdpt_lds[pln0 + i * BDPT_PLN_F4 + 1];
        if (i != skip_p) test_plane(b, q0, q1, i, o, d);
    }
    for (int i = 0; i < S.n_triangles; i++) {
        const int r = i * BDPT_TRI_F4;
        const float4 q0 = bdpt_lds[r], q1 = bdpt_lds[r + 1], q2 = bdpt_lds[r + 2], q3 = bdpt_lds[r + 3],
                     q4 = bdpt_lds[r + 4];
        const float e2z = bdpt_lds[r + 5].x;
        /* pin the whole record here: one LDS round trip per triangle instead of one per test stage */
        asm volatile("" ::"v"(q1.x), "v"(q2.x), "v"(q3.x), "v"(q4.x), "v"(e2z));
        if (SCATTERED) test_triangle_scattered(b, q0, q1, q2, q3, q4, e2z, i, skip_t, o, d);
        else if (i != skip_t) test_triangle(b, q0, q1, q2, q3, q4, e2z, i, o, d);
    }
    return finish_hit_lds<false>(S, b, d);
}

/* ------------------------------------------------------------------------- */
/* Two-stage LDS-resident variant.  In the walk above a wavefront runs the whole
 * triangle test for a triangle as soon as one of its 64 rays passes the plane
 * culls, with the other lanes idle: for rays that point every way that is nearly
 * every triangle, although each ray hits one or two of them.  Here the walk is
 * split in two:
 *
 *  stage 1, all lanes together, no branches: a conservative test of every triangle.
 *    val = d.n is the reference's own value (kernel0.cl:286), so its cull (:288)
 *    is decided exactly.  The point p' where the ray meets the triangle's plane is
 *    then formed with approximate arithmetic (a precomputed a.n, v_rcp_f32) and
 *    its barycentric coordinates are read off precomputed edge planes (3 fma each
 *    instead of the 12 operations of a cross and a dot product; the third is what
 *    is left of the sum).  The triangle is dropped for this ray only if the
 *    smallest coordinate is negative by more than rounding can explain.
 *
 *    The margin.  Let u = 2^-24.  The reference's numerator (a-o).n and this
 *    stage's k - o.n each carry at most 4u (|a|_1 + |o|_1); divided by val that
 *    moves the point along the ray by 8u (|a|_1 + |o|_1) |d| / |val|; the quotient,
 *    the fma that forms the point, and the difference p - a add some u (|a| + L +
 *    |o|) (L = longest edge; a point the reference accepts lies within the
 *    triangle up to such terms).  A coordinate changes by at most 1/h per unit of
 *    point movement (h = the triangle's smallest height); the reference's own
 *    cross and dot product are good to 42 u L |e| against the doubled area, i.e.
 *    42 u L / h; the stored planes and their fma chain to 4 u sqrt(3) (2|a| + L) / h;
 *    the angle theta between the reference's float normal and the true one tilts
 *    the reference's expressions by theta L / h.  With |d|_inf / |val| >= 1/sqrt(3)
 *    all of it is below
 *        (104 u / h) (L + |a|_1 + |o|_1 + theta L / 3.5e-6) |d|_inf / |val| .
 *    Upload scales each triangle's coordinates by s = h / (104 u), takes C = the
 *    largest L + |a|_1 + theta L / 3.5e-6 of the scene, and the device compares the
 *    smallest scaled coordinate with -(C + |o|_1) |d|_inf / |val|, doubled for
 *    good measure.  A NaN or an infinity anywhere fails the comparison and keeps
 *    the triangle; triangles whose normal is unreliable have s = 0 and all-zero
 *    planes, which keeps them for every ray.  Each lane ends with a bit mask of
 *    the triangles it could not rule out.
 *
 *  stage 2, per lane: the reference's test (test_triangle_flat, the arithmetic of
 *    test_triangle_scattered) on the lane's remaining triangles, in ascending index
 *    order, each lane reading its own triangle's record from LDS.  A ray keeps one
 *    triangle, seldom two, so the wavefront makes one or two passes here instead
 *    of one per triangle.
 *
 * Every triangle the reference's test would accept survives stage 1, candidates
 * go through the reference's arithmetic in the reference's order, so results are
 * bit-identical with the other variants.                                       */
/* ------------------------------------------------------------------------- */

/* -DBDPT_PHASE_CLOCKS: an instrumented build for tools/phase_clocks.py.  Every wavefront adds up the clock ticks it
 * spends in each part of the search (sphere and plane walk, stage 1, stage 2, winner's record) and in the samples as a
 * whole; bdpt_phase_clocks() reads the sums.  Slower than the plain build, same results. */
#ifdef BDPT_PHASE_CLOCKS
__device__ unsigned long long bdpt_phase_clocks_dev[12];
#define BDPT_PHASE_BEGIN unsigned long long phase_t = clock64()
#define BDPT_PHASE_END(i) do { if (clk) { const unsigned long long now = clock64(); clk[i] += now - phase_t; phase_t = now; } } while (0)
#define BDPT_PHASE_RESUME phase_t = clock64()
#else
#define BDPT_PHASE_BEGIN
#define BDPT_PHASE_END(i)
#define BDPT_PHASE_RESUME
#endif

template <bool COUNT>
BDPT_DEV Hit closest_hit_lds2(const bdpt_dev_scene &S, clm_f3 o, clm_f3 d, int otype, int oidx,
                              unsigned long long *n_cull, unsigned long long *n_exact, unsigned long long *clk = nullptr)
{
    BDPT_PHASE_BEGIN;
    Best b;
    best_init(b);                                                         /* kernel0.cl:374, :384-385 */
    const int skip_s = otype == 0 ? oidx : -1;                            /* :390 */
    const int skip_p = otype == 1 ? oidx : -1;                            /* :415 */
    const int skip_t = otype == 2 ? oidx : -1;                            /* :436 */
    const LdsLayout L = lds_layout<true>(S);
    for (int i = 0; i < S.n_spheres; i++) {
        const float4 s = bdpt_lds[L.sph + i];
        if (i != skip_s) test_sphere(b, s, i, o, d);
    }
    for (int i = 0; i < S.n_planes; i++) {
        const float4 q0 = bdpt_lds[L.pln + i * BDPT_PLN_F4], q1 = bdpt_lds[L.pln + i * BDPT_PLN_F4 + 1];
        if (i != skip_p) test_plane(b, q0, q1, i, o, d);
    }
    BDPT_PHASE_END(1);
    /* (C + |o|_1) |d|_inf, doubled: the ray's part of the margin */
    const float ray_margin = 2.0f * (S.cull_c + fabsf(o.x) + fabsf(o.y) + fabsf(o.z)) * fmaxf(fmaxf(fabsf(d.x), fabsf(d.y)), fabsf(d.z));
    for (int base = 0; base < S.n_triangles; base += 32) {
        const int cnt = min(32, S.n_triangles - base);
        uint32_t mask = 0u;
        const uint32_t group = S.cull_group[base >> 5];
        const float4 *c = bdpt_lds + L.cull + (group & 0x7ffu);
        /* the plane part of a triangle's test: val (exact, :286), -1/val... and the approximate point on the plane */
        struct PlaneHit { float val, r, px, py, pz; };
        auto plane_part = [&](const float4 c0) {
            PlaneHit h;
            h.val = clm_dot3(d, clm_make3(c0.x, c0.y, c0.z));
            const float num = __builtin_fmaf(-o.z, c0.z, __builtin_fmaf(-o.y, c0.y, __builtin_fmaf(-o.x, c0.x, c0.w)));
            h.r = __builtin_amdgcn_rcpf(h.val);
            const float t = num * h.r;
            h.px = __builtin_fmaf(t, d.x, o.x);
            h.py = __builtin_fmaf(t, d.y, o.y);
            h.pz = __builtin_fmaf(t, d.z, o.z);
            return h;
        };
        /* the triangle part: is the point outside by more than the margin?  (val <= 0 for a triangle that is kept, so
         * |1/val| = -r; a front-facing ray, val > 0 (:288), makes -val negative) */
        auto kept = [&](const PlaneHit &h, const float4 c1, const float4 c2, const float sc) {
            const float b0 = __builtin_fmaf(h.pz, c1.z, __builtin_fmaf(h.py, c1.y, __builtin_fmaf(h.px, c1.x, c1.w)));
            const float b1 = __builtin_fmaf(h.pz, c2.z, __builtin_fmaf(h.py, c2.y, __builtin_fmaf(h.px, c2.x, c2.w)));
            const float b2 = (sc - b0) - b1;
            const float m = __builtin_fminf(__builtin_fminf(b0, b1), b2);
            const float z = __builtin_fminf(__builtin_fmaf(-ray_margin, h.r, m), -h.val);
            return !(z < 0.0f);
        };
        for (int k = (int)((group >> 11) & 0xfu); k > 0; k--, c += BDPT_CULL_SLAB_F4) {      /* stage 1: two opposite walls of a box */
            const float4 c0 = c[0], c1 = c[1], c2 = c[2], c3 = c[3], c4 = c[4];
            const float val = clm_dot3(d, clm_make3(c0.x, c0.y, c0.z));
            const bool second = val > 0.0f;                                   /* the ray runs against -n: only the second wall can be hit (:288) */
            const float num = __builtin_fmaf(-o.z, c0.z, __builtin_fmaf(-o.y, c0.y, __builtin_fmaf(-o.x, c0.x, second ? c3.y : c0.w)));
            const float r = __builtin_amdgcn_rcpf(val);                       /* second wall: numerator and val both negated, same quotient */
            const float t = num * r;
            const float px = __builtin_fmaf(t, d.x, o.x), py = __builtin_fmaf(t, d.y, o.y), pz = __builtin_fmaf(t, d.z, o.z);
            const float u = __builtin_fmaf(pz, c1.z, __builtin_fmaf(py, c1.y, __builtin_fmaf(px, c1.x, c1.w)));
            const float v = __builtin_fmaf(pz, c2.z, __builtin_fmaf(py, c2.y, __builtin_fmaf(px, c2.x, c2.w)));
            const float su = c3.x - u, sv = c3.x - v, w = su - v;
            const float m0 = __builtin_fminf(__builtin_fminf(u, v), w), m1 = __builtin_fminf(__builtin_fminf(su, sv), -w);
            const float z0 = __builtin_fmaf(ray_margin, __builtin_fabsf(r), m0);   /* |1/val| of the wall that is tested */
            const float z1 = __builtin_fmaf(ray_margin, __builtin_fabsf(r), m1);
            mask |= (!(z0 < 0.0f) ? __float_as_uint(second ? c4.z : c4.x) : 0u) | (!(z1 < 0.0f) ? __float_as_uint(second ? c4.w : c4.y) : 0u);
        }
        /* two records per trip: one address register, one counter update and one branch for both (1 % of a Cornell launch) */
#pragma unroll 2
        for (int k = (int)((group >> 16) & 0x1fu); k > 0; k--, c += BDPT_CULL_QUAD_F4) {     /* stage 1: the two halves of a parallelogram */
            const float4 c0 = c[0], c1 = c[1], c2 = c[2], c3 = c[3];
            const PlaneHit h = plane_part(c0);
            const float u = __builtin_fmaf(h.pz, c1.z, __builtin_fmaf(h.py, c1.y, __builtin_fmaf(h.px, c1.x, c1.w)));
            const float v = __builtin_fmaf(h.pz, c2.z, __builtin_fmaf(h.py, c2.y, __builtin_fmaf(h.px, c2.x, c2.w)));
            const float su = c3.x - u, sv = c3.x - v, w = su - v;            /* first: u, v, w; second: s - u, s - v, -w */
            const float m0 = __builtin_fminf(__builtin_fminf(u, v), w), m1 = __builtin_fminf(__builtin_fminf(su, sv), -w);
            const float z0 = __builtin_fminf(__builtin_fmaf(-ray_margin, h.r, m0), -h.val);
            const float z1 = __builtin_fminf(__builtin_fmaf(-ray_margin, h.r, m1), -h.val);
            mask |= (!(z0 < 0.0f) ? __float_as_uint(c3.y) : 0u) | (!(z1 < 0.0f) ? __float_as_uint(c3.z) : 0u);
        }
        for (int k = (int)((group >> 21) & 0x1fu); k > 0; k--, c += BDPT_CULL_PAIR_F4) {     /* ... two triangles in one plane */
            const float4 c0 = c[0], c1 = c[1], c2 = c[2], c3 = c[3], c4 = c[4], c5 = c[5];
            const PlaneHit h = plane_part(c0);
            const bool k0 = kept(h, c1, c2, c5.x), k1 = kept(h, c3, c4, c5.y);
            mask |= (k0 ? __float_as_uint(c5.z) : 0u) | (k1 ? __float_as_uint(c5.w) : 0u);
        }
        for (int k = (int)(group >> 26); k > 0; k--, c += BDPT_CULL_SINGLE_F4) {              /* ... and the others */
            const float4 c0 = c[0], c1 = c[1], c2 = c[2], c3 = c[3];
            const PlaneHit h = plane_part(c0);
            mask |= kept(h, c1, c2, c3.x) ? __float_as_uint(c3.y) : 0u;
        }
        if (COUNT) *n_cull += (unsigned long long)cnt;
        BDPT_PHASE_END(2);
        if (skip_t >= base && skip_t < base + 32) mask &= ~(1u << (skip_t - base));                   /* :436 */
        while (__builtin_amdgcn_ballot_w64(mask != 0u) != 0) {            /* stage 2 */
            const bool wanted = mask != 0u;
            const int i = base + (wanted ? __builtin_ctz(mask) : 0);      /* a lane with nothing left reads a record it ignores */
            mask &= mask - 1u;
            const int r = i * BDPT_TRI_F4;
            const float4 q0 = bdpt_lds[r], q1 = bdpt_lds[r + 1], q2 = bdpt_lds[r + 2], q3 = bdpt_lds[r + 3],
                         q4 = bdpt_lds[r + 4];
            const float e2z = bdpt_lds[r + 5].x;
            test_triangle_flat(b, q0, q1, q2, q3, q4, e2z, i, wanted, o, d);
            if (COUNT && wanted) ++*n_exact;
        }
        BDPT_PHASE_END(3);
    }
    const Hit found = finish_hit_lds<true>(S, b, d);
    BDPT_PHASE_END(4);
    return found;
}

/* ------------------------------------------------------------------------- */
/* Scenes of many triangles: the hierarchy of bdpt_bvh.h, walked by every lane for itself.
 *
 * Spheres and planes first, in the reference's order and with its strict comparison (they come before the triangles
 * in kernel0.cl:388-449, so they win ties against triangles).  Then the triangles, in whatever order the walk finds
 * them: a candidate replaces the best when its distance is less, or equal with a lower triangle index -- the outcome
 * of the reference's in-order walk with its strict "<" (:441).  A candidate is a triangle whose leaf the ray reaches;
 * it goes through the reference's own test, operation for operation.
 *
 * The walk: a node holds the boxes of its two children; a child is visited when the ray passes through its box
 * widened by q(o) (bdpt_bvh.h), in front of the origin and not beyond the best hit so far; the nearer child first,
 * the other one pushed on the lane's stack (one column of a [bvh_stack_limit][64] array of LDS per wavefront, so the
 * lanes' accesses never share a bank).  A ray whose numbers are not finite visits every node.                     */
/* ------------------------------------------------------------------------- */

/* best_consider for triangles met in any order: distance first (as the reference compares it: after the root), then index */
BDPT_DEV void best_consider_unordered(Best &b, clm_f3 p, clm_f3 o, int i)
{
    const clm_f3 v = clm_sub3(p, o);
    const float d2 = clm_dot3(v, v);
    bool take = d2 < b.d2_lo;
    const bool close_call = !take && d2 <= b.d2 * 1.000001f;   /* equal roots are at most two parts in 1e7 apart before the root */
    if (__builtin_amdgcn_ballot_w64(close_call) != 0) {
        asm volatile("");
        if (close_call) {
            const float s = clm_sqrt(d2), sb = clm_sqrt(b.d2);
            take = s < sb || (s == sb && b.type == 2 && i < b.idx);
        }
    }
    if (take) {
        b.d2 = d2;
        b.d2_lo = d2 > 1.0e-30f ? d2 * 0.999999f : 0.0f;
        b.type = 2;
        b.idx = i;
        b.pos = p;
    }
}

/* the reference's triangle test (kernel0.cl:273-312) on a record this lane loaded for itself; `i` goes through
 * best_consider_unordered */
BDPT_DEV void test_triangle_unordered(Best &b, const float4 *rec, int i, int skip, clm_f3 o, clm_f3 d)
{
    const float4 q0 = rec[0], q1 = rec[1], q2 = rec[2], q3 = rec[3], q4 = rec[4];
    const float e2z = rec[5].x;
    clm_f3 n = clm_make3(q0.x, q0.y, q0.z);
    float val = clm_dot3(d, n);                                                      /* :286 */
    clm_f3 a = clm_make3(q0.w, q1.x, q1.y);
    float t = clm_div(clm_dot3(clm_sub3(a, o), n), val);                             /* :291 */
    if (val > 0 || t < 0) return;                                                    /* :288, :292 */
    clm_f3 p = clm_mad3(t, d, o);                                                    /* :295 */
    clm_f3 e0 = clm_make3(q1.z, q1.w, q2.x);
    clm_f3 c = clm_make3(q2.y, q2.z, q2.w);
    clm_f3 e1 = clm_make3(q3.x, q3.y, q3.z);
    clm_f3 bb = clm_make3(q3.w, q4.x, q4.y);
    clm_f3 e2 = clm_make3(q4.z, q4.w, e2z);
    float dA = clm_dot3(clm_cross3(clm_sub3(p, a), e0), n);                          /* :298 */
    float dB = clm_dot3(clm_cross3(clm_sub3(p, c), e1), n);                          /* :299 */
    float dC = clm_dot3(clm_cross3(clm_sub3(p, bb), e2), n);                         /* :300 */
    if (dA < 0 || dB < 0 || dC < 0) return;                                          /* :304 */
    if (i != skip) best_consider_unordered(b, p, o, i);                              /* :436, :441-448 */
}

#ifndef BDPT_BVH_STEAL
#define BDPT_BVH_STEAL 1          /* the lanes of a wavefront share the work of their walks (closest_hit_bvh) */
#endif
#ifndef BDPT_BVH_STEAL_MIN
#define BDPT_BVH_STEAL_MIN 8      /* idle lanes it takes for a round of sharing */
#endif
/* bytes of LDS behind a wavefront's stacks when its lanes share work: 64 words of 8 bytes and one more */
#define BDPT_BVH_SHARE_BYTES (BDPT_BVH_STEAL ? 64 * 8 + 64 : 0)

/* the reference's triangle test (kernel0.cl:273-312) as test_triangle_unordered runs it, up to the accepted point:
 * true and distance(p, o) as the search compares it (:441) when the ray hits */
BDPT_DEV bool triangle_hit_distance(const float4 *rec, clm_f3 o, clm_f3 d, float &dist)
{
    const float4 q0 = rec[0], q1 = rec[1], q2 = rec[2], q3 = rec[3], q4 = rec[4];
    const float e2z = rec[5].x;
    clm_f3 n = clm_make3(q0.x, q0.y, q0.z);
    float val = clm_dot3(d, n);                                                      /* :286 */
    clm_f3 a = clm_make3(q0.w, q1.x, q1.y);
    float t = clm_div(clm_dot3(clm_sub3(a, o), n), val);                             /* :291 */
    if (val > 0 || t < 0) return false;                                              /* :288, :292 */
    clm_f3 p = clm_mad3(t, d, o);                                                    /* :295 */
    clm_f3 e0 = clm_make3(q1.z, q1.w, q2.x);
    clm_f3 c = clm_make3(q2.y, q2.z, q2.w);
    clm_f3 e1 = clm_make3(q3.x, q3.y, q3.z);
    clm_f3 bb = clm_make3(q3.w, q4.x, q4.y);
    clm_f3 e2 = clm_make3(q4.z, q4.w, e2z);
    float dA = clm_dot3(clm_cross3(clm_sub3(p, a), e0), n);                          /* :298 */
    float dB = clm_dot3(clm_cross3(clm_sub3(p, c), e1), n);                          /* :299 */
    float dC = clm_dot3(clm_cross3(clm_sub3(p, bb), e2), n);                         /* :300 */
    if (dA < 0 || dB < 0 || dC < 0) return false;                                    /* :304 */
    const clm_f3 v = clm_sub3(p, o);
    dist = clm_sqrt(clm_dot3(v, v));
    return true;
}

/* -DBDPT_BVH_STATS (tools/bvh_stats.py): how well the lanes of a wavefront keep together in the walk.  Sums over all
 * searches of a launch: node steps of the lanes / trips of the wavefronts through the node loop / leaves of the lanes /
 * trips through the leaf block / searches of the lanes / searches of the wavefronts. */
#ifdef BDPT_BVH_STATS
__device__ unsigned long long bvh_stats[8];
#define BDPT_BVH_STAT(k, n) atomicAdd(&bvh_stats[k], (unsigned long long)(n))
#define BDPT_BVH_FIRST_LANE (__builtin_amdgcn_mbcnt_hi((uint32_t)(__builtin_amdgcn_ballot_w64(true) >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)__builtin_amdgcn_ballot_w64(true), 0u)) == 0)
#endif

template <bool COUNT = false>
BDPT_DEV Hit closest_hit_bvh(const bdpt_dev_scene &S, uint32_t *stack, clm_f3 o, clm_f3 d, int otype, int oidx,
                             unsigned long long *n_cull = nullptr, unsigned long long *n_exact = nullptr)
{
    Best b;
    best_init(b);                                                         /* kernel0.cl:374, :384-385 */
    const int skip_s = otype == 0 ? oidx : -1;                            /* :390 */
    const int skip_p = otype == 1 ? oidx : -1;                            /* :415 */
    const int skip_t = otype == 2 ? oidx : -1;                            /* :436 */
    for (int i = 0; i < S.n_spheres; i++) {
        const float4 s = uload(S.sph, i);
        if (i != skip_s) test_sphere(b, s, i, o, d);
    }
    for (int i = 0; i < S.n_planes; i++) {
        const float4 q0 = uload(S.pln, (size_t)i * BDPT_PLN_F4 + 0), q1 = uload(S.pln, (size_t)i * BDPT_PLN_F4 + 1);
        if (i != skip_p) test_plane(b, q0, q1, i, o, d);
    }
    for (int k = 0; k < S.bvh_n_brute; k++) {                             /* the triangles outside the hierarchy */
        const int i = (int)S.bvh_brute[k];
        test_triangle_unordered(b, S.tri + (size_t)i * BDPT_TRI_F4, i, skip_t, o, d);
        if (COUNT) ++*n_exact;
    }
    if (S.bvh_root == BDPT_BVH_NONE) return finish_hit(S, b, d);

    /* what the box tests need of a ray (its own, or -- when lanes share work, below -- another lane's) */
    bool all;                                                             /* no box test for this ray (a NaN lands here too) */
    clm_f3 op, om, inv;                                                   /* origin plus and minus the boxes' widening for this ray; 1 / direction */
    float per_length;                                                     /* ray parameter per unit of distance, rounded up */
    auto ray_numbers = [&](clm_f3 ro, clm_f3 rd) {
        const float dd = clm_dot3(rd, rd), oo = clm_dot3(ro, ro);
        all = !(oo < 1.0e30f) || !(dd > 1.0e-30f && dd < 1.0e30f);
        const float q = 2.0e-5f * (1.0f + fabsf(ro.x) + fabsf(ro.y) + fabsf(ro.z));
        op = clm_make3(ro.x + q, ro.y + q, ro.z + q);
        om = clm_make3(ro.x - q, ro.y - q, ro.z - q);
        auto inverse = [](float x) { return 1.0f / (fabsf(x) < 1.0e-30f ? __builtin_copysignf(1.0e-30f, x) : x); };
        inv = clm_make3(inverse(rd.x), inverse(rd.y), inverse(rd.z));
        per_length = 1.0001f / __builtin_sqrtf(dd);
    };
    ray_numbers(o, d);
    float reach = __builtin_sqrtf(b.d2) * per_length;                     /* no hit beyond this parameter can win */
    /* the ray's line against one box: does it pass, and where does it enter */
    auto slab = [&](float lox, float loy, float loz, float hix, float hiy, float hiz, float &enter) {
        const float ax = (lox - op.x) * inv.x, bx = (hix - om.x) * inv.x;
        const float ay = (loy - op.y) * inv.y, by = (hiy - om.y) * inv.y;
        const float az = (loz - op.z) * inv.z, bz = (hiz - om.z) * inv.z;
        const float t0 = __builtin_fmaxf(__builtin_fmaxf(__builtin_fminf(ax, bx), __builtin_fminf(ay, by)), __builtin_fminf(az, bz));
        const float t1 = __builtin_fminf(__builtin_fminf(__builtin_fmaxf(ax, bx), __builtin_fmaxf(ay, by)), __builtin_fmaxf(az, bz));
        enter = t0;
        /* entry no later than exit, exit not behind the origin, entry not beyond the best hit: with the entry rounded
         * down and the exit up by 1e-5 of themselves.  (The box reaches q(o) beyond any point the triangle test can
         * accept, so an exit parameter that should be positive is not zero or less by rounding.) */
        const float in = __builtin_fmaxf(t0 * 0.99999f, 0.0f), out = __builtin_fminf(t1 * 1.00001f, reach);
        return all || !(in > out);
    };
    /* Two loops in turn, so that the lanes of a wavefront run the same code most of the time: the lanes walk down,
     * and a lane that reaches a leaf waits there; when BDPT_BVH_WAIT lanes wait (or no lane stands at a node) the
     * waiting lanes run the reference's test on their leaves' triangles together and the walk goes on.  100 000
     * triangles, 2 samples: waiting for all lanes 39 ms (a lane idles at its leaf through the others' node steps: 282
     * trips through the node loop per search where a lane needs 44), for any lane 27.2 (every trip pays for a triangle
     * test), for four 26.5 (145 trips).  Letting a lane collect two, four or eight leaves in a queue before the tests
     * costs more in every trip than it saves in trips. */
#ifdef BDPT_BVH_STATS
    unsigned int stat[4] = {0, 0, 0, 0};
#endif
    constexpr int kDone = BDPT_BVH_EMPTY;                                 /* no leaf reference has this value */
    /* One walk from the root with a stack of `capacity` entries behind store() / load(); true for a lane whose stack
     * would have run over (a ray that passes all four boxes of node after node needs three entries per level): it has
     * left the walk early and its best hit so far is a candidate like any other. */
    auto walk = [&](auto store, auto load, int capacity) {
        int cur = S.bvh_root, sp = 0;
        bool overflow = false;
        for (;;) {
            for (;;) {
                /* the node loop ends for the whole wavefront when no lane stands at a node, or when BDPT_BVH_WAIT lanes
                 * wait at leaves: they would otherwise idle through the others' node steps */
                const bool at_node = cur >= 0;
                if (__builtin_amdgcn_ballot_w64(at_node) == 0) break;
                if (__builtin_popcountll(__builtin_amdgcn_ballot_w64(cur < 0 && cur != kDone)) >= BDPT_BVH_WAIT_ALONE) break;
                if (!at_node) continue;
#ifdef BDPT_BVH_STATS
                stat[0]++; if (BDPT_BVH_FIRST_LANE) stat[1]++;
#endif
                const float4 *node = S.bvh_nodes + (size_t)cur * BDPT_BVH_NODE_F4;
#if BDPT_BVH_WIDE == 2
                const float4 n0 = node[0], n1 = node[1], n2 = node[2], n3 = node[3];
                float t0, t1;
                if (COUNT) *n_cull += 2;                                      /* box tests */
                const bool h0 = slab(n0.x, n0.y, n0.z, n0.w, n1.x, n1.y, t0);
                const bool h1 = slab(n1.z, n1.w, n2.x, n2.y, n2.z, n2.w, t1);
                const int c0 = __float_as_int(n3.x), c1 = __float_as_int(n3.y);
                if (h0 && h1) {
                    const bool first0 = !(t1 < t0);
                    store(sp, first0 ? c1 : c0); sp++;
                    cur = first0 ? c0 : c1;
                } else if (h0 || h1) {
                    cur = h0 ? c0 : c1;
                } else if (sp == 0) {
                    cur = kDone;
                } else {
                    { sp--; cur = load(sp); }
                }
#else
                const float4 lx = node[0], ly = node[1], lz = node[2], hx = node[3], hy = node[4], hz = node[5], rf = node[6];
                if (COUNT) *n_cull += 4;                                      /* box tests */
                float t0, t1, t2, t3;
                const bool h0 = slab(lx.x, ly.x, lz.x, hx.x, hy.x, hz.x, t0), h1 = slab(lx.y, ly.y, lz.y, hx.y, hy.y, hz.y, t1);
                const bool h2 = slab(lx.z, ly.z, lz.z, hx.z, hy.z, hz.z, t2), h3 = slab(lx.w, ly.w, lz.w, hx.w, hy.w, hz.w, t3);
                /* the children this ray visits, nearest first: (entry parameter, reference) pairs through a sorting
                 * network; a child that is not visited (an empty slot's box is inside out: never hit) sorts last */
                float k0 = h0 ? t0 : INFINITY, k1 = h1 ? t1 : INFINITY, k2 = h2 ? t2 : INFINITY, k3 = h3 ? t3 : INFINITY;
                int r0 = h0 ? __float_as_int(rf.x) : kDone, r1 = h1 ? __float_as_int(rf.y) : kDone, r2 = h2 ? __float_as_int(rf.z) : kDone,
                    r3 = h3 ? __float_as_int(rf.w) : kDone;
                auto order = [](float &ka, int &ra, float &kb, int &rb) {
                    const bool swap = kb < ka;
                    const float k = swap ? kb : ka; kb = swap ? ka : kb; ka = k;
                    const int r = swap ? rb : ra; rb = swap ? ra : rb; ra = r;
                };
                order(k0, r0, k1, r1); order(k2, r2, k3, r3); order(k0, r0, k2, r2); order(k1, r1, k3, r3); order(k1, r1, k2, r2);
                const int more = (r1 != kDone) + (r2 != kDone) + (r3 != kDone);
                if (sp + more > capacity) { overflow = true; cur = kDone; break; }
                if (r3 != kDone) { store(sp, r3); sp++; }
                if (r2 != kDone) { store(sp, r2); sp++; }
                if (r1 != kDone) { store(sp, r1); sp++; }
                if (r0 != kDone) {
                    cur = r0;
                } else if (sp == 0) {
                    cur = kDone;
                } else {
                    { sp--; cur = load(sp); }
                }
#endif
            }
            if (__builtin_amdgcn_ballot_w64(cur != kDone) == 0) break;
            if (cur < 0 && cur != kDone) {
#ifdef BDPT_BVH_STATS
                stat[2]++; if (BDPT_BVH_FIRST_LANE) stat[3]++;
#endif
                const uint32_t leaf = ~(uint32_t)cur;
                const uint32_t first = leaf & 0x0FFFFFFFu, count = (leaf >> 28) + 1u;
                for (uint32_t k = 0; k < count; k++) {
                    const int i = (int)S.bvh_leaf_tri[first + k];
                    test_triangle_unordered(b, S.tri + (size_t)i * BDPT_TRI_F4, i, skip_t, o, d);
                }
                if (COUNT) *n_exact += count;
                reach = __builtin_sqrtf(b.d2) * per_length;
                if (sp == 0) cur = kDone;
                else { sp--; cur = load(sp); }
            }
        }
        return overflow;
    };
#if BDPT_BVH_STEAL
    /*
     * The first walk with the lanes of the wavefront sharing work.  The rays of a wavefront are unrelated and a search
     * lasts as long as its longest walk, with most lanes idle for most of it; so a lane whose walk has ended takes a
     * subtree off the bottom of another lane's stack -- the farthest, largest one that lane has waiting -- and walks it
     * for that lane's ray.  What a lane finds for a ray, its own included, goes to the ray owner's word of LDS: the
     * least (distance, index) of all hits, kept by a 64-bit atomic minimum of the distance's bits over the triangle's
     * index (distances are not negative, so their bits order as they do).  That word also bounds the walks of
     * everybody working for the ray.  When no lane has a node, a leaf or anything to give, every owner runs the
     * reference's test once more on its word's triangle: the comparison with the spheres, planes and listed triangles
     * met before, and the hit point.  Lanes that were not part of the search when it began take no part.
     */
    auto walk_shared = [&]() {
        const int lane = threadIdx.x & 63;
        uint32_t *const wave_lds = stack - lane;
        unsigned long long *const found = (unsigned long long *)(wave_lds + S.bvh_stack_limit * 64);    /* [64] best (distance, index) per owner */
        unsigned long long *const ran_over = found + 64;                                               /* one bit per owner */
        constexpr unsigned long long kNothing = ~0ull;
        found[lane] = kNothing;
        __hip_atomic_fetch_and(ran_over, ~(1ull << lane), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        int owner = lane;                                                 /* the lane whose ray this lane works for; op, om, inv, all, reach are that ray's */
        int cur = S.bvh_root, sp = 0, lo = 0;                             /* this lane's stack holds [lo, sp) */
        auto pop = [&]() {
            if (sp == lo) cur = kDone;
            else { sp--; cur = (int)stack[sp * 64]; }
        };
        for (;;) {
            for (;;) {
                const bool at_node = cur >= 0;
                if (__builtin_amdgcn_ballot_w64(at_node) == 0) break;
                if (__builtin_popcountll(__builtin_amdgcn_ballot_w64(cur < 0 && cur != kDone)) >= BDPT_BVH_WAIT) break;
                if (!at_node) continue;
#ifdef BDPT_BVH_STATS
                stat[0]++; if (BDPT_BVH_FIRST_LANE) stat[1]++;
#endif
                const float4 *node = S.bvh_nodes + (size_t)cur * BDPT_BVH_NODE_F4;
                const float4 lx = node[0], ly = node[1], lz = node[2], hx = node[3], hy = node[4], hz = node[5], rf = node[6];
                if (COUNT) *n_cull += 4;
                float t0, t1, t2, t3;
                const bool h0 = slab(lx.x, ly.x, lz.x, hx.x, hy.x, hz.x, t0), h1 = slab(lx.y, ly.y, lz.y, hx.y, hy.y, hz.y, t1);
                const bool h2 = slab(lx.z, ly.z, lz.z, hx.z, hy.z, hz.z, t2), h3 = slab(lx.w, ly.w, lz.w, hx.w, hy.w, hz.w, t3);
                float k0 = h0 ? t0 : INFINITY, k1 = h1 ? t1 : INFINITY, k2 = h2 ? t2 : INFINITY, k3 = h3 ? t3 : INFINITY;
                int r0 = h0 ? __float_as_int(rf.x) : kDone, r1 = h1 ? __float_as_int(rf.y) : kDone, r2 = h2 ? __float_as_int(rf.z) : kDone,
                    r3 = h3 ? __float_as_int(rf.w) : kDone;
                auto order = [](float &ka, int &ra, float &kb, int &rb) {
                    const bool swap = kb < ka;
                    const float k = swap ? kb : ka; kb = swap ? ka : kb; ka = k;
                    const int r = swap ? rb : ra; rb = swap ? ra : rb; ra = r;
                };
                order(k0, r0, k1, r1); order(k2, r2, k3, r3); order(k0, r0, k2, r2); order(k1, r1, k3, r3); order(k1, r1, k2, r2);
                const int more = (r1 != kDone) + (r2 != kDone) + (r3 != kDone);
                if (sp + more > S.bvh_stack_limit) {                      /* the owner walks again by itself, afterwards */
                    __hip_atomic_fetch_or(ran_over, 1ull << owner, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                    sp = lo;
                    cur = kDone;
                    continue;
                }
                if (r3 != kDone) { stack[sp * 64] = (uint32_t)r3; sp++; }
                if (r2 != kDone) { stack[sp * 64] = (uint32_t)r2; sp++; }
                if (r1 != kDone) { stack[sp * 64] = (uint32_t)r1; sp++; }
                if (r0 != kDone) cur = r0;
                else pop();
            }
            /* the triangle tests want the ray itself: fetched from its owner's registers by every lane, here where all
             * lanes of the search still run together (a lane that is switched off has nothing to give) */
            auto own = [&](float v) { return __uint_as_float((uint32_t)__builtin_amdgcn_ds_bpermute(owner << 2, (int)__float_as_uint(v))); };
            const clm_f3 jo = clm_make3(own(o.x), own(o.y), own(o.z)), jd = clm_make3(own(d.x), own(d.y), own(d.z));
            const int jskip = __builtin_amdgcn_ds_bpermute(owner << 2, skip_t);
            if (cur < 0 && cur != kDone) {
#ifdef BDPT_BVH_STATS
                stat[2]++; if (BDPT_BVH_FIRST_LANE) stat[3]++;
#endif
                const uint32_t leaf = ~(uint32_t)cur;
                const uint32_t first = leaf & 0x0FFFFFFFu, count = (leaf >> 28) + 1u;
                for (uint32_t k = 0; k < count; k++) {
                    const int i = (int)S.bvh_leaf_tri[first + k];
                    float dist;
                    if (triangle_hit_distance(S.tri + (size_t)i * BDPT_TRI_F4, jo, jd, dist) && i != jskip)
                        __hip_atomic_fetch_min(&found[owner], ((unsigned long long)__float_as_uint(dist) << 32) | (uint32_t)i,
                                               __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                }
                if (COUNT) *n_exact += count;
                /* whatever anybody has found for this ray bounds the rest of this walk */
                reach = __builtin_fminf(reach, __uint_as_float((uint32_t)(found[owner] >> 32) & 0x7FFFFFFFu) * per_length);
                pop();
            }
            /* ---- lanes without work take some from lanes that have it waiting ---- */
            const bool idle = cur == kDone, giving = !idle && sp - lo >= 1;
            const unsigned long long idle_m = __builtin_amdgcn_ballot_w64(idle), giving_m = __builtin_amdgcn_ballot_w64(giving);
            if (__builtin_amdgcn_ballot_w64(!idle) == 0) break;           /* nobody has anything left */
            if (giving_m != 0 && __builtin_popcountll(idle_m) >= BDPT_BVH_STEAL_MIN) {
                const int pairs = min(__builtin_popcountll(idle_m), __builtin_popcountll(giving_m));
                auto rank_in = [&](unsigned long long m) {
                    return (int)__builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u));
                };
                const bool taking = idle && rank_in(idle_m) < pairs;
                const bool robbed = giving && rank_in(giving_m) < pairs;
                /* the lane this lane takes from: the giving lane of its own rank */
                int from = 0, r = rank_in(idle_m);
#pragma unroll
                for (int w = 32; w >= 1; w >>= 1) {
                    const int c = __builtin_popcountll((giving_m >> from) & ((1ull << w) - 1ull));
                    const bool up = r >= c;
                    r -= up ? c : 0;
                    from += up ? w : 0;
                }
                from = taking ? from : lane;
                auto pull = [&](float v) { return __uint_as_float((uint32_t)__builtin_amdgcn_ds_bpermute(from << 2, (int)__float_as_uint(v))); };
                auto pull_i = [&](int v) { return __builtin_amdgcn_ds_bpermute(from << 2, v); };
                const clm_f3 nop = clm_make3(pull(op.x), pull(op.y), pull(op.z)), nom = clm_make3(pull(om.x), pull(om.y), pull(om.z)),
                             ninv = clm_make3(pull(inv.x), pull(inv.y), pull(inv.z));
                const float nreach = pull(reach), nper = pull(per_length);
                const int nowner = pull_i(owner), nlo = pull_i(lo), nall = pull_i(all ? 1 : 0);
                if (taking) {
                    op = nop; om = nom; inv = ninv; reach = nreach; per_length = nper; all = nall != 0; owner = nowner;
                    cur = (int)wave_lds[nlo * 64 + from];
                    sp = lo = 0;
                }
                if (robbed) lo++;
            }
        }
        /* every owner: its ray's best triangle against what it had before, and whether somebody's stack ran over */
        ray_numbers(o, d);
        const unsigned long long mine = found[lane];
        if (mine != kNothing) {
            const int i = (int)(uint32_t)mine;
            test_triangle_unordered(b, S.tri + (size_t)i * BDPT_TRI_F4, i, skip_t, o, d);
        }
        reach = __builtin_sqrtf(b.d2) * per_length;
        return ((*ran_over >> lane) & 1ull) != 0ull;
    };
#endif
    /* First with the stack in LDS (bvh_stack_limit entries: no more than the tree can need, no more than
     * BDPT_BVH_STACK).  The lanes that ran over walk again, from the root, with BDPT_BVH_DEEP entries in their private
     * memory (rays that graze a fine mesh do; on a soup of 100 000 triangles none does), and whoever runs over there
     * too meets every triangle of the scene.  The comparison by (distance, index) makes a second meeting with a triangle
     * harmless. */
#if BDPT_BVH_STEAL
    /* sharing pays from some 8 000 triangles on (soups of 3 000 / 10 000 / 30 000 / 100 000 triangles, 16 samples at 1080p:
     * 40.3 / 58.9 / 86 / 132 ms shared, 38.2 / 59.9 / 104 / 154 ms each lane for itself): the host decides */
    bool overflow = S.bvh_share ? walk_shared()
                                : walk([&](int at, int ref) { stack[at * 64] = (uint32_t)ref; }, [&](int at) { return (int)stack[at * 64]; }, S.bvh_stack_limit);
#else
    bool overflow = walk([&](int at, int ref) { stack[at * 64] = (uint32_t)ref; }, [&](int at) { return (int)stack[at * 64]; }, S.bvh_stack_limit);
#endif
    if (__builtin_amdgcn_ballot_w64(overflow) != 0) {
        asm volatile("");                                                  /* a real branch: the second walk is rare */
        if (overflow) {
            uint32_t deep[BDPT_BVH_DEEP];
            overflow = walk([&](int at, int ref) { deep[at] = (uint32_t)ref; }, [&](int at) { return (int)deep[at]; }, S.bvh_stack_total);
        }
    }
    if (__builtin_amdgcn_ballot_w64(overflow) != 0) {
        for (int i = 0; i < S.n_triangles; i++) {
            const size_t r = (size_t)i * BDPT_TRI_F4;
            const float4 q0 = uload(S.tri, r), q1 = uload(S.tri, r + 1), q2 = uload(S.tri, r + 2), q3 = uload(S.tri, r + 3), q4 = uload(S.tri, r + 4);
            const float4 rec[6] = {q0, q1, q2, q3, q4, uload(S.tri, r + 5)};
            if (overflow) test_triangle_unordered(b, rec, i, skip_t, o, d);
        }
    }
#ifdef BDPT_BVH_STATS
    for (int k = 0; k < 4; k++) BDPT_BVH_STAT(k, stat[k]);
    BDPT_BVH_STAT(4, 1);
    if (BDPT_BVH_FIRST_LANE) BDPT_BVH_STAT(5, 1);
#endif
    return finish_hit(S, b, d);
}

enum { MODE_DIRECT = 0, MODE_LDS = 2, MODE_LDS2 = 3, MODE_BVH = 4 };

/* per-lane work counters of a counting launch */
struct Tally {
    unsigned long long rays, full, early, cull, exact;
#ifdef BDPT_PHASE_CLOCKS
    unsigned long long clk[12];
#endif
};

/* row `i` of the surface table (2 float4 per surface: colour, emissive) */
template <int MODE>
BDPT_DEV float4 surface_row(const bdpt_dev_scene &S, uint32_t i)
{
    if (MODE == MODE_LDS) return bdpt_lds[lds_layout<false>(S).surf + (int)i];
    if (MODE == MODE_LDS2) return bdpt_lds[lds_layout<true>(S).surf + (int)i];
    return S.surf[i];
}

/*
 * The eye vertex of a sample is needed once per light-subpath vertex (connection direction, connection test) and
 * at the end (its colour), with closest-hit searches in between that want every register: the LDS kernels keep
 * it in three float4 of LDS per lane, behind the scene (48-byte stride: 16 lanes cover the 64 banks exactly).
 * Volatile accesses: each read is a read, not a value the compiler may keep in a register across the searches.
 */
#define BDPT_PARK_F4 3
template <int MODE>
BDPT_DEV volatile float4 *lane_park(const bdpt_dev_scene &S)
{
    const LdsLayout L = MODE == MODE_LDS2 ? lds_layout<true>(S) : lds_layout<false>(S);
    return bdpt_lds + L.ids + L.n_ids / 4 + (int)threadIdx.x * BDPT_PARK_F4;
}
BDPT_DEV void park_store(volatile float4 *p, float x, float y, float z, float w)
{
    p->x = x; p->y = y; p->z = z; p->w = w;
}
BDPT_DEV float4 park_load(const volatile float4 *p)
{
    return make_float4(p->x, p->y, p->z, p->w);
}

/* SCATTERED: the rays of a wavefront point in unrelated directions (every ray but the eye ray) */
template <int MODE, bool SCATTERED, bool COUNT>
BDPT_DEV Hit trace_one(const bdpt_dev_scene &S, clm_f3 o, clm_f3 d, int otype, int oidx, Tally &T)
{
#ifdef BDPT_PHASE_CLOCKS
    if (MODE == MODE_LDS2) return closest_hit_lds2<COUNT>(S, o, d, otype, oidx, &T.cull, &T.exact, T.clk);
#endif
    if (MODE == MODE_LDS2) return closest_hit_lds2<COUNT>(S, o, d, otype, oidx, &T.cull, &T.exact);
    if (MODE == MODE_BVH)                                   /* the lane's column of its wavefront's stack, at the start of LDS */
        return closest_hit_bvh<COUNT>(S, (uint32_t *)bdpt_lds + (threadIdx.x >> 6) * (S.bvh_stack_limit * 64 + BDPT_BVH_SHARE_BYTES / 4) + (threadIdx.x & 63), o, d, otype, oidx, &T.cull, &T.exact);
    return MODE == MODE_LDS ? closest_hit_lds<SCATTERED>(S, o, d, otype, oidx) : closest_hit_direct(S, o, d, otype, oidx);
}

/* One sample (kernel0.cl:592-847).  Returns true when the reference would have
 * returned early (eye ray missed or hit a light, :645-656); col is then the raw
 * value it writes. */
/* ALT: bounce directions from one of the samplers the reference holds but does not call (K.sampler).
 * ALL_RAYS: also trace the rays whose result nothing but the ray counters and the hit-id chain can
 * see (the bounce ray of the last vertex, kernel0.cl:793 at lti = maxL-1). */
template <bool AOV, int MODE, bool ALT, bool ALL_RAYS, bool COUNT>
BDPT_DEV bool sample_direct(const bdpt_kargs &K, float xt, float yt, uint32_t &w, uint32_t &draws, uint32_t &rays,
                            uint32_t &hash, int eye_id[2], float col[4], Tally &T)
{
    const bdpt_dev_scene &S = K.scene;
#ifdef BDPT_PHASE_CLOCKS
    unsigned long long *const clk = T.clk;
#endif
    BDPT_PHASE_BEGIN;
    rng_float(w, draws);                                                  /* :592, discarded */
    clm_f3 dir = camera_ray(w, draws, K, xt, yt);
    clm_f3 eye_pos = clm_mad3(0.00001f, dir, f3(K.cam_pos));              /* :632 */
    BDPT_PHASE_END(5);
    Hit eye = trace_one<MODE, false, COUNT>(S, eye_pos, dir, 4, 0, T);        /* :634 */
    BDPT_PHASE_RESUME;
    rays++;
    if (AOV) { hash = hash_step(hash, hit_id(eye.type, eye.idx)); eye_id[0] = eye.type; eye_id[1] = eye.idx; }

    if (eye.type < 0) {                                                   /* :652-656 */
        col[0] = col[1] = col[2] = col[3] = 0.0f;
        return true;
    }
    const float4 eye_diffuse = surface_row<MODE>(S, eye.surf * 2);
    const float4 eye_emissive = surface_row<MODE>(S, eye.surf * 2 + 1);
    if (eye_emissive.w != 0.0f) {                                         /* :645-650 */
        col[0] = eye_diffuse.x; col[1] = eye_diffuse.y; col[2] = eye_diffuse.z; col[3] = eye_diffuse.w;
        return true;
    }

    const bool parked = MODE == MODE_LDS || MODE == MODE_LDS2;
    volatile float4 *park = parked ? lane_park<MODE>(S) : nullptr;
    if (parked) {
        park_store(park + 0, eye.pos.x, eye.pos.y, eye.pos.z, __int_as_float(eye.type));
        park_store(park + 1, eye.n.x, eye.n.y, eye.n.z, __int_as_float(eye.idx));
        park_store(park + 2, eye_diffuse.x, eye_diffuse.y, eye_diffuse.z, eye_diffuse.w);
    }

    clm_f3 vpos, vn;                                                      /* current light-subpath vertex */
    BDPT_PHASE_END(5);
    sample_light(w, draws, K, vpos, vn);                                  /* :669 */
    BDPT_PHASE_END(6);
    int vtype = K.light_origin_type, vidx = K.light_origin_index;

    float full[4] = {0.0f, 0.0f, 0.0f, 0.0f};                             /* :689 */
    float acc[4] = {1.0f, 1.0f, 1.0f, 1.0f};                              /* :692 */
    float pre = K.pre_factor0;                                            /* :694 */
    int taken = 0;
    clm_f3 incoming = clm_make3(0.0f, 0.0f, 0.0f);                        /* :659 */
    for (int lti = 0; lti < K.max_l; lti++) {                             /* :696 */
        if (ALL_RAYS || !K.indirect_only || lti != 0) {                   /* INDIRECT_ONLY drops the light's own connection (:752-755) */
            clm_f3 to = eye.pos;
            if (parked) { const float4 e0 = park_load(park + 0); to = clm_make3(e0.x, e0.y, e0.z); }
            clm_f3 tdir = clm_normalize3(clm_sub3(to, vpos));             /* :699 */
            BDPT_PHASE_END(7);
            Hit vis = trace_one<MODE, true, COUNT>(S, vpos, tdir, vtype, vidx, T);  /* :705 */
            BDPT_PHASE_RESUME;
            rays++;
            if (AOV) hash = hash_step(hash, hit_id(vis.type, vis.idx));
            float contrib[4];
            Hit at = eye;                                                 /* what the connection term reads: type, idx, pos, n */
            if (parked) {
                const float4 e0 = park_load(park + 0), e1 = park_load(park + 1);
                at.pos = clm_make3(e0.x, e0.y, e0.z); at.type = __float_as_int(e0.w);
                at.n = clm_make3(e1.x, e1.y, e1.z); at.idx = __float_as_int(e1.w);
            }
            connection(K, at, vis, vpos, vn, tdir, acc, inv_pre_factor(K, lti, pre), contrib);
            if (!K.indirect_only || lti != 0) {                           /* :752-755 */
#pragma unroll
                for (int k = 0; k < 4; k++) full[k] += contrib[k];
            }
        }
        pre *= K.p_con;                                                   /* :745 */
        taken++;                                                          /* :757 */
        BDPT_PHASE_END(7);

        clm_f3 lpos = clm_mad3(0.00001f, vn, vpos);                       /* :761 */
        /* The bounce direction (:767) is used only if roulette (:787) lets the walk go on, and it
         * consumes a fixed number of draws: step over them, draw the fate, and build the direction
         * from the stepped-over numbers only when it is needed. */
        uint32_t w_dir = w, draws_dir = 0;
        const bool four = ALT && sampler_draws_four(K.sampler);
        w *= four ? lcg_pow(4) : lcg_pow(3);
        draws += four ? 4u : 3u;
        float fate = rng_float(w, draws);                                 /* :787 */
        if (fate > K.p_con) break;                                        /* :788 */
        if (!ALL_RAYS && lti + 1 == K.max_l) break;                       /* nothing reads what the last bounce ray finds */
        clm_f3 ldir = ALT ? sample_bounce_alt(K, w_dir, draws_dir, vn, incoming, lpos)
                          : sample_cone(w_dir, draws_dir, vn, 3.1415f * 0.50f);   /* :767, :486 */

        BDPT_PHASE_END(8);
        Hit nxt = trace_one<MODE, true, COUNT>(S, lpos, ldir, vtype, vidx, T);   /* :793 */
        BDPT_PHASE_RESUME;
        rays++;
        if (AOV) hash = hash_step(hash, hit_id(nxt.type, nxt.idx));
        if (nxt.type < 0) break;                                          /* :812 */
        const float4 nd = surface_row<MODE>(S, nxt.surf * 2);
        const float4 ne = surface_row<MODE>(S, nxt.surf * 2 + 1);
        if (ne.w != 0.0f) break;                                          /* :797-803 */
        acc[0] *= nd.x; acc[1] *= nd.y; acc[2] *= nd.z; acc[3] *= nd.w;   /* :827 with :822, :824 */
        if (ALT) incoming = ldir;                                         /* :831 */
        vpos = nxt.pos; vn = nxt.n; vtype = nxt.type; vidx = nxt.idx;     /* :832 */
        BDPT_PHASE_END(9);
    }
    BDPT_PHASE_RESUME;

    const float inv_taken = clm_div(1.0f, (float)taken);                  /* :846 */
    const float4 ed4 = parked ? park_load(park + 2) : eye_diffuse;
    const float ed[4] = {ed4.x, ed4.y, ed4.z, ed4.w};
#pragma unroll
    for (int k = 0; k < 4; k++) {
        float c = K.light_emissive[k] * full[k] * inv_taken;
        if (!K.indirect_only) c = c * ed[k];                              /* :844 / :846 */
        col[k] = c;
    }
    BDPT_PHASE_END(9);
    return false;
}

/* all samples of one run of one pixel (the whole launch when it is not sliced) */
template <bool AOV, bool COUNT, int MODE, bool ALT>
BDPT_DEV void direct_pixel(const bdpt_kargs &K, const Slice &sl, int x, int y);

/* Six wavefronts per SIMD (80 VGPRs, a handful of spills outside the closest-hit loops) measure 4 % faster than
 * the five the register allocator would settle for; seven and eight spill too much. */
template <bool AOV, bool COUNT, int MODE, bool ALT>
#ifndef BDPT_LDS2_WAVES
#define BDPT_LDS2_WAVES 6
#endif
#ifndef BDPT_BVH_WAVES
#define BDPT_BVH_WAVES 6
#endif
__global__ __launch_bounds__(BDPT_BLOCK_MAX) __attribute__((amdgpu_waves_per_eu(MODE == MODE_LDS2 ? BDPT_LDS2_WAVES : MODE == MODE_BVH ? BDPT_BVH_WAVES : 6, MODE == MODE_LDS2 ? BDPT_LDS2_WAVES : MODE == MODE_BVH ? BDPT_BVH_WAVES : 6)))
void bdpt_direct_kernel(const bdpt_kargs K)
{
    if (MODE == MODE_LDS) lds_fill_scene<false>(K.scene);   /* every thread of the workgroup, before any early exit */
    if (MODE == MODE_LDS2) lds_fill_scene<true>(K.scene);
    const Slice sl = slice_begin(K);
    if (sl.abandoned) return;                               /* the whole workgroup: nothing read, nothing written */
    int x, y;
    lane_pixel(K, sl.tile, x, y);
    if (x < K.rx0 + K.rw && y < K.ry0 + K.rh)
        direct_pixel<AOV, COUNT, MODE, ALT>(K, sl, x, y);
    slice_end(K, sl);
}

template <bool AOV, bool COUNT, int MODE, bool ALT>
BDPT_DEV void direct_pixel(const bdpt_kargs &K, const Slice &sl, int x, int y)
{
    const size_t pix = (size_t)y * K.width + x;

    float4 a4 = K.accum[pix];
    float acc[4] = {a4.x, a4.y, a4.z, a4.w};
    const uint32_t seed = K.seed_w[pix];
    uint32_t w = seed;
    SampleCursor cur;
    if (sl.s0 == 0) cursor_init(cur, K, seed);
    else cursor_load(cur, K, pix);

    uint32_t draws = 0, rays = 0, hash = 0;
    if (AOV) { draws = K.aov_rng_draws[pix]; rays = K.aov_rays[pix]; hash = K.aov_hit_hash[pix]; }
    Tally T = {};
#ifdef BDPT_PHASE_CLOCKS
    const unsigned long long pixel_t0 = clock64();
#endif
    int eye_id[2] = {-1, -1};
    float col[4] = {0.0f, 0.0f, 0.0f, 0.0f};

    float xt, yt;
    pixel_coord(K, x, y, xt, yt);

    if (K.reorder == 2 && sl.s0 == 0) {
        /* ADAPTIVE order: does the eye ray of the launch's first sample index reach the light loop? */
        uint32_t wp = cur.st_head, dp = 0;
        rng_float(wp, dp);                                                /* :592 */
        const clm_f3 dir = camera_ray(wp, dp, K, xt, yt);
        const Hit e = trace_one<MODE, false, COUNT>(K.scene, clm_mad3(0.00001f, dir, f3(K.cam_pos)), dir, 4, 0, T);
        const bool early = e.type < 0 || surface_row<MODE>(K.scene, e.surf * 2 + 1).w != 0.0f;   /* :645-656 */
        /* a select, not an assignment under `if (early)`: hipcc 7.2 has compiled the latter, in builds of the tiled
         * kernel that differed from this one only in how pixel state is stored, to code that sets cls for the lanes
         * whose ray leaves the scene and forgets the lanes that hit a light (the same few pixels on the light's
         * silhouette then took their samples in class order; tests/test_gpu_parity.py::test_adaptive_order_decides_per_pixel) */
        cur.cls = early ? -1 : cur.cls;
        if (COUNT && !AOV) T.rays++;                 /* a ray this launch form traces on top of the samples' own */
    }

    uint32_t start;
#ifdef BDPT_PHASE_CLOCKS
    unsigned long long *const clk = T.clk;
    T.clk[11] = clock64() - pixel_t0;
#endif
    BDPT_PHASE_BEGIN;
    if (K.class_masks && sl.s0 == 0 && cur.cls >= 1) {
        if (K.max_l <= 5) cursor_fill_masks<4>(K, sl.shared, pix, seed);
        else cursor_fill_masks<6>(K, sl.shared, pix, seed);
    }
    for (int s = sl.s0; s < sl.s1 && sample_next(cur, K, pix, start); s++) {   /* s counts the samples taken so far */
        BDPT_PHASE_END(10);
        const int n = K.num_samples + s;                                  /* kernel0.cl:588 */
        if (K.rng_mode == BDPT_RNG_STRIDED) w = start;
        const uint32_t w_before = w;
        const uint32_t rays_before = rays;
        const bool early = sample_direct<AOV, MODE, ALT, AOV, COUNT>(K, xt, yt, w, draws, rays, hash, eye_id, col, T);
        BDPT_PHASE_RESUME;
        if (COUNT) { T.rays += rays - rays_before; if (early) T.early++; else T.full++; }

        if (early && K.rng_mode == BDPT_RNG_STREAM) {
            /* kernel0.cl:645-656: raw value written, history dropped, the seed is not
             * written back -- so every later sample of this launch repeats this one
             * exactly.  Account for the repeats and stop. */
            w = w_before;
            acc[0] = col[0]; acc[1] = col[1]; acc[2] = col[2]; acc[3] = col[3];
            const int rest = sl.s1 - 1 - s;
            if (AOV) {
                const uint32_t id = hit_id(eye_id[0], eye_id[1]);
                for (int r = 0; r < rest; r++) hash = hash_step(hash, id);
                draws += 3u * (uint32_t)rest;
                rays += (uint32_t)rest;
            }
            if (COUNT) T.early += (unsigned long long)rest;
            break;
        }
        if (n == 0) {                                                     /* :855 */
#pragma unroll
            for (int k = 0; k < 4; k++) acc[k] = col[k];
        } else if (K.accum_mode == BDPT_ACCUM_SUM) {
#pragma unroll
            for (int k = 0; k < 4; k++) acc[k] = acc[k] + col[k];
        } else {                                                          /* :854 */
            const float s1 = clm_div((float)n, (float)(n + 1));
            const float s2 = clm_div(1.0f, (float)(n + 1));
#pragma unroll
            for (int k = 0; k < 4; k++) acc[k] = clm_fma(col[k], s2, acc[k] * s1);
        }
    }

    pixel_state_store<AOV>(K, sl, pix, acc, w, cur, eye_id, hash, col, draws, rays);
#ifdef BDPT_PHASE_CLOCKS
    T.clk[0] = clock64() - pixel_t0;
    if (__builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u)) == 0)    /* the wavefront's first active lane */
        for (int i = 0; i < 12; i++) atomicAdd(&bdpt_phase_clocks_dev[i], T.clk[i]);
#endif
    if (COUNT) {
        atomicAdd(&K.counters[0], T.rays);
        atomicAdd(&K.counters[1], T.full);
        atomicAdd(&K.counters[2], T.early);
        atomicAdd(&K.counters[3], T.cull);
        atomicAdd(&K.counters[4], T.exact);
    }
}

/* ------------------------------------------------------------------------- */
/* Tiled variant, for scenes too large to keep in LDS.  Each wavefront stages
 * the triangle records through its own LDS buffer in tiles of BDPT_TILE_TRIS:
 * all 64 lanes copy the tile with coalesced float4 loads (lane l copies float4
 * l, l+64, ... of the tile), then every lane walks the tile with wave-uniform
 * ds_read_b128 broadcasts, as the LDS-resident variant walks the whole scene.
 *
 * What is staged is not the triangle records but one bounding sphere per
 * triangle (bdpt_layout.h).  A ray whose line passes the sphere by more than
 * rounding can account for cannot produce a hit point inside the triangle, so
 * for that ray the reference's test would return "no hit" and nothing needs to
 * be computed; only when some lane's ray comes close is the full record read
 * (wave-uniform scalar loads) and the reference's test run, for those lanes.
 * Every ray still meets every triangle, in index order, and every candidate
 * that could be accepted goes through the same arithmetic as before.
 *
 * The margin: an accepted hit point p = o + t*d lies within a few ulp of the
 * coordinates of the ray line and, by the three edge tests, within a few ulp of
 * the triangle, hence within the sphere of radius R up to such terms.  The test
 * rejects when  (1 - 1e-5) |v|^2 - (v.d)^2  >  1.05 R^2 + 1e-5 (|o|^2 + 1),  v =
 * centre - o: the right-hand slack is some hundred times the rounding of the
 * left-hand side (relative 1e-7 of |v|^2) and of p (relative 1e-7 of |o| + |v|),
 * and a NaN anywhere fails the comparison, i.e. keeps the triangle.  The left-hand side is
 * evaluated expanded, |v|^2 = |c|^2 - 2 c.o + |o|^2 and v.d = c.d - o.d, with everything that depends on
 * the sphere alone folded into one stored number at upload and everything that depends on the ray alone
 * into one bound per query: seven fused operations per sphere (c.d - o.d and w - 2(1 - 1e-5) c.o as one
 * chain of three each, then one more) and one compare per eight spheres, instead of thirteen.  The expansion cancels, so
 * it is good to about 1.2e-6 (|c|^2 + |o|^2) only; both sides of the comparison give 2e-6 of their share
 * of that away (bdpt_api.cpp for |c|^2, `bound` below for |o|^2).  (v.d)^2 is the squared
 * length of v along the ray only for |d| = 1: with |d|^2 = 1 - e the left-hand side is too large by
 * e (v.d)^2 <= e |v|^2, which the factor 1 - 1e-5 absorbs for e <= 4e-6; rays further from unit length
 * are not culled at all.  Slivers
 * whose computed normal is unreliable carry an infinite radius (bdpt_api.cpp).
 * Two buffers per wavefront: the global loads of tile k+1 are issued
 * before tile k is walked and written to the other buffer afterwards.
 *
 * The copy needs every lane of the wavefront, so the sample is written with a
 * per-lane `alive` predicate instead of early exits: all lanes make the same
 * sequence of closest-hit calls and a lane whose path has ended is masked.
 * Buffers are private to a wavefront, so no workgroup barrier is involved and
 * wavefronts of a workgroup run independently.                                 */
/* ------------------------------------------------------------------------- */

#define BDPT_TILE_TRIS 192                               /* bounding spheres per tile: 3 float4 per lane */
#define BDPT_TILE_F4 BDPT_TILE_TRIS
typedef float pk_f2 __attribute__((ext_vector_type(2)));   /* two floats in a register pair, for the packed two-wide instructions */

/* COUNT: also count, per lane, the bounding-sphere tests made and the triangle tests that followed them */
template <bool COUNT = false>
BDPT_DEV Hit closest_hit_tiled(const bdpt_dev_scene &S, float4 *wave_lds, bool active, clm_f3 o, clm_f3 d, int otype,
                               int oidx, unsigned long long *n_cull = nullptr, unsigned long long *n_exact = nullptr)
{
    Best b;
    best_init(b);
    const int lane = threadIdx.x & 63;
    const int skip_s = otype == 0 ? oidx : -1;
    const int skip_p = otype == 1 ? oidx : -1;
    const int skip_t = otype == 2 ? oidx : -1;
    if (active) {
        for (int i = 0; i < S.n_spheres; i++) {
            const float4 s = uload(S.sph, i);
            if (i != skip_s) test_sphere(b, s, i, o, d);
        }
        for (int i = 0; i < S.n_planes; i++) {
            const float4 q0 = uload(S.pln, (size_t)i * BDPT_PLN_F4 + 0), q1 = uload(S.pln, (size_t)i * BDPT_PLN_F4 + 1);
            if (i != skip_p) test_plane(b, q0, q1, i, o, d);
        }
    }
    /* The test below takes |d| = 1.  Directions come out of normalisations and of combinations of unit vectors,
     * which keeps |d|^2 within a few 1e-7 of 1; a direction built around a plane normal that the scene gives
     * with another length, or a cone sample whose tangent came out of a nearly degenerate cross product, need
     * not be.  Such a ray skips the cull altogether (an infinite slack keeps every triangle). */
    const float dd = clm_dot3(d, d);
    const float oo = clm_dot3(o, o), od = clm_dot3(o, d);
    /* a sphere is out of reach when  w - 2 (1 - 1e-5) c.o - (c.d - o.d)^2  exceeds this (w: see bdpt_layout.h) */
    const float bound = fabsf(dd - 1.0f) <= 4.0e-6f ? 1.0e-5f * (oo + 1.0f) + 2.0e-6f * oo - 0.99999f * oo : INFINITY;
    const clm_f3 o2 = clm_scale3(-2.0f * 0.99999f, o);                      /* the factor of c.o, folded into the ray's side (one more rounding of 6e-8 |c||o|) */
    const int n_tiles = (S.n_triangles + BDPT_TILE_TRIS - 1) / BDPT_TILE_TRIS;
    const size_t total_f4 = ((size_t)S.n_triangles + 2) / 2 * 2;           /* whole pairs of spheres */
    float4 r0, r1, r2;                                                     /* this lane's three float4 of the next tile */
    auto fetch = [&](int tile) {
        const size_t base = (size_t)tile * BDPT_TILE_F4 + lane;
        r0 = S.tri_bound[base < total_f4 ? base : 0];
        r1 = S.tri_bound[base + 64 < total_f4 ? base + 64 : 0];
        r2 = S.tri_bound[base + 128 < total_f4 ? base + 128 : 0];
    };
    /* the ray's constants, each in both halves of a register pair */
    const pk_f2 dx2 = {d.x, d.x}, dy2 = {d.y, d.y}, dz2 = {d.z, d.z}, nod2 = {-od, -od};
    const pk_f2 ox2 = {o2.x, o2.x}, oy2 = {o2.y, o2.y}, oz2 = {o2.z, o2.z};
    if (n_tiles > 0) fetch(0);
    for (int tile = 0; tile < n_tiles; tile++) {
        float4 *buf = wave_lds + (tile & 1) * BDPT_TILE_F4;
        buf[lane] = r0;
        buf[lane + 64] = r1;
        buf[lane + 128] = r2;
        if (tile + 1 < n_tiles) fetch(tile + 1);
        __builtin_amdgcn_wave_barrier();       /* LDS operations of one wavefront execute in order; this only pins the compiler */
        if (active) {
            const int first = tile * BDPT_TILE_TRIS;
            const int count = min(BDPT_TILE_TRIS, S.n_triangles - first);
            if (COUNT) *n_cull += (unsigned long long)count;
            /* see the margin above: seven fused operations per sphere, here as seven packed ones (v_pk_fma_f32) for the two
             * spheres of a pair -- the same operations on the same numbers as one sphere at a time */
            auto excess2 = [&](float4 a, float4 bb) {
                const pk_f2 cx = {a.x, a.y}, cy = {a.z, a.w}, cz = {bb.x, bb.y}, w = {bb.z, bb.w};
                const pk_f2 u = __builtin_elementwise_fma(cz, dz2, __builtin_elementwise_fma(cy, dy2, __builtin_elementwise_fma(cx, dx2, nod2)));
                const pk_f2 t = __builtin_elementwise_fma(cz, oz2, __builtin_elementwise_fma(cy, oy2, __builtin_elementwise_fma(cx, ox2, w)));
                return __builtin_elementwise_fma(-u, u, t);
            };
            auto full = [&](int i, bool wanted) {                            /* the reference's test, for the lanes that want it */
                wanted = wanted && i < S.n_triangles;                        /* the sphere that completes an odd count has no triangle */
                if (__builtin_amdgcn_ballot_w64(wanted) == 0) return;
                const size_t r = (size_t)i * BDPT_TRI_F4;
                const float4 q0 = uload(S.tri, r), q1 = uload(S.tri, r + 1), q2 = uload(S.tri, r + 2),
                             q3 = uload(S.tri, r + 3), q4 = uload(S.tri, r + 4);
                const float e2z = uload(S.tri, r + 5).x;
                if (wanted) test_triangle_scattered(b, q0, q1, q2, q3, q4, e2z, i, skip_t, o, d);
                if (COUNT && wanted) ++*n_exact;
            };
            const int n_pairs = (count + 1) / 2;
            int j = 0;
            for (; j + 4 <= n_pairs; j += 4) {                               /* eight spheres per LDS round trip and per compare */
                const float4 a0 = buf[2 * j], b0 = buf[2 * j + 1], a1 = buf[2 * j + 2], b1 = buf[2 * j + 3], a2 = buf[2 * j + 4], b2 = buf[2 * j + 5],
                             a3 = buf[2 * j + 6], b3 = buf[2 * j + 7];
                const pk_f2 e0 = excess2(a0, b0), e1 = excess2(a1, b1), e2 = excess2(a2, b2), e3 = excess2(a3, b3);
                /* one compare for the eight: no input here is a NaN (upload replaces what could make one by an infinity) */
                const float least = __builtin_fminf(__builtin_fminf(__builtin_fminf(e0.x, e0.y), __builtin_fminf(e1.x, e1.y)),
                                                    __builtin_fminf(__builtin_fminf(e2.x, e2.y), __builtin_fminf(e3.x, e3.y)));
                if (__builtin_amdgcn_ballot_w64(!(least > bound)) != 0) {
                    const int i0 = first + 2 * j;
                    full(i0, !(e0.x > bound));
                    full(i0 + 1, !(e0.y > bound));
                    full(i0 + 2, !(e1.x > bound));
                    full(i0 + 3, !(e1.y > bound));
                    full(i0 + 4, !(e2.x > bound));
                    full(i0 + 5, !(e2.y > bound));
                    full(i0 + 6, !(e3.x > bound));
                    full(i0 + 7, !(e3.y > bound));
                }
            }
            for (; j < n_pairs; j++) {
                const pk_f2 e = excess2(buf[2 * j], buf[2 * j + 1]);
                full(first + 2 * j, !(e.x > bound));
                full(first + 2 * j + 1, !(e.y > bound));
            }
        }
        __builtin_amdgcn_wave_barrier();
    }
    return finish_hit(S, b, d);
}

/* sample_direct with every early exit turned into the `alive` predicate (see above) */
template <bool AOV, bool ALT, bool ALL_RAYS, bool COUNT>
BDPT_DEV bool sample_tiled(const bdpt_kargs &K, float4 *wave_lds, bool live, float xt, float yt, uint32_t &w,
                           uint32_t &draws, uint32_t &rays, uint32_t &hash, int eye_id[2], float col[4],
                           unsigned long long *n_cull, unsigned long long *n_exact)
{
    const bdpt_dev_scene &S = K.scene;
    clm_f3 dir = clm_make3(0.f, 0.f, 1.f), eye_pos = dir;
    if (live) {
        rng_float(w, draws);                                              /* :592, discarded */
        dir = camera_ray(w, draws, K, xt, yt);
        eye_pos = clm_mad3(0.00001f, dir, f3(K.cam_pos));                 /* :632 */
    }
    Hit eye = closest_hit_tiled<COUNT>(S, wave_lds, live, eye_pos, dir, 4, 0, n_cull, n_exact);   /* :634 */
    bool early = false;
    float4 eye_diffuse = make_float4(0.f, 0.f, 0.f, 0.f);
    if (live) {
        rays++;
        if (AOV) { hash = hash_step(hash, hit_id(eye.type, eye.idx)); eye_id[0] = eye.type; eye_id[1] = eye.idx; }
        if (eye.type < 0) {                                               /* :652-656 */
            col[0] = col[1] = col[2] = col[3] = 0.0f;
            early = true;
        } else {
            eye_diffuse = S.surf[eye.surf * 2];
            if (S.surf[eye.surf * 2 + 1].w != 0.0f) {                     /* :645-650 */
                col[0] = eye_diffuse.x; col[1] = eye_diffuse.y; col[2] = eye_diffuse.z; col[3] = eye_diffuse.w;
                early = true;
            }
        }
    }
    bool alive = live && !early;
    clm_f3 vpos = clm_make3(0.f, 0.f, 0.f), vn = clm_make3(0.f, 0.f, 1.f);
    if (alive) sample_light(w, draws, K, vpos, vn);                       /* :669 */
    int vtype = K.light_origin_type, vidx = K.light_origin_index;

    float full[4] = {0.0f, 0.0f, 0.0f, 0.0f};                             /* :689 */
    float acc[4] = {1.0f, 1.0f, 1.0f, 1.0f};                              /* :692 */
    float pre = K.pre_factor0;                                            /* :694 */
    int taken = 0;
    clm_f3 incoming = clm_make3(0.0f, 0.0f, 0.0f);                        /* :659 */
    for (int lti = 0; lti < K.max_l; lti++) {                             /* :696 */
        if (__builtin_amdgcn_ballot_w64(alive) == 0) break;               /* wave-uniform */
        clm_f3 tdir = clm_make3(0.f, 0.f, 1.f);
        if (alive) tdir = clm_normalize3(clm_sub3(eye.pos, vpos));        /* :699 */
        Hit vis = closest_hit_tiled<COUNT>(S, wave_lds, alive, vpos, tdir, vtype, vidx, n_cull, n_exact);   /* :705 */
        clm_f3 lpos = vpos, ldir = tdir;
        if (alive) {
            rays++;
            if (AOV) hash = hash_step(hash, hit_id(vis.type, vis.idx));
            float contrib[4];
            connection(K, eye, vis, vpos, vn, tdir, acc, inv_pre_factor(K, lti, pre), contrib);
            if (!K.indirect_only || lti != 0) {                           /* :752-755 */
#pragma unroll
                for (int k = 0; k < 4; k++) full[k] += contrib[k];
            }
            taken++;                                                      /* :757 */
            lpos = clm_mad3(0.00001f, vn, vpos);                          /* :761 */
            /* as in sample_direct: the fate first, the direction only for the lanes that go on */
            uint32_t w_dir = w, draws_dir = 0;
            const bool four = ALT && sampler_draws_four(K.sampler);
            w *= four ? lcg_pow(4) : lcg_pow(3);
            draws += four ? 4u : 3u;
            const float fate = rng_float(w, draws);                       /* :787 */
            if (fate > K.p_con) alive = false;                            /* :788 */
            if (!ALL_RAYS && lti + 1 == K.max_l) alive = false;           /* nothing reads what the last bounce ray finds */
            if (alive)
                ldir = ALT ? sample_bounce_alt(K, w_dir, draws_dir, vn, incoming, lpos)
                           : sample_cone(w_dir, draws_dir, vn, 3.1415f * 0.50f);   /* :767, :486 */
        }
        pre *= K.p_con;                                                   /* :745 */
        if (__builtin_amdgcn_ballot_w64(alive) == 0) break;
        Hit nxt = closest_hit_tiled<COUNT>(S, wave_lds, alive, lpos, ldir, vtype, vidx, n_cull, n_exact);   /* :793 */
        if (alive) {
            rays++;
            if (AOV) hash = hash_step(hash, hit_id(nxt.type, nxt.idx));
            if (nxt.type < 0) {                                           /* :812 */
                alive = false;
            } else {
                const float4 nd = S.surf[nxt.surf * 2];
                if (S.surf[nxt.surf * 2 + 1].w != 0.0f) {                 /* :797-803 */
                    alive = false;
                } else {
                    acc[0] *= nd.x; acc[1] *= nd.y; acc[2] *= nd.z; acc[3] *= nd.w;   /* :827 with :822, :824 */
                    if (ALT) incoming = ldir;                             /* :831 */
                    vpos = nxt.pos; vn = nxt.n; vtype = nxt.type; vidx = nxt.idx;     /* :832 */
                }
            }
        }
    }
    if (live && !early) {
        const float inv_taken = clm_div(1.0f, (float)taken);              /* :846 */
        const float ed[4] = {eye_diffuse.x, eye_diffuse.y, eye_diffuse.z, eye_diffuse.w};
#pragma unroll
        for (int k = 0; k < 4; k++) {
            float c = K.light_emissive[k] * full[k] * inv_taken;
            if (!K.indirect_only) c = c * ed[k];                          /* :844 / :846 */
            col[k] = c;
        }
    }
    return early;
}

/* Five wavefronts per SIMD (96 VGPRs; the sphere walk needs few, the spills sit in the path logic around it):
 * 100 000 triangles 600 -> 548 ms per 2-sample launch against the four the allocator settles for, 500 triangles
 * 88.7 -> 82.8 ms per 64 samples; six: 543 and 86.2. */
#ifndef BDPT_TILED_WAVES
#define BDPT_TILED_WAVES 5
#endif
template <bool AOV, bool COUNT, bool ALT>
__global__ __launch_bounds__(BDPT_BLOCK_MAX) __attribute__((amdgpu_waves_per_eu(BDPT_TILED_WAVES, BDPT_TILED_WAVES)))
void bdpt_tiled_kernel(const bdpt_kargs K)
{
    float4 *wave_lds = bdpt_lds + (threadIdx.x >> 6) * (2 * BDPT_TILE_F4);   /* dynamic LDS: two tile buffers per wavefront */
    const Slice sl = slice_begin(K);
    if (sl.abandoned) return;
    int x, y;
    lane_pixel(K, sl.tile, x, y);
    const bool inside = x < K.rx0 + K.rw && y < K.ry0 + K.rh;             /* lanes outside still help copy tiles */
    const size_t pix = inside ? (size_t)y * K.width + x : 0;

    float4 a4 = inside ? K.accum[pix] : make_float4(0.f, 0.f, 0.f, 0.f);
    float acc[4] = {a4.x, a4.y, a4.z, a4.w};
    const uint32_t seed = inside ? K.seed_w[pix] : 1u;
    uint32_t w = seed;
    SampleCursor cur;
    if (sl.s0 == 0 || !inside) cursor_init(cur, K, seed);
    else cursor_load(cur, K, pix);

    uint32_t draws = 0, rays = 0, hash = 0;
    if (AOV && inside) { draws = K.aov_rng_draws[pix]; rays = K.aov_rays[pix]; hash = K.aov_hit_hash[pix]; }
    unsigned long long c_rays = 0, c_full = 0, c_early = 0, c_cull = 0, c_exact = 0;
    int eye_id[2] = {-1, -1};
    float col[4] = {0.0f, 0.0f, 0.0f, 0.0f};
    float xt = 0.f, yt = 0.f;
    if (inside) pixel_coord(K, x, y, xt, yt);

    if (K.reorder == 2 && sl.s0 == 0) {                   /* ADAPTIVE order, as in bdpt_direct_kernel */
        uint32_t wp = cur.st_head, dp = 0;
        clm_f3 dir = clm_make3(0.f, 0.f, 1.f);
        if (inside) {
            rng_float(wp, dp);
            dir = camera_ray(wp, dp, K, xt, yt);
        }
        const Hit e = closest_hit_tiled<COUNT>(K.scene, wave_lds, inside, clm_mad3(0.00001f, dir, f3(K.cam_pos)), dir, 4, 0, &c_cull, &c_exact);
        const bool early = e.type < 0 || K.scene.surf[e.surf * 2 + 1].w != 0.0f;
        cur.cls = early ? -1 : cur.cls;
        if (COUNT && !AOV && inside) c_rays++;
    }

    bool live = inside;                                   /* false once the pixel has no more samples in this launch */
    for (int s = sl.s0; s < sl.s1; s++) {                                 /* s counts the samples taken so far */
        uint32_t start = 0;
        if (live) live = cursor_next(cur, K, &K.seed_w[pix], start);
        if (__builtin_amdgcn_ballot_w64(live) == 0) break;
        const int n = K.num_samples + s;                                  /* kernel0.cl:588 */
        if (K.rng_mode == BDPT_RNG_STRIDED) w = start;
        const uint32_t w_before = w;
        const uint32_t rays_before = rays;
        const bool early = sample_tiled<AOV, ALT, AOV, COUNT>(K, wave_lds, live, xt, yt, w, draws, rays, hash, eye_id, col, &c_cull, &c_exact);
        if (!live) continue;
        if (COUNT) { c_rays += rays - rays_before; if (early) c_early++; else c_full++; }
        if (early && K.rng_mode == BDPT_RNG_STREAM) {
            w = w_before;
            acc[0] = col[0]; acc[1] = col[1]; acc[2] = col[2]; acc[3] = col[3];
            const int rest = sl.s1 - 1 - s;
            if (AOV) {
                const uint32_t id = hit_id(eye_id[0], eye_id[1]);
                for (int r = 0; r < rest; r++) hash = hash_step(hash, id);
                draws += 3u * (uint32_t)rest;
                rays += (uint32_t)rest;
            }
            if (COUNT) c_early += (unsigned long long)rest;
            live = false;
            continue;
        }
        if (n == 0) {                                                     /* :855 */
#pragma unroll
            for (int k = 0; k < 4; k++) acc[k] = col[k];
        } else if (K.accum_mode == BDPT_ACCUM_SUM) {
#pragma unroll
            for (int k = 0; k < 4; k++) acc[k] = acc[k] + col[k];
        } else {                                                          /* :854 */
            const float s1 = clm_div((float)n, (float)(n + 1));
            const float s2 = clm_div(1.0f, (float)(n + 1));
#pragma unroll
            for (int k = 0; k < 4; k++) acc[k] = clm_fma(col[k], s2, acc[k] * s1);
        }
    }
    if (inside) {
        pixel_state_store<AOV>(K, sl, pix, acc, w, cur, eye_id, hash, col, draws, rays);
        if (COUNT) {
            atomicAdd(&K.counters[0], c_rays);
            atomicAdd(&K.counters[1], c_full);
            atomicAdd(&K.counters[2], c_early);
            atomicAdd(&K.counters[3], c_cull);
            atomicAdd(&K.counters[4], c_exact);
        }
    }
    slice_end(K, sl);
}

/* ------------------------------------------------------------------------- */
/* path_trace, kernel0.cl:872-1001: the unidirectional kernel the reference's file
 * keeps commented out.  One lane per pixel, spp samples per launch as that
 * kernel's own sample loop (:908) runs them; the plain walk over the primitives.  */
/* ------------------------------------------------------------------------- */

template <bool AOV>
__global__ __launch_bounds__(BDPT_BLOCK_MAX) void bdpt_path_trace_kernel(const bdpt_kargs K)
{
    int x, y;
    lane_pixel(K, (int)(blockIdx.y * gridDim.x + blockIdx.x), x, y);
    if (x >= K.rx0 + K.rw || y >= K.ry0 + K.rh) return;
    const size_t pix = (size_t)y * K.width + x;
    const bdpt_dev_scene &S = K.scene;
    float4 a4 = K.accum[pix];
    float prev[4] = {a4.x, a4.y, a4.z, a4.w};                             /* :887 */
    uint32_t w = K.seed_w[pix];                                           /* :890 */
    uint32_t draws = 0, rays = 0, hash = 0;
    if (AOV) { draws = K.aov_rng_draws[pix]; rays = K.aov_rays[pix]; hash = K.aov_hit_hash[pix]; }
    int eye_id[2] = {-1, -1};
    float col[4] = {0.0f, 0.0f, 0.0f, 0.0f};
    rng_float(w, draws);                                                  /* :892, discarded; once per launch */
    float xt, yt;
    pixel_coord(K, x, y, xt, yt);                                         /* :899-903 */
    const float wt_brdf = (float)((double)0.5f * CLM_1_PI_F64);           /* launchDir.w, :505 */
    const float inv_2pi = (float)CLM_1_PI_F64 * 0.5f;                     /* (float)M_1_PI * 0.5f, :971 */
    for (int k = 0; k < K.spp; k++) {                                     /* :908 */
        const int n = K.num_samples + k;
        clm_f3 vdir = camera_ray(w, draws, K, xt, yt);                    /* :910-912; eyePatch.vDir :920 */
        clm_f3 vpos = f3(K.cam_pos);                                      /* :921 */
        int vtype = -1, vidx = -1;                                        /* :922-923 */
        col[0] = col[1] = col[2] = col[3] = 1.0f;                         /* :925 */
        clm_f3 incoming = clm_make3(0.0f, 0.0f, 0.0f);                    /* :930 */
        for (int i = 0; i < 5; i++) {                                     /* :932 */
            const clm_f3 lpos = clm_mad3(0.00001f, vdir, vpos);           /* :936 */
            /* brdf_reflective (:943, :492-508): p4, then a cone about the mirror direction or about the normal */
            const float p4 = rng_float(w, draws);
            const bool mirror = p4 > 0.6f;
            const clm_f3 axis = mirror ? clm_reflect3(vdir, incoming) : vdir;
            clm_f3 ldir = sample_cone(w, draws, axis, mirror ? 3.1415f * 0.02f : 3.1415f * 0.50f);
            float wt = wt_brdf;                                           /* :947 */
            if (i == 0) { wt = 1.0f; ldir = vdir; }                       /* :951-954 */
            if (i == 4) {                                                 /* :956-959 */
#pragma unroll
                for (int c = 0; c < 4; c++) col[c] *= 0.3f;
                break;
            }
            const Hit h = closest_hit_direct(S, lpos, ldir, vtype, vidx); /* :964 */
            rays++;
            if (AOV) {
                hash = hash_step(hash, hit_id(h.type, h.idx));
                if (i == 0) { eye_id[0] = h.type; eye_id[1] = h.idx; }
            }
            incoming = ldir;                                              /* :966 */
            if (h.type < 0) {                                             /* :978-982 */
                col[0] *= 0.3f; col[1] *= 0.3f; col[2] *= 0.4f; col[3] *= 0.3f;
                break;
            }
            const float4 sd = S.surf[h.surf * 2], se = S.surf[h.surf * 2 + 1];
            if (se.w != 0.0f) {                                           /* :969-974: (emissive / wt) * (M_1_PI * 0.5f) */
                col[0] *= clm_div(se.x, wt) * inv_2pi; col[1] *= clm_div(se.y, wt) * inv_2pi;
                col[2] *= clm_div(se.z, wt) * inv_2pi; col[3] *= clm_div(se.w, wt) * inv_2pi;
                break;
            }
            const float m1pi = (float)CLM_1_PI_F64;                       /* :976: ((diffuse / wt) * M_1_PI) * 0.5f */
            col[0] *= clm_div(sd.x, wt) * m1pi * 0.5f; col[1] *= clm_div(sd.y, wt) * m1pi * 0.5f;
            col[2] *= clm_div(sd.z, wt) * m1pi * 0.5f; col[3] *= clm_div(sd.w, wt) * m1pi * 0.5f;
            vpos = h.pos; vdir = h.n; vtype = h.type; vidx = h.idx;       /* :984 */
        }
        if (n == 0) {                                                     /* :991-992 */
#pragma unroll
            for (int c = 0; c < 4; c++) prev[c] = col[c];
        } else {                                                          /* :990 */
            const float s1 = clm_div((float)n, (float)(n + 1));
            const float s2 = clm_div(1.0f, (float)(n + 1));
#pragma unroll
            for (int c = 0; c < 4; c++) prev[c] = clm_fma(col[c], s2, prev[c] * s1);
        }
    }
    K.seed_w[pix] = w;                                                    /* :998 */
    K.accum[pix] = make_float4(prev[0], prev[1], prev[2], prev[3]);       /* :1000 */
    if (AOV) {
        K.aov_eye_hit[pix * 2] = eye_id[0];
        K.aov_eye_hit[pix * 2 + 1] = eye_id[1];
        K.aov_hit_hash[pix] = hash;
        K.aov_radiance[pix] = make_float4(col[0], col[1], col[2], col[3]);
        K.aov_rng_draws[pix] = draws;
        K.aov_rays[pix] = rays;
    }
}

/* dst += src and dst *= f over n float4: combining the partial sums of several GPUs on one of them */
__global__ void bdpt_add_kernel(float4 *dst, const float4 *src, size_t n)
{
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float4 a = dst[i], b = src[i];
    dst[i] = make_float4(a.x + b.x, a.y + b.y, a.z + b.z, a.w + b.w);
}

__global__ void bdpt_scale_kernel(float4 *dst, size_t n, float f)
{
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float4 a = dst[i];
    dst[i] = make_float4(a.x * f, a.y * f, a.z * f, a.w * f);
}

/* Closest hit for caller-supplied rays: the unit the known-answer tests drive. */
__global__ void bdpt_query_kernel(bdpt_dev_scene S, const float *rays, const int *origin, int n, int *out_id,
                                  float *out_pos, float *out_n)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    clm_f3 o = clm_make3(rays[i * 6 + 0], rays[i * 6 + 1], rays[i * 6 + 2]);
    clm_f3 d = clm_make3(rays[i * 6 + 3], rays[i * 6 + 4], rays[i * 6 + 5]);
    Hit h = closest_hit_direct(S, o, d, origin[i * 2], origin[i * 2 + 1]);
    out_id[i * 2] = h.type;
    out_id[i * 2 + 1] = h.idx;
    out_pos[i * 3 + 0] = h.pos.x; out_pos[i * 3 + 1] = h.pos.y; out_pos[i * 3 + 2] = h.pos.z;
    out_n[i * 3 + 0] = h.n.x; out_n[i * 3 + 1] = h.n.y; out_n[i * 3 + 2] = h.n.z;
}

/* Evaluates the numeric contract's built-ins on the device so that a test can
 * compare them bit for bit with the host evaluation of the same header.
 * op: 0 sin, 1 cos, 2 acosf, 3 acos_f64 (in/out as double), 4 sqrt, 5 a/b,
 * 6 fma(a,b,c), 7 1/sqrt-normalised x of (a,b,c). */
__global__ void bdpt_clmath_kernel(int op, const double *in, double *out, int n)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const double a = in[i * 3], b = in[i * 3 + 1], c = in[i * 3 + 2];
    double r = 0.0;
    float s, co;
    switch (op) {
    case 0: clm_sincos((float)a, &s, &co); r = s; break;
    case 1: clm_sincos((float)a, &s, &co); r = co; break;
    case 2: r = clm_acosf((float)a); break;
    case 3: r = clm_acos_f64(a); break;
    case 4: r = clm_sqrt((float)a); break;
    case 5: r = clm_div((float)a, (float)b); break;
    case 6: r = clm_fma((float)a, (float)b, (float)c); break;
    case 7: r = clm_normalize3(clm_make3((float)a, (float)b, (float)c)).x; break;
    default: break;
    }
    out[i] = r;
}

}  // namespace bdpt

#include "bdpt_pool.h"
#include "bdpt_class_pool.h"

/* ---- launch wrappers ------------------------------------------------------ */

/* grid of a per-pixel launch: see lane_pixel */
static dim3 pixel_grid(const bdpt_kargs &K)
{
    if (K.pool_rows) {                                   /* whole region only: see bdpt_pool_supported */
        const int block_w = (K.block >> 6) * K.pool_w;
        return dim3((K.rw + block_w - 1) / block_w, (K.rh + K.pool_rows - 1) / K.pool_rows);
    }
    const int block_w = K.block >> 3;
    if (K.tile_stride <= 1) return dim3((K.rw + block_w - 1) / block_w, (K.rh + 7) / 8);
    const int tiles_y = (K.rh + K.tile_h - 1) / K.tile_h, n_tiles = K.tiles_x * tiles_y;
    const int owned = n_tiles > K.tile_phase ? (n_tiles - K.tile_phase + K.tile_stride - 1) / K.tile_stride : 0;
    return dim3((unsigned)owned * (K.tile_w / block_w) * (K.tile_h / 8), 1);
}

/* Grid of a launch that may be sliced.  On entry K.n_tiles is the number of tiles K.slice_sync has room for; a
 * launch with more tiles than that runs unsliced.  Clears the ticket counter and the per-tile words. */
static hipError_t sliced_grid(bdpt_kargs &K, dim3 &grid, hipStream_t stream)
{
    grid = pixel_grid(K);
    const long long tiles = (long long)grid.x * grid.y, room = K.n_tiles;
    K.wg_x = (int)grid.x;
    K.n_tiles = (int)tiles;
    if (K.n_chunks > 1 && (tiles > room || tiles * K.n_chunks > 0x7fffffffLL || !K.slice_sync || !K.slice_cursor || !K.slice_abort)) K.n_chunks = 1;
    if (K.n_chunks <= 1 || tiles == 0) { K.n_chunks = 1; K.chunk_spp = K.spp; return hipSuccess; }
    grid = dim3((unsigned)(tiles * K.n_chunks));
    /* ticket counter, give-up mark and the tiles' words; the allocation is padded so that whole 16-byte units can be cleared */
    const size_t bytes = ((size_t)(BDPT_SLICE_HEADER + tiles) * sizeof(uint32_t) + 15) / 16 * 16;
    return hipMemsetAsync(K.slice_sync, 0, bytes < 256 ? 256 : bytes, stream);
}

/* Threads per workgroup of a launch whose workgroups need `per_wave` bytes of LDS for each wavefront and `shared`
 * bytes once: one wavefront where 24 such workgroups (six per SIMD) fit the 160 KB of a compute unit, else two. */
static int launch_block(size_t shared, size_t per_wave)
{
    return 24 * (shared + per_wave) <= 160 * 1024 ? 64 : BDPT_BLOCK_MAX;
}

template <int MODE>
static hipError_t launch_per_pixel(bdpt_kargs K, bool aov, bool count, hipStream_t stream)
{
    if (K.rw <= 0 || K.rh <= 0 || K.spp <= 0) return hipErrorInvalidValue;
    size_t lds = 0;
    K.block = 64;
    if (MODE == bdpt::MODE_LDS || MODE == bdpt::MODE_LDS2) {
        lds = bdpt_lds_scene_bytes(K.scene, MODE == bdpt::MODE_LDS2);
        if (lds > BDPT_LDS_SCENE_MAX) return hipErrorInvalidValue;
        const size_t park = (size_t)64 * BDPT_PARK_F4 * sizeof(float4);   /* the lanes' parked eye vertices, behind the scene */
        K.block = launch_block(lds, park);
        lds += park * (K.block / 64);
        if (MODE == bdpt::MODE_LDS2 && K.scene.n_triangles > BDPT_CULL_MAX_TRIANGLES) return hipErrorInvalidValue;
    }
    if (MODE == bdpt::MODE_BVH) {
        /* a hierarchy has leaves; one with inner nodes has those too */
        if (K.scene.bvh_root != BDPT_BVH_NONE && (!K.scene.bvh_leaf_tri || (K.scene.bvh_root >= 0 && !K.scene.bvh_nodes))) return hipErrorInvalidValue;
        if (K.scene.bvh_n_brute > 0 && !K.scene.bvh_brute) return hipErrorInvalidValue;
        if (K.scene.bvh_stack_limit < 1 || K.scene.bvh_stack_limit > BDPT_BVH_STACK) return hipErrorInvalidValue;
        if (K.scene.bvh_stack_total < K.scene.bvh_stack_limit || K.scene.bvh_stack_total > BDPT_BVH_DEEP) return hipErrorInvalidValue;
        lds = (size_t)K.scene.bvh_stack_limit * 64 * sizeof(uint32_t) + BDPT_BVH_SHARE_BYTES;   /* one wavefront per workgroup: its lanes' stacks, what they share */
    }
    dim3 grid;
    const dim3 block(K.block);
    const hipError_t e = sliced_grid(K, grid, stream);
    if (e != hipSuccess) return e;
    if (grid.x == 0) return hipSuccess;                  /* this share owns no tile of the region */
    const bool alt = K.sampler != BDPT_SAMPLER_LAMBERT_CONE;
#define BDPT_LAUNCH(A, C, X) hipLaunchKernelGGL((bdpt::bdpt_direct_kernel<A, C, MODE, X>), grid, block, lds, stream, K)
    if (alt) {
        if (aov && count) BDPT_LAUNCH(true, true, true);
        else if (aov) BDPT_LAUNCH(true, false, true);
        else if (count) BDPT_LAUNCH(false, true, true);
        else BDPT_LAUNCH(false, false, true);
    } else {
        if (aov && count) BDPT_LAUNCH(true, true, false);
        else if (aov) BDPT_LAUNCH(true, false, false);
        else if (count) BDPT_LAUNCH(false, true, false);
        else BDPT_LAUNCH(false, false, false);
    }
#undef BDPT_LAUNCH
    return hipGetLastError();
}

size_t bdpt_lds_scene_bytes(const bdpt_dev_scene &S, bool with_cull_records)
{
    const size_t ids = ((size_t)S.n_triangles + S.n_spheres + S.n_planes + 3) / 4;         /* uint32 ids, in float4 */
    return ((size_t)S.n_triangles * BDPT_TRI_F4 + (with_cull_records ? (size_t)S.cull_f4 : 0) + (size_t)S.n_spheres +
            (size_t)S.n_planes * BDPT_PLN_F4 + (size_t)S.n_surfaces * 2 + ids) * sizeof(float4);
}

hipError_t bdpt_launch_direct(const bdpt_kargs &K, bool aov, bool count, hipStream_t stream)
{
    return launch_per_pixel<bdpt::MODE_DIRECT>(K, aov, count, stream);
}

hipError_t bdpt_launch_lds(const bdpt_kargs &K, bool aov, bool count, hipStream_t stream)
{
    return launch_per_pixel<bdpt::MODE_LDS>(K, aov, count, stream);
}

hipError_t bdpt_launch_lds2(const bdpt_kargs &K, bool aov, bool count, hipStream_t stream)
{
    return launch_per_pixel<bdpt::MODE_LDS2>(K, aov, count, stream);
}

hipError_t bdpt_launch_bvh(const bdpt_kargs &K, bool aov, bool count, hipStream_t stream)
{
    return launch_per_pixel<bdpt::MODE_BVH>(K, aov, count, stream);
}

/* Paths sorted by phase (bdpt_pool.h).  Wavefronts per workgroup: as many as keep the most wavefronts on a compute unit
 * (one copy of the scene per workgroup, one pool per wavefront, 160 KB of LDS per compute unit, 64 KB per workgroup). */
bool bdpt_pool_supported(const bdpt_kargs &K, int variant)
{
    if (variant != BDPT_KERNEL_LDS && variant != BDPT_KERNEL_LDS2) return false;
    if (K.sampler != BDPT_SAMPLER_LAMBERT_CONE || K.tile_stride > 1) return false;
    if ((uint32_t)K.scene.n_surfaces > BDPT_POOL_SURF_MASK || K.max_l > 255) return false;     /* what a slot's misc word has room for */
    if (variant == BDPT_KERNEL_LDS2 && K.scene.n_triangles > BDPT_CULL_MAX_TRIANGLES) return false;
    return bdpt_lds_scene_bytes(K.scene, variant == BDPT_KERNEL_LDS2) <= BDPT_LDS_SCENE_MAX;
}

/* wavefronts per workgroup of a pooled launch, and the bytes of K.pool_state it needs for `chunks` runs per tile */
static int pool_waves_per_workgroup(const bdpt_kargs &K, int variant)
{
    const size_t scene = bdpt_lds_scene_bytes(K.scene, variant == BDPT_KERNEL_LDS2), pool = (size_t)BDPT_POOL_WAVE_F4 * sizeof(float4);
    int nw = 0;
    size_t best = 0;
    for (int n = 1; n <= 4; n++) {
        const size_t wg = scene + n * pool + 64;                              /* + the words slice_begin keeps */
        if (wg > 64 * 1024) break;
        const size_t waves = std::min((size_t)(160 * 1024) / wg, (size_t)32 / n) * n;
        if (waves > best) { best = waves; nw = n; }
    }
    return nw;
}

size_t bdpt_pool_state_bytes(const bdpt_kargs &K, int variant, int chunks)
{
    const int nw = pool_waves_per_workgroup(K, variant);
    if (nw == 0) return 0;
    const size_t block_w = (size_t)nw * BDPT_POOL_W;
    const size_t tiles = ((K.rw + block_w - 1) / block_w) * ((K.rh + BDPT_POOL_ROWS - 1) / BDPT_POOL_ROWS);
    return tiles * (size_t)std::max(1, chunks) * nw * BDPT_POOL_SLOTS * BDPT_POOL_GLOBAL_F4 * sizeof(float4);
}

template <int MODE>
static hipError_t launch_pool(bdpt_kargs K, bool aov, bool count, hipStream_t stream)
{
    if (K.rw <= 0 || K.rh <= 0 || K.spp <= 0 || !K.pool_state || !K.xt_tab || !K.yt_tab) return hipErrorInvalidValue;
    const int variant = MODE == bdpt::MODE_LDS2 ? BDPT_KERNEL_LDS2 : BDPT_KERNEL_LDS;
    const int nw = pool_waves_per_workgroup(K, variant);
    if (nw == 0) return hipErrorInvalidValue;
    K.block = 64 * nw;
    K.pool_w = BDPT_POOL_W;
    K.pool_rows = BDPT_POOL_ROWS;
    const size_t lds = bdpt_lds_scene_bytes(K.scene, MODE == bdpt::MODE_LDS2) + (size_t)nw * BDPT_POOL_WAVE_F4 * sizeof(float4);
    dim3 grid;
    const dim3 block(K.block);
    const hipError_t e = sliced_grid(K, grid, stream);
    if (e != hipSuccess) return e;
    if (grid.x == 0) return hipSuccess;
#define BDPT_LAUNCH(A, C) hipLaunchKernelGGL((bdpt::bdpt_pool_kernel<A, C, MODE>), grid, block, lds, stream, K)
    if (aov && count) BDPT_LAUNCH(true, true);
    else if (aov) BDPT_LAUNCH(true, false);
    else if (count) BDPT_LAUNCH(false, true);
    else BDPT_LAUNCH(false, false);
#undef BDPT_LAUNCH
    return hipGetLastError();
}

hipError_t bdpt_launch_pool(const bdpt_kargs &K, int variant, bool aov, bool count, hipStream_t stream)
{
    if (!bdpt_pool_supported(K, variant)) return hipErrorInvalidValue;
    return variant == BDPT_KERNEL_LDS2 ? launch_pool<bdpt::MODE_LDS2>(K, aov, count, stream) : launch_pool<bdpt::MODE_LDS>(K, aov, count, stream);
}

/* Pixels sorted by the class of their next sample (bdpt_class_pool.h): launches that take every pixel's samples in index
 * order, with the per-pixel kernels' workgroup shapes (one wavefront where 24 workgroups fit a compute unit, else two). */
bool bdpt_class_pool_supported(const bdpt_kargs &K, int variant)
{
    if (variant != BDPT_KERNEL_DIRECT && variant != BDPT_KERNEL_LDS && variant != BDPT_KERNEL_LDS2 && variant != BDPT_KERNEL_BVH) return false;
    if (K.reorder != 0 || K.tile_stride > 1 || K.max_l > BDPT_CPOOL_MAX_CLASS || K.spp >= (1 << 24)) return false;
    if (variant == BDPT_KERNEL_LDS2 && K.scene.n_triangles > BDPT_CULL_MAX_TRIANGLES) return false;
    return variant == BDPT_KERNEL_DIRECT || variant == BDPT_KERNEL_BVH || bdpt_lds_scene_bytes(K.scene, variant == BDPT_KERNEL_LDS2) <= BDPT_LDS_SCENE_MAX;
}

/* Threads per workgroup and bytes of LDS of such a launch: the fewest wavefronts per workgroup (they share one copy of
 * the scene) with which 24 wavefronts, six per SIMD, fit the 160 KB of a compute unit. */
static void class_pool_shape(const bdpt_kargs &K, int variant, int &block, size_t &lds)
{
    const size_t pool = (size_t)BDPT_CPOOL_WAVE_F4 * sizeof(float4);
    if (variant == BDPT_KERNEL_DIRECT || variant == BDPT_KERNEL_BVH) {
        block = 64;
        lds = pool + (variant == BDPT_KERNEL_BVH ? (size_t)K.scene.bvh_stack_limit * 64 * sizeof(uint32_t) + BDPT_BVH_SHARE_BYTES : 0);   /* the hierarchy walk's stacks come first */
        return;
    }
    const size_t scene = bdpt_lds_scene_bytes(K.scene, variant == BDPT_KERNEL_LDS2);
    const size_t park = (size_t)64 * BDPT_PARK_F4 * sizeof(float4);       /* the lanes' parked eye vertices, behind the scene */
    int nw = 1;
    while (nw < BDPT_CPOOL_BLOCK_MAX / 64 && (24 / nw) * (scene + nw * (park + pool)) > 160 * 1024) nw *= 2;
    block = 64 * nw;
    lds = scene + nw * (park + pool);
}

template <int MODE>
static hipError_t launch_class_pool(bdpt_kargs K, bool aov, bool count, hipStream_t stream)
{
    if (K.rw <= 0 || K.rh <= 0 || K.spp <= 0) return hipErrorInvalidValue;
    size_t lds = 0;
    class_pool_shape(K, MODE == bdpt::MODE_LDS2 ? BDPT_KERNEL_LDS2 : MODE == bdpt::MODE_LDS ? BDPT_KERNEL_LDS : MODE == bdpt::MODE_BVH ? BDPT_KERNEL_BVH : BDPT_KERNEL_DIRECT, K.block, lds);
    K.pool_w = BDPT_CPOOL_W;
    K.pool_rows = BDPT_CPOOL_ROWS;
    dim3 grid;
    const dim3 block(K.block);
    const hipError_t e = sliced_grid(K, grid, stream);
    if (e != hipSuccess) return e;
    if (grid.x == 0) return hipSuccess;
    const bool alt = K.sampler != BDPT_SAMPLER_LAMBERT_CONE;
#define BDPT_LAUNCH(A, C, X) hipLaunchKernelGGL((bdpt::bdpt_class_pool_kernel<A, C, MODE, X>), grid, block, lds, stream, K)
    if (alt) {
        if (aov && count) BDPT_LAUNCH(true, true, true);
        else if (aov) BDPT_LAUNCH(true, false, true);
        else if (count) BDPT_LAUNCH(false, true, true);
        else BDPT_LAUNCH(false, false, true);
    } else {
        if (aov && count) BDPT_LAUNCH(true, true, false);
        else if (aov) BDPT_LAUNCH(true, false, false);
        else if (count) BDPT_LAUNCH(false, true, false);
        else BDPT_LAUNCH(false, false, false);
    }
#undef BDPT_LAUNCH
    return hipGetLastError();
}

hipError_t bdpt_launch_class_pool(const bdpt_kargs &K, int variant, bool aov, bool count, hipStream_t stream)
{
    if (!bdpt_class_pool_supported(K, variant)) return hipErrorInvalidValue;
    if (variant == BDPT_KERNEL_LDS2) return launch_class_pool<bdpt::MODE_LDS2>(K, aov, count, stream);
    if (variant == BDPT_KERNEL_LDS) return launch_class_pool<bdpt::MODE_LDS>(K, aov, count, stream);
    if (variant == BDPT_KERNEL_BVH) return launch_class_pool<bdpt::MODE_BVH>(K, aov, count, stream);
    return launch_class_pool<bdpt::MODE_DIRECT>(K, aov, count, stream);
}

hipError_t bdpt_launch_tiled(const bdpt_kargs &K_in, bool aov, bool count, hipStream_t stream)
{
    bdpt_kargs K = K_in;
    if (K.rw <= 0 || K.rh <= 0 || K.spp <= 0) return hipErrorInvalidValue;
    const size_t tile_bytes = (size_t)2 * BDPT_TILE_F4 * sizeof(float4);  /* per wavefront */
    K.block = launch_block(0, tile_bytes);
    dim3 grid;
    const dim3 block(K.block);
    const hipError_t e = sliced_grid(K, grid, stream);
    if (e != hipSuccess) return e;
    if (grid.x == 0) return hipSuccess;
    const bool alt = K.sampler != BDPT_SAMPLER_LAMBERT_CONE;
    const size_t lds = tile_bytes * (K.block / 64);
#define BDPT_LAUNCH(A, C, X) hipLaunchKernelGGL((bdpt::bdpt_tiled_kernel<A, C, X>), grid, block, lds, stream, K)
    if (alt) {
        if (aov && count) BDPT_LAUNCH(true, true, true);
        else if (aov) BDPT_LAUNCH(true, false, true);
        else if (count) BDPT_LAUNCH(false, true, true);
        else BDPT_LAUNCH(false, false, true);
    } else {
        if (aov && count) BDPT_LAUNCH(true, true, false);
        else if (aov) BDPT_LAUNCH(true, false, false);
        else if (count) BDPT_LAUNCH(false, true, false);
        else BDPT_LAUNCH(false, false, false);
    }
#undef BDPT_LAUNCH
    return hipGetLastError();
}

hipError_t bdpt_launch_add(float4 *dst, const float4 *src, size_t n, hipStream_t stream)
{
    if (n == 0) return hipSuccess;
    hipLaunchKernelGGL(bdpt::bdpt_add_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, dst, src, n);
    return hipGetLastError();
}

hipError_t bdpt_launch_scale(float4 *dst, size_t n, float factor, hipStream_t stream)
{
    if (n == 0) return hipSuccess;
    hipLaunchKernelGGL(bdpt::bdpt_scale_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, dst, n, factor);
    return hipGetLastError();
}

hipError_t bdpt_launch_path_trace(const bdpt_kargs &K_in, bool aov, hipStream_t stream)
{
    bdpt_kargs K = K_in;
    if (K.rw <= 0 || K.rh <= 0 || K.spp <= 0) return hipErrorInvalidValue;
    K.block = 64;
    const dim3 grid = pixel_grid(K), block(K.block);
    if (grid.x == 0) return hipSuccess;
    if (K.n_chunks > 1) return hipErrorInvalidValue;    /* this kernel has no sliced form */
    if (aov) hipLaunchKernelGGL(bdpt::bdpt_path_trace_kernel<true>, grid, block, 0, stream, K);
    else hipLaunchKernelGGL(bdpt::bdpt_path_trace_kernel<false>, grid, block, 0, stream, K);
    return hipGetLastError();
}

hipError_t bdpt_launch_query(const bdpt_dev_scene &S, const float *rays, const int *origin, int n, int *out_id,
                             float *out_pos, float *out_n, hipStream_t stream)
{
    if (n <= 0) return hipSuccess;
    hipLaunchKernelGGL(bdpt::bdpt_query_kernel, dim3((n + 255) / 256), dim3(256), 0, stream, S, rays, origin, n, out_id,
                       out_pos, out_n);
    return hipGetLastError();
}

/* ---- device-resident staging for bdpt_query_rays_noalloc -------------------- */
/* One array per GPU, held from the moment the library is loaded on it until the process ends.
 * Layout, in 32-bit words, for a capacity of BDPT_QUERY_STATIC_MAX rays: rays [6*MAX] | origin ids
 * [2*MAX] | hit ids [2*MAX] | hit positions [3*MAX] | normals [3*MAX].  The name `scratch` is only a
 * host-side handle of the symbol; its device address is looked up once per device below and that
 * pointer is what copies and kernels are given. */
static __device__ __align__(256) uint32_t scratch[(size_t)BDPT_QUERY_STATIC_MAX * 16];

#include <mutex>
static std::mutex g_scratch_mutex;
static void *g_scratch_ptr[BDPT_MAX_DEVICES];   /* device address of `scratch` on each GPU, null until looked up */

hipError_t bdpt_query_scratch(int device, void **out)
{
    if (!out || device < 0 || device >= BDPT_MAX_DEVICES) return hipErrorInvalidValue;
    std::lock_guard<std::mutex> lock(g_scratch_mutex);
    if (!g_scratch_ptr[device]) {
        int current = -1;
        hipError_t e = hipGetDevice(&current);
        if (e != hipSuccess) return e;
        if (current != device) return hipErrorInvalidDevice;      /* the symbol is resolved for the current device */
        void *p = nullptr;
        e = hipGetSymbolAddress(&p, HIP_SYMBOL(scratch));
        if (e != hipSuccess) return e;
        if (!p) return hipErrorInvalidSymbol;
        g_scratch_ptr[device] = p;
    }
    *out = g_scratch_ptr[device];
    return hipSuccess;
}

hipError_t bdpt_launch_clmath(int op, const double *in, double *out, int n, hipStream_t stream)
{
    if (n <= 0) return hipSuccess;
    hipLaunchKernelGGL(bdpt::bdpt_clmath_kernel, dim3((n + 255) / 256), dim3(256), 0, stream, op, in, out, n);
    return hipGetLastError();
}

#ifdef BDPT_PHASE_CLOCKS
/* instrumented builds only: the clock sums of every launch since the last reset (tools/phase_clocks.py) */
extern "C" int bdpt_phase_clocks(unsigned long long out[12], int reset)
{
    hipError_t e = hipDeviceSynchronize();
    if (e == hipSuccess) e = hipMemcpyFromSymbol(out, HIP_SYMBOL(bdpt::bdpt_phase_clocks_dev), 12 * sizeof(unsigned long long));
    if (e == hipSuccess && reset) {
        const unsigned long long zero[12] = {};
        e = hipMemcpyToSymbol(HIP_SYMBOL(bdpt::bdpt_phase_clocks_dev), zero, sizeof zero);
    }
    return e == hipSuccess ? 0 : -1;
}
#endif

#ifdef BDPT_BVH_STATS
/* reads and clears the walk's statistics (tools/bvh_stats.py; builds with -DBDPT_BVH_STATS only) */
extern "C" void bdpt_bvh_stats_read(unsigned long long *out)
{
    (void)hipDeviceSynchronize();
    (void)hipMemcpyFromSymbol(out, HIP_SYMBOL(bdpt::bvh_stats), 8 * sizeof(unsigned long long));
    const unsigned long long zero[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    (void)hipMemcpyToSymbol(HIP_SYMBOL(bdpt::bvh_stats), zero, sizeof zero);
}
#endif

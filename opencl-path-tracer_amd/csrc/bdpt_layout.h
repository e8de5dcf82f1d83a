/*
 * bdpt_layout.h -- device-side data layout and the by-value kernel argument.
 *
 * The upload structs of include/bdpt.h (the reference's host AoS structs,
 * src/include/geometry/primitives.h:6-64) are re-laid out once per upload into
 * records of float4 that a wavefront reads with one uniform address per
 * primitive.  Everything that does not depend on the ray is computed at upload
 * with the same arithmetic the per-ray code of the reference uses, so hoisting
 * it changes no bit of any result:
 *
 *   sphere   1 x float4 : (c.x, c.y, c.z, r*r)            kernel0.cl:331-337
 *   plane    2 x float4 : (n.xyz, -), (pt.xyz, -)         kernel0.cl:249-264
 *   triangle 6 x float4 :                                  kernel0.cl:285-300
 *       q0 (n.x,  n.y,  n.z,  a.x)    n  = normalize(cross(b-a, c-a))
 *       q1 (a.y,  a.z,  e0.x, e0.y)   e0 = c-a
 *       q2 (e0.z, c.x,  c.y,  c.z)
 *       q3 (e1.x, e1.y, e1.z, b.x)    e1 = b-c
 *       q4 (b.y,  b.z,  e2.x, e2.y)   e2 = a-b
 *       q5 (e2.z, ns.x, ns.y, ns.z)   ns = normalize(cross(a-b, b-c)), the
 *                                     shading normal of kernel0.cl:467
 *   triangle bound 1 x float4 : (centre c, (1 - 1e-5 - 2e-6) |c|^2 - 1.05 radius^2) of a sphere around the three
 *       vertices: what the tiled kernel streams; a ray that passes the sphere by more than
 *       rounding can explain cannot hit the triangle, and only the others read the record.  Stored two spheres
 *       A, B (triangles 2j, 2j+1) per pair of float4, component by component: (cx_A, cx_B, cy_A, cy_B),
 *       (cz_A, cz_B, w_A, w_B) -- the operands of packed two-wide instructions; an odd count ends with a
 *       sphere of w = +inf, which no ray reaches
 *   triangle cull : what the two-stage LDS kernel's conservative test reads, computed in double at upload and
 *       rounded (the test carries a margin for that, see closest_hit_lds2).  Per triangle:
 *       c0 (n.x, n.y, n.z, k)       n as in q0 (the reference's normal, same bits), k = a.n
 *       c1 (g0.x, g0.y, g0.z, h0)   g_j . p + h_j = the reference's j-th edge expression (kernel0.cl:298-300)
 *       c2 (g1.x, g1.y, g1.z, h1)   for a point p of the triangle's plane, as a barycentric coordinate (1 at the
 *                                   opposite vertex) times s; the third coordinate is s minus the other two
 *       s                           1 / (error of a coordinate per unit of the ray-dependent factor): in these
 *                                   units every triangle has the same margin; 0 (with g = h = 0) switches the
 *                                   test off for a triangle it cannot be trusted on (slivers)
 *       The records are stored as the test walks them, group of 32 triangles after group (a group's candidates
 *       are one mask word).  Within a group first the pairs of triangles that lie in one plane (bit-identical
 *       c0: the two halves of a wall), which share the plane part of the test, six float4 each:
 *           c0, c1 c2 of the first, c1 c2 of the second, (s first, s second, mask bit first, mask bit second)
 *       then the other triangles, four float4 each:  c0, c1, c2, (s, mask bit, -, -).
 *       In front of those, the pairs that are the two halves of a parallelogram, four float4 each:
 *           c0, u, v, (s, mask bit first, mask bit second, -)
 *       u and v are the first triangle's coordinate planes that vanish on its two edges other than the shared
 *       one; the first triangle's coordinates are u, v, s - u - v, the second one's s - u, s - v, u + v - s.
 *       And in front of those, two such parallelograms in parallel planes that face opposite ways (normals n and
 *       -n) and have the same u, v and s as numbers -- two opposite walls of a box, cut along the same diagonal
 *       -- five float4:
 *           c0 of the first, u, v, (s, -k of the second, -, -), (mask bits: first's halves, second's halves)
 *       A ray can only hit the one whose normal it runs against (kernel0.cl:288), so the test is made for that
 *       one alone: with n and k or -k it is, operation for operation, the test of that parallelogram's own record.
 *       A mask bit is 1 << (triangle index - first index of the group), kept as the bits of a float.
 *   surface  2 x float4 : colour, emissive                kernel0.cl:90-94
 * Surface ids live in separate uint32 arrays (read once per query, for the
 * winning primitive only).
 */
#ifndef BDPT_LAYOUT_H
#define BDPT_LAYOUT_H

#include <stdint.h>
#include <hip/hip_runtime.h>

#define BDPT_TRI_F4 6
#define BDPT_PLN_F4 2
#define BDPT_CULL_MAX_TRIANGLES 256   /* the two-stage kernel's scenes are bounded by LDS well below this */
#define BDPT_CULL_SLAB_F4 5
#define BDPT_CULL_QUAD_F4 4
#define BDPT_CULL_PAIR_F4 6
#define BDPT_CULL_SINGLE_F4 4

struct bdpt_dev_scene {
    const float4 *sph;          /* [n_spheres] */
    const uint32_t *sph_surf;   /* [n_spheres] */
    const float4 *pln;          /* [n_planes * BDPT_PLN_F4] */
    const uint32_t *pln_surf;   /* [n_planes] */
    const float4 *tri;          /* [(n_triangles + 1) * BDPT_TRI_F4] */
    const float4 *tri_bound;    /* [2 * ceil((n_triangles + 1) / 2)] bounding spheres, in pairs */
    const float4 *tri_cull;     /* [cull_f4] records of the conservative test; empty for scenes of more than BDPT_CULL_MAX_TRIANGLES */
    int cull_f4;
    float cull_c;               /* scene constant of that test's margin */
    /* per group of 32 triangles: offset of its records in tri_cull (float4, bits 0-10), pairs of facing
     * parallelograms (11-14), parallelograms (16-20), other coplanar pairs (21-25), other triangles (26-31) */
    uint32_t cull_group[BDPT_CULL_MAX_TRIANGLES / 32];
    const uint32_t *tri_surf;   /* [n_triangles] */
    const float4 *surf;         /* [n_surfaces * 2] */
    int n_spheres, n_planes, n_triangles, n_surfaces;
    /* hierarchy of boxes over the triangles (bdpt_bvh.h) */
    const float4 *bvh_nodes;    /* [inner nodes * BDPT_BVH_NODE_F4] */
    const uint32_t *bvh_leaf_tri;   /* [triangles in the hierarchy] triangle indices in leaf order */
    const uint32_t *bvh_brute;  /* [bvh_n_brute] triangles every ray tests in full, ascending */
    int bvh_root, bvh_n_brute;
    int bvh_stack_limit;        /* entries of the walk's per-lane stack in LDS, at most BDPT_BVH_STACK */
    int bvh_share;              /* the lanes of a wavefront share the work of their walks (pays for large hierarchies) */
    int bvh_stack_total;        /* entries of the private-memory stack of a lane's second walk, at most BDPT_BVH_DEEP */
};

#define BDPT_INV_PRE_N 8

/* launches that sort a wavefront's paths by phase (bdpt_pool.h): a wavefront owns BDPT_POOL_W x BDPT_POOL_ROWS pixels and keeps
 * BDPT_POOL_GLOBAL_F4 float4 of path state per pixel in device memory */
#define BDPT_POOL_W 16
#ifndef BDPT_POOL_ROWS
#define BDPT_POOL_ROWS 6
#endif
#define BDPT_POOL_GLOBAL_F4 4
/* launches that sort a wavefront's pixels by the class of their next sample (bdpt_class_pool.h): BDPT_CPOOL_W x BDPT_CPOOL_ROWS pixels per wavefront */
#ifndef BDPT_CPOOL_W
#define BDPT_CPOOL_W 16
#endif
#ifndef BDPT_CPOOL_ROWS
#define BDPT_CPOOL_ROWS 8
#endif

struct bdpt_kargs {
    bdpt_dev_scene scene;
    /* per-pixel state, row-major y*width + x */
    uint32_t *seed_w;           /* state.w of the reference's uint4 seed */
    float4 *accum;
    int32_t *aov_eye_hit;       /* 2 per pixel */
    uint32_t *aov_hit_hash;
    float4 *aov_radiance;
    uint32_t *aov_rng_draws;
    uint32_t *aov_rays;
    unsigned long long *counters;   /* [5] rays, light-loop samples, eye-only samples, cull tests, exact triangle tests after a cull test */
    int width, height;
    int rx0, ry0, rw, rh;       /* region rendered by this launch */
    /* Interleaved tile split: the region is cut into tiles of tile_w x tile_h pixels, numbered row by row
     * (tiles_x per row), and this launch renders the tiles whose number is tile_phase modulo tile_stride.
     * tile_stride <= 1: the whole region.  tile_w is a multiple of 16 (a workgroup is 8 or 16 pixels wide), tile_h of 8. */
    int tile_w, tile_h, tiles_x, tile_stride, tile_phase;
    /* Sliced launch (see slice_begin in bdpt_kernels.hip): n_chunks > 1 starts n_chunks workgroups per tile, each
     * running chunk_spp samples.  slice_sync[0] is the ticket counter, slice_sync[1] the launch's give-up mark, slice_sync[4 + t] the
     * number of runs of tile t that are complete; all zero when the launch starts.  slice_cursor keeps a pixel's sample cursor between
     * runs.  n_tiles = workgroups of the unsliced grid, wg_x of them per row of the region. */
    uint32_t *slice_sync;
    uint4 *slice_cursor;
    /* slice_abort: one word of host memory the device can write (the context's sticky "a sliced launch gave up" mark);
     * slice_wait_ticks: how long, in ticks of the device's steady counter, a workgroup waits for the run before it */
    uint32_t *slice_abort;
    unsigned long long slice_wait_ticks;
    /* Class order from a table (cursor_fill_masks in bdpt_kernels.hip): word w of class c of pixel p, at
     * ((c - 1) * mask_words + w) * width * height + p, has bit b set when sample 32 w + b of the launch is of class c.
     * Null: the classes are worked out again for every class (cursor_rounds in bdpt_device.h). */
    uint32_t *class_masks;
    int mask_words;
    int n_chunks, chunk_spp, n_tiles, wg_x;
    int block;                  /* threads per workgroup, 64 or 128: one 8x8-pixel tile per wavefront, side by side (set by the launcher) */
    float4 *pool_state;         /* pooled launches: BDPT_POOL_GLOBAL_F4 float4 per pixel slot, one region per run of a tile and wavefront */
    const float *xt_tab, *yt_tab;   /* the normalised screen coordinate of every pixel column and row (kernel0.cl:600-604) */
    int pool_w;                 /* with pool_rows: pixels per wavefront of a pooled launch, pool_w x pool_rows */
    int pool_rows;              /* 0: one lane per pixel.  Else the launch sorts its paths by phase (bdpt_pool.h): a wavefront owns 16 x pool_rows pixels, block / 64 wavefronts side by side */
    int num_samples;            /* imgDesc.numSamples before the first sample */
    int spp;
    uint32_t first_sample;      /* STRIDED: sample_offset + num_samples */
    /* camera (kernel0.cl:22-29) and what shoot_ray derives from it */
    float cam_pos[3], cam_look[3], cam_up[3], cam_right[3];   /* right = cross(up, look) */
    float aspect;               /* (float)W / (float)H, kernel0.cl:603 */
    float delta_x, delta_y;     /* BLUR_FACTOR / W, / H, kernel0.cl:196-197 */
    /* light (kernel0.cl:663-669) */
    float light_c[3], light_r;
    float light_emissive[4];
    int light_origin_type, light_origin_index;
    /* loop constants (kernel0.cl:683-694) */
    int max_l;
    float p_con;
    float pre_factor0;
    float inv_pre[BDPT_INV_PRE_N];   /* 1 / preFactor at light-subpath vertex k: preFactor starts at pre_factor0 and is multiplied by p_con per vertex (:745) */
    float same_point_d2;        /* smallest float whose square root is >= 0.0001f: length(v) < 0.0001f (:714) iff dot(v,v) < same_point_d2 */
    int indirect_only;
    int rng_mode;
    int falloff;
    float falloff_c;
    int accum_mode;
    int sampler;                /* BDPT_SAMPLER_* */
    float imp_light_weight, imp_half_angle;   /* brdf_lambertian_imp's lightWeight and halfAngle */
    /* STRIDED + SUM launches take the samples of a pixel grouped by the length Russian roulette
     * gives their light subpath (see SampleCursor in bdpt_device.h) */
    int reorder;                /* 0 index order, 1 roulette classes, 2 roulette classes for the pixels whose first eye ray reaches a surface */
    uint32_t fate_above;        /* a roulette draw x ends the subpath iff x > fate_above */
};

#endif

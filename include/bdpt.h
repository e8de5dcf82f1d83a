/*
 * bdpt.h -- C-ABI of the MI355X bidirectional path tracer.
 *
 * The reference has no plugin or FFI layer: its device boundary is the literal
 * sequence of OpenCL calls in src/renderer/main.cpp.  Each entry point below
 * replaces one group of those calls (cited per function).  Plain pointers and
 * sizes only; every function returns BDPT_OK (0) or a negative bdpt_status and
 * records a message readable through bdpt_last_error().
 *
 * Threading: a context is bound to one GPU and is not re-entrant; use one
 * context per device (one process per GPU in the multi-GPU drivers).
 *
 * Image layout: every per-pixel buffer that crosses this boundary is row-major,
 * pixel (x, y) at index y*width + x, except the seed buffer, which keeps the
 * reference's column-major index x*height + y (kernel0.cl:590, :861).
 */
#ifndef BDPT_H
#define BDPT_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef enum bdpt_status {
    BDPT_OK = 0,
    BDPT_ERR_INVALID = -1,   /* bad argument */
    BDPT_ERR_HIP = -2,       /* a HIP runtime call failed */
    BDPT_ERR_STATE = -3,     /* call out of order (e.g. render before upload) */
    BDPT_ERR_NO_DEVICE = -4  /* no usable GPU / ordinal out of range */
} bdpt_status;

/* ---- upload structs: byte-for-byte the reference's host structs ---------- */

/* src/include/geometry/primitives.h:6-17 (32 B); kernel view kernel0.cl:32-38 */
typedef struct bdpt_sphere {
    float center[4];
    float radius;
    uint32_t surface;
    uint32_t reserved0, reserved1;
} bdpt_sphere;

/* primitives.h:19-32 (64 B); kernel view kernel0.cl:72-79 (p0,p1,p2 = a,b,c) */
typedef struct bdpt_triangle {
    float p0[4], p1[4], p2[4];
    uint32_t surface;
    uint32_t reserved0, reserved1, reserved2;
} bdpt_triangle;

/* primitives.h:34-46 (48 B); kernel view kernel0.cl:81-86 */
typedef struct bdpt_plane {
    float normal[4];
    float point[4];
    uint32_t surface;
    uint32_t reserved0, reserved1, reserved2;
} bdpt_plane;

/* primitives.h:57-64 (32 B); kernel view kernel0.cl:90-94 */
typedef struct bdpt_surface {
    float color[4];
    float emissive[4];   /* emissive[3] != 0 marks a light (kernel0.cl:645, :797) */
} bdpt_surface;

/* src/include/renderer/camera.h:37-41 (48 B); look_at is a DIRECTION (camera.cpp:47) */
typedef struct bdpt_camera {
    float pos[4];
    float look_at[4];
    float up[4];
} bdpt_camera;

/* ---- parameters that are macros / literals in the reference -------------- */

enum { BDPT_RNG_STREAM = 0, BDPT_RNG_STRIDED = 1 };
enum { BDPT_LIGHT_ORIGIN_SELF = 0, BDPT_LIGHT_ORIGIN_NONE = 1 };
enum { BDPT_FALLOFF_INV_SQ = 0, BDPT_FALLOFF_INV_SQ_PLUS_C = 1 };
enum { BDPT_ACCUM_MEAN = 0, BDPT_ACCUM_SUM = 1 };
/* Order in which one launch takes the samples of a pixel.
 * INDEX:          ascending sample index (the only order STREAM mode and MEAN accumulation allow).
 * ROULETTE_CLASS: STRIDED + SUM only.  The length Russian roulette gives a sample's light subpath
 *                 (its class, 1..max_light_bounces) follows from the sample's start state alone; the
 *                 launch adds the samples of the longest class first, then the next shorter one, ...,
 *                 ascending sample index within a class.  Every sample is the same as in INDEX order;
 *                 only the order of the float additions into the accumulator differs.  Wavefronts then
 *                 trace subpaths of equal length together, which pays when most samples reach the
 *                 light loop (closed scenes) and costs when most eye rays leave the scene.
 * ADAPTIVE:       STRIDED + SUM only.  Per pixel: ROULETTE_CLASS if the eye ray of the launch's first
 *                 sample index reaches a surface that is not a light, i.e. if that sample enters the
 *                 light loop; INDEX if it ends at the eye ray (kernel0.cl:645-656), as then nearly all
 *                 samples of the pixel do and there is nothing to order.
 * AUTO:           ADAPTIVE when it is allowed, else INDEX. */
enum { BDPT_ORDER_AUTO = 0, BDPT_ORDER_INDEX = 1, BDPT_ORDER_ROULETTE_CLASS = 2, BDPT_ORDER_ADAPTIVE = 3 };
/* How a light-subpath vertex picks its bounce direction.  The live code calls brdf_lambertian
 * (kernel0.cl:767); the file holds two more samplers that nothing calls, offered here as they stand:
 * LAMBERT_CONE: brdf_lambertian, kernel0.cl:484-490 -- random_sample_cone about the normal with
 *               half-angle 3.1415f*0.5f (three draws).
 * HEMISPHERE:   random_sample_hemisphere, kernel0.cl:117-153, about the normal: elevation uniform
 *               in angle (three draws).
 * REFLECTIVE:   brdf_reflective, kernel0.cl:492-508: one more draw p4 first; p4 > 0.6f samples a
 *               cone of half-angle 3.1415f*0.02f about the mirror image of the direction the subpath
 *               arrived from (the zero vector at the light, as kernel0.cl:659 initialises it), else
 *               the Lambert cone (four draws).
 * LAMBERT_IMP:  brdf_lambertian_imp, kernel0.cl:512-543: one more draw p4 first; p4 > imp_light_weight samples
 *               a cone of half-angle 0.50f*M_PI about the normal, else a cone of half-angle imp_half_angle
 *               about the direction toward the light (the only call the file holds, commented out at :945,
 *               aims at a fixed point; here it is the centre of the light sphere); four draws.  The density
 *               the function also returns (:527-540) is not used by the live kernel (:772-778). */
enum { BDPT_SAMPLER_LAMBERT_CONE = 0, BDPT_SAMPLER_HEMISPHERE = 1, BDPT_SAMPLER_REFLECTIVE = 2, BDPT_SAMPLER_LAMBERT_IMP = 3 };
/* Which of the two kernels of the reference's file a render call evaluates.
 * BIDIRECTIONAL: bi_directional_path_trace, kernel0.cl:571-864, the live kernel (everything else in this header).
 * PATH_TRACE:    path_trace, kernel0.cl:872-1001, the older unidirectional kernel the file keeps commented out:
 *                up to four rays per sample from the eye, brdf_reflective at every vertex (:943), a surface's
 *                colour or a light's emission times 1/(2 pi) / weight per hit, sky colour (0.3, 0.3, 0.4, 0.3)
 *                on a miss (:979) and (0.3, 0.3, 0.3, 0.3) when the fifth vertex is reached (:957).  A render
 *                call of spp samples is one launch of that kernel with sampleRate = spp: one discarded draw
 *                (:892), then spp samples blended into the running mean (:990-995).  Needs rng_mode STREAM and
 *                accum_mode MEAN; sampler, light and roulette parameters do not apply.  One plain kernel,
 *                whatever kernel_variant says; offered for completeness, not tuned. */
enum { BDPT_INTEGRATOR_BIDIRECTIONAL = 0, BDPT_INTEGRATOR_PATH_TRACE = 1 };
/* Kernel variants; all give identical results.
 * DIRECT: every lane walks the primitive arrays with wave-uniform scalar loads.
 * LDS:    the workgroup keeps the whole scene in LDS (scenes up to 24 KB of records).
 * TILED:  per-triangle bounding spheres are staged through wave-private LDS tiles and the
 *         triangle itself is read only for rays that come near it (any scene size).
 * LDS2:   scene in LDS as for LDS (cull records included, so about 130 triangles at most); every
 *         triangle first gets a cheap conservative test from all lanes together, and each lane
 *         then runs the reference's test only on the one or two triangles it could not rule out.
 * BVH:    a hierarchy of boxes over the same bounding spheres (built at scene upload), walked by
 *         every lane for itself, nearest boxes first: a ray meets the few triangles near its line
 *         instead of all of them.  The reference's in-order walk keeps the nearest hit and among
 *         equally near ones the lowest index; the walk keeps exactly that one (any scene size;
 *         refused for scenes with more than 1024 triangles too thin for a bounding sphere).
 * AUTO:   BVH (TILED for fewer than 512 triangles or if the scene cannot have a hierarchy) when the scene does not fit LDS2 or
 *         when enough of its triangles are small against the scene for the bounding spheres to
 *         pay (estimated at upload), else LDS2 (LDS for a scene without triangles). */
enum { BDPT_KERNEL_AUTO = 0, BDPT_KERNEL_DIRECT = 1, /* 2: a variant that was removed */ BDPT_KERNEL_LDS = 3, BDPT_KERNEL_TILED = 4, BDPT_KERNEL_LDS2 = 5, BDPT_KERNEL_BVH = 6 };

typedef struct bdpt_params {
    int32_t max_light_bounces;  /* maxL, kernel0.cl:683 (5) */
    float p_continue;           /* pCon, kernel0.cl:687 (0.8f) */
    float blur_factor;          /* BLUR_FACTOR, kernel0.cl:13 (2.0f) */
    int32_t indirect_only;      /* INDIRECT_ONLY, kernel0.cl:11, :752-754, :843-847 (0) */
    int32_t light_index;        /* lIndex, kernel0.cl:663 (sphere 0) */
    /* BDPT_RNG_STREAM: the reference's sequential per-pixel LCG, including its
     * behaviour on early-out samples (no seed write-back, raw value written).
     * BDPT_RNG_STRIDED: sample s of a pixel starts at seed * 48271^(32*s), so
     * any rank can render any sample range; early-outs are ordinary samples.
     * A sample draws 5 + 4 per light-subpath vertex numbers, the first of them
     * discarded (kernel0.cl:592): no number is used by two samples while
     * max_light_bounces <= 7 (<= 5 with the extra draw per vertex of the REFLECTIVE and LAMBERT_IMP samplers);
     * bdpt_set_params refuses STRIDED with longer subpaths. */
    int32_t rng_mode;
    /* Which primitive the first light-subpath rays skip.  The reference reads
     * an uninitialised id here (kernel0.cl:547-569, :675, :705); SELF = the
     * emitting sphere (0, light_index), NONE = no primitive is skipped. */
    int32_t light_origin;
    int32_t connection_falloff; /* kernel0.cl:719 is 1/r^2 (INV_SQ) */
    float falloff_c;            /* c of 1/(c + r^2) when INV_SQ_PLUS_C */
    /* MEAN: running mean prev*n/(n+1) + new/(n+1) (kernel0.cl:853-856).
     * SUM: plain sum of samples, for combining partial images across GPUs.
     * In both modes the first sample after a reset (counter 0) overwrites. */
    int32_t accum_mode;
    int32_t aov_enable;         /* maintain the bdpt_read_aov() buffers */
    int32_t kernel_variant;     /* BDPT_KERNEL_* */
    int32_t sample_order;       /* BDPT_ORDER_* */
    int32_t sampler;            /* BDPT_SAMPLER_* */
    float imp_light_weight;     /* lightWeight of brdf_lambertian_imp (0.0f in the call at kernel0.cl:945) */
    float imp_half_angle;       /* halfAngle of brdf_lambertian_imp (0.2 * M_PI there) */
    int32_t integrator;         /* BDPT_INTEGRATOR_* */
} bdpt_params;

typedef struct bdpt_ctx bdpt_ctx;

/* ---- context: replaces cl::Context / devices / cl::CommandQueue
 *      (main.cpp:227-260, :462-466, :522) and the two W x H images (:265-268,
 *      :448-452).  The HIP code object is linked into the library, so there is
 *      no counterpart of the run-time program build (main.cpp:468-484). */
int bdpt_create(int device_ordinal, int width, int height, bdpt_ctx **out);
void bdpt_destroy(bdpt_ctx *ctx);
const char *bdpt_version(void);
/* checkErr (main.cpp:41-46) printed and exited; here the caller decides. */
const char *bdpt_last_error(const bdpt_ctx *ctx);

/* Fills *p with the reference's literals (see the field comments). */
void bdpt_default_params(bdpt_params *p);
int bdpt_set_params(bdpt_ctx *ctx, const bdpt_params *p);
int bdpt_get_params(const bdpt_ctx *ctx, bdpt_params *p);

/* cl::Buffer creation + blocking enqueueWriteBuffer of spheres, planes,
 * triangles, GeometryDescriptor and surfaces (main.cpp:424-446, :534-539).
 * The arrays are copied and re-laid out before return; null pointers are
 * allowed for zero counts.  Resets the sample counter.
 * The fourth component of every centre, vertex, normal and point must be 0, as in the reference's own scene
 * (main.cpp:282-394): the reference's dot / distance / length are four-component operations (kernel0.cl:331,
 * :336, :398), so any other w changes its image in a way this library does not follow; such a scene is refused
 * (BDPT_ERR_INVALID).  Colours and emissive values use all four components. */
int bdpt_upload_scene(bdpt_ctx *ctx,
                      const bdpt_sphere *spheres, int n_spheres,
                      const bdpt_plane *planes, int n_planes,
                      const bdpt_triangle *triangles, int n_triangles,
                      const bdpt_surface *surfaces, int n_surfaces);

/* Seed upload (main.cpp:456-460, :538): width*height uint4, index x*height+y;
 * only .w drives the generator (kernel0.cl:106-109). */
int bdpt_upload_seeds(bdpt_ctx *ctx, const uint32_t *seeds);
int bdpt_read_seeds(bdpt_ctx *ctx, uint32_t *seeds);

/* Camera upload; like the reference it also zeroes the sample counter
 * (main.cpp:150-156, :524-533).  pos[3], look_at[3] and up[3] must be 0 (camera.cpp:137-139), else BDPT_ERR_INVALID. */
int bdpt_set_camera(bdpt_ctx *ctx, const bdpt_camera *cam);

/* ImageDescriptor write + kernel launch + wait + numSamples += spp
 * (main.cpp:108-131), for `spp` consecutive samples per pixel in one launch.
 * The accumulator is updated in place (the reference's out->in blit,
 * main.cpp:133-138).  bdpt_render waits; bdpt_render_async only enqueues. */
int bdpt_render(bdpt_ctx *ctx, int spp);
int bdpt_render_async(bdpt_ctx *ctx, int spp);
int bdpt_sync(bdpt_ctx *ctx);
int bdpt_num_samples(const bdpt_ctx *ctx);
/* Sets the sample counter (imgDesc.numSamples, main.cpp:152, :546). */
int bdpt_set_num_samples(bdpt_ctx *ctx, int n);

/* Read-back of the accumulated image, width*height RGBA float, row-major
 * (stands in for the present blit, main.cpp:140-145). */
int bdpt_read_accum(bdpt_ctx *ctx, float *rgba);
/* The reverse: replaces the accumulated image (the reference keeps it only in a GL texture and
 * cannot do this).  With bdpt_read_seeds / bdpt_upload_seeds and bdpt_set_num_samples it lets a
 * host stop a progressive render and resume it later with identical results. */
int bdpt_write_accum(bdpt_ctx *ctx, const float *rgba);

/* ---- additions with no reference counterpart ----------------------------- */

/* Multi-GPU partitioning.  Region: only pixels inside the rectangle are
 * rendered (image-tile split).  Sample offset: in STRIDED mode the first
 * sample index this context renders after a camera reset (spp split). */
int bdpt_set_region(bdpt_ctx *ctx, int x0, int y0, int w, int h);
int bdpt_set_sample_offset(bdpt_ctx *ctx, int first_sample);
/* Interleaved tile split, for balance: the region is cut into tiles of tile_w x tile_h pixels (multiples of
 * 16 and 8; edge tiles are clipped), numbered row by row from the region's top-left corner, and render calls
 * touch only the tiles whose number is `phase` modulo `stride` (rank r of N passes stride N, phase r).  Eye
 * rays that leave the scene end a path after one ray (kernel0.cl:652-656), so contiguous bands of a frame
 * cost very different amounts; tiles dealt round-robin even that out.  stride 1, phase 0: the whole region. */
int bdpt_set_tile_interleave(bdpt_ctx *ctx, int tile_w, int tile_h, int stride, int phase);
/* How a render call of many samples is run: as `chunks` runs of spp/chunks samples per workgroup tile (8x8 or 16x8 pixels), each run
 * a workgroup of its own that picks up the tile's pixel state where the run before left it, so that the launch
 * does not end with the GPU draining behind a few long workgroups.  The result is bit-identical for any value.
 * 0 (the default): chosen per launch (up to 16 runs, 32 for BDPT_PATHS_BY_CLASS launches, their length set by how long a sample of the scene took on the last call; one sample per run for scenes of many
 * triangles); 1: one run. */
int bdpt_set_launch_chunks(bdpt_ctx *ctx, int chunks);
/* runs per tile the last render call asked its launch for (1 = unsliced) */
int bdpt_last_launch_chunks(const bdpt_ctx *ctx);
/* How a render call puts paths on lanes; the result is bit-identical either way.
 * PER_LANE: a lane owns one pixel for the whole launch and walks each sample of it from the eye ray to the end of its
 *           light subpath; lanes whose subpath ends early (roulette, kernel0.cl:788; a bounce ray that leaves the scene
 *           or hits a light, :796-813) wait for the longest subpath of their wavefront.
 * POOLED:   a wavefront owns more pixels than it has lanes (16 x 6), keeps the state of their paths in LDS, and every
 *           round runs one phase (eye ray / connection ray / bounce ray, each with what follows it) for up to 64 of
 *           the paths that wait in that phase, picked by ballot and prefix count: terminated paths cost no lane time.
 *           A pixel still has one sample in flight at a time, in the launch's order, so generator streams, running
 *           means and sums are untouched.  For the kernel variants that keep the scene in LDS and the reference's live
 *           sampler; other launches run PER_LANE whatever is asked (bdpt_last_launch_path_scheduling tells).
 * BY_CLASS: a wavefront owns 16 x 8 pixels and every round runs ONE WHOLE SAMPLE for up to 64 of them, chosen so that
 *           their light subpaths end at about the same vertex: how many vertices roulette allows a sample (:787-788)
 *           follows from the generator word the sample starts with, so the pixels are sorted by the class of their next
 *           sample and the round takes the window of classes that wastes the fewest lane-steps.  A pixel's samples
 *           are still taken one after the other in index order, which is what the reference's own mode (sequential
 *           generator, running mean) requires; nothing is allocated.  For launches in index order (BDPT_ORDER_INDEX, or
 *           any mode that does not allow another order) of the small-scene kernel variants with at most five light-
 *           subpath vertices and no interleaved tile split; other launches run PER_LANE whatever is asked.  The work
 *           counters of rays and primitive tests (bdpt_read_counters) of a sliced BDPT_RNG_STREAM launch are lower than
 *           a PER_LANE launch's: a pixel whose eye ray ends the path (:645-656) is traced once per launch, not once per
 *           run.
 *           1080p Cornell box, 1024 samples, sequential generator and running mean: 118 ms against 134 ms PER_LANE.
 * AUTO:     BY_CLASS for the launches that allow it from 32 samples per pixel on, else PER_LANE.
 * No reference counterpart. */
enum { BDPT_PATHS_AUTO = 0, BDPT_PATHS_PER_LANE = 1, BDPT_PATHS_POOLED = 2, BDPT_PATHS_BY_CLASS = 3 };
int bdpt_set_path_scheduling(bdpt_ctx *ctx, int mode);
int bdpt_last_launch_path_scheduling(const bdpt_ctx *ctx);
/* A run of a sliced launch waits for the run before it on the same tile.  That wait always ends unless something
 * else has gone wrong on the device; as a guard, a workgroup that has waited longer than a limit gives up, the rest of
 * the launch drains without touching the image, and the next call that waits for the launch (bdpt_sync, bdpt_render,
 * bdpt_read_accum / _aov / _seeds / _counters, bdpt_last_launch_ms) or the next bdpt_render_async returns
 * BDPT_ERR_HIP once: the image is incomplete, reset with bdpt_set_camera and render again.  seconds = 0 (the
 * default): the limit is a hundred times the launch's expected duration, at least a minute.  No reference
 * counterpart. */
int bdpt_set_slice_wait_limit(bdpt_ctx *ctx, double seconds);

/* Interop with a caller that owns device memory and streams (torch):
 * render on `hip_stream` (NULL = the context's own stream), and accumulate into
 * caller-owned device memory of width*height*4 floats (NULL = internal). */
int bdpt_set_stream(bdpt_ctx *ctx, void *hip_stream);
/* Render on HIP's default stream (handle 0, which bdpt_set_stream reads as "the context's own"): what a
 * caller whose own work runs on the default stream (torch without a stream context) selects, so that
 * launches, collectives and copies are ordered by one stream instead of by the default stream's implicit
 * synchronisation with blocking streams. */
int bdpt_use_default_stream(bdpt_ctx *ctx);
int bdpt_bind_accum(bdpt_ctx *ctx, void *device_rgba);
void *bdpt_accum_device_ptr(bdpt_ctx *ctx);

/* Parity / accounting outputs, maintained when params.aov_enable != 0 and
 * cleared by bdpt_set_camera / bdpt_upload_scene / bdpt_upload_seeds. */
enum {
    BDPT_AOV_EYE_HIT = 0,   /* int32[2] per pixel: (type, index) of the last sample's eye hit, (-1,-1) on a miss */
    BDPT_AOV_HIT_HASH = 1,  /* uint32 per pixel: h = (h ^ id) * 16777619 + 0x9E3779B9 over every closest-hit query, id = type<<28 | index, 0xFFFFFFFF on a miss */
    BDPT_AOV_RADIANCE = 2,  /* float[4] per pixel: the last sample's value before accumulation */
    BDPT_AOV_RNG_DRAWS = 3, /* uint32 per pixel: generator draws since reset */
    BDPT_AOV_RAYS = 4       /* uint32 per pixel: closest-hit queries since reset */
};
int bdpt_read_aov(bdpt_ctx *ctx, int kind, void *out);

/* Work counters of the most recent render call that was made with counting on:
 * out[0] closest-hit queries traced, out[1] samples that reached the light loop,
 * out[2] samples that ended at the eye ray; for the kernel variants that put a
 * cheap conservative test in front of the triangle test, out[3] (ray, triangle)
 * pairs given that cheap test and out[4] pairs that then went through the
 * reference's triangle test (both 0 for the variants that test every pair in full).
 * A launch without parity buffers (aov_enable = 0) leaves out the rays whose
 * result nothing reads, and out[0] counts what it traced; with aov_enable = 1
 * every ray of the reference's formulation is traced and counted.  Counting adds
 * device atomics, so it is off unless enabled here. */
#define BDPT_N_COUNTERS 5
int bdpt_enable_counters(bdpt_ctx *ctx, int on);
int bdpt_read_counters(bdpt_ctx *ctx, uint64_t out[BDPT_N_COUNTERS]);

/* Closest hit (ray_intersect, kernel0.cl:365-478) of n caller-supplied rays
 * against the uploaded scene, on the GPU.  rays: 6 floats per ray (origin xyz,
 * direction xyz); origin_ids: 2 ints per ray, the (type, index) of the
 * primitive the ray leaves and must skip, (4, 0) for none.  out_ids: 2 ints per
 * ray, (-1,-1) on a miss; out_pos / out_normal: 3 floats per ray.  Host
 * pointers; returns after the results are copied back. */
int bdpt_query_rays(bdpt_ctx *ctx, const float *rays, const int32_t *origin_ids, int n,
                    int32_t *out_ids, float *out_pos, float *out_normal);

/* bdpt_query_rays for callers that must not allocate device memory (tight serving loops, harnesses that
 * forbid hipMalloc, processes whose memory pool is fully committed).  It has no counterpart in the
 * reference.  Same arguments and results as bdpt_query_rays, which allocates and frees five device buffers
 * per call; this one stages rays and results in a fixed array that is part of the library's code object
 * (4 MiB on every GPU the library is loaded on, held until the process ends) and calls no hipMalloc /
 * hipFree.  It allocates nothing, but it is not capturable: it copies from and to the caller's host buffers
 * and waits for the stream before it returns, neither of which a stream under HIP graph capture allows.  n may be at most BDPT_QUERY_NOALLOC_MAX_RAYS; more is BDPT_ERR_INVALID.  All contexts on one
 * GPU share that array: calls that overlap in time on one GPU (from several threads, or several contexts
 * on different streams) race and must be serialised by the caller. */
#define BDPT_QUERY_NOALLOC_MAX_RAYS 65536
int bdpt_query_rays_noalloc(bdpt_ctx *ctx, const float *rays, const int32_t *origin_ids, int n,
                            int32_t *out_ids, float *out_pos, float *out_normal);

/* Evaluates one built-in of the numeric contract (include/bdpt_clmath.h) on the
 * GPU for n inputs of 3 doubles each: op 0 sin, 1 cos, 2 acos(float),
 * 3 acos(double), 4 sqrt, 5 a/b, 6 fma(a,b,c), 7 x of normalize(a,b,c).
 * For checking that device and host arithmetic agree bit for bit. */
int bdpt_clmath_eval(bdpt_ctx *ctx, int op, const double *in, double *out, int n);

/* Limit, in bytes, of the device memory the context may take to keep the order of a launch's samples (class
 * order, bdpt_params.sample_order): width x height x max_light_bounces x ceil(spp / 32) words, allocated by the
 * first launch that needs them and kept until the context is destroyed (1.3 GB for 1920x1080 x 1024 samples,
 * 5.3 GB for 3840x2160 x 1024).  That first bdpt_render_async waits for the stream and allocates before it
 * enqueues.  A launch that would need more than the limit, or more than half of the device memory that is free at
 * that moment, works the order out as it goes: slower (9 % for 1024 samples of the Cornell box), same result;
 * bdpt_last_launch_order_table tells which.  Default 8 GiB; 0 = never.  No reference counterpart. */
int bdpt_set_order_scratch_limit(bdpt_ctx *ctx, size_t bytes);
/* 1 if the last launch took its class order from that table, 0 if it worked the order out as it went (no table
 * wanted, over the limit, or the memory was not to be had). */
int bdpt_last_launch_order_table(const bdpt_ctx *ctx);
/* BDPT_KERNEL_* that the last launch ran (what AUTO resolved to), 0 before the first launch. */
int bdpt_last_launch_variant(const bdpt_ctx *ctx);
/* The hierarchy bdpt_upload_scene built over the scene's triangles (BDPT_KERNEL_BVH): info[0] = inner nodes (four
 * children each), info[1] = levels of the binary tree it was folded from, info[2] = entries of the walk's per-lane stack a
 * launch reserves in LDS (256 bytes per wavefront each; deeper walks continue in private memory), info[3] = triangles kept outside the hierarchy, which every ray
 * tests in full (unbounded and oversized ones).  Returns 1, or 0 when the scene has no hierarchy (none uploaded, or more
 * than 1024 triangles that nothing bounds).  No reference counterpart. */
int bdpt_hierarchy_info(const bdpt_ctx *ctx, int info[4]);
/* How the uploaded scene's triangles are grouped for the two-stage kernel's conservative test: out[0] pairs that
 * are the two halves of a parallelogram, out[1] other pairs in one plane, out[2] triangles on their own, out[3]
 * pairs of parallelograms that are opposite walls of a box (four triangles each, not counted in out[0]); all 0
 * for scenes that kernel does not take.  Diagnostic; no reference counterpart. */
int bdpt_two_stage_record_counts(const bdpt_ctx *ctx, int out[4]);
/* Device time of the most recent render launch in milliseconds (HIP events on
 * the stream the kernel ran on); waits for that launch. */
int bdpt_last_launch_ms(bdpt_ctx *ctx, float *ms);

/* ---- several GPUs of one node behind one handle ---------------------------
 * The reference drives one device (main.cpp:242-260, :522).  A group holds one context per GPU, in one
 * process, driven by the calling thread: uploads and camera changes go to every member, a render call
 * deals the samples of each pixel to the members (rng_mode STRIDED + accum_mode SUM, set by the group and
 * required of bdpt_group_set_params), and bdpt_group_read_image sums the members' partial images on member
 * 0 -- one RCCL reduce over xGMI (BDPT_COMBINE_RCCL; the library is looked up at run time), or
 * device-to-device copies and an add kernel (BDPT_COMBINE_COPY; also what a group that names one GPU
 * twice uses) -- and returns the mean.  BDPT_COMBINE_AUTO takes RCCL when it loads.  A render call of spp
 * samples deals the sample indices [N, N + spp) after the N rendered since the last camera change to the
 * members in contiguous ranges, so the image depends on spp and on the number of GPUs only through the order of
 * float additions; the left-over samples of a call (spp not a multiple of the group's size, or one sample per
 * frame) go to the members in turn, so no GPU idles frame after frame.  A render call that fails part-way
 * leaves the group unusable (BDPT_ERR_STATE) until bdpt_group_set_camera or bdpt_group_upload_scene resets it.
 * Errors: as for contexts, message through bdpt_group_last_error. */
typedef struct bdpt_group bdpt_group;
enum { BDPT_COMBINE_AUTO = 0, BDPT_COMBINE_RCCL = 1, BDPT_COMBINE_COPY = 2 };
int bdpt_group_create(const int *device_ordinals, int n_devices, int width, int height, int combine, bdpt_group **out);
void bdpt_group_destroy(bdpt_group *g);
const char *bdpt_group_last_error(const bdpt_group *g);
int bdpt_group_size(const bdpt_group *g);
bdpt_ctx *bdpt_group_member(bdpt_group *g, int i);      /* for queries (bdpt_last_launch_ms, ...); owned by the group */
int bdpt_group_combine(const bdpt_group *g);            /* BDPT_COMBINE_RCCL or BDPT_COMBINE_COPY: what read_image uses */
int bdpt_group_set_params(bdpt_group *g, const bdpt_params *p);
int bdpt_group_upload_scene(bdpt_group *g, const bdpt_sphere *spheres, int n_spheres, const bdpt_plane *planes, int n_planes,
                            const bdpt_triangle *triangles, int n_triangles, const bdpt_surface *surfaces, int n_surfaces);
int bdpt_group_upload_seeds(bdpt_group *g, const uint32_t *seeds);
int bdpt_group_set_camera(bdpt_group *g, const bdpt_camera *cam);       /* also zeroes the sample counters */
int bdpt_group_render(bdpt_group *g, int spp);                          /* spp more samples per pixel, all members together; waits */
int bdpt_group_num_samples(const bdpt_group *g);
int bdpt_group_read_image(bdpt_group *g, float *rgba);                  /* mean image, width*height RGBA float, row-major */

#ifdef __cplusplus
}
#endif
#endif /* BDPT_H */

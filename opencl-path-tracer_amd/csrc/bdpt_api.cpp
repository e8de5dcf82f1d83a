/*
 * bdpt_api.cpp -- the C-ABI of include/bdpt.h over the HIP runtime.
 *
 * Replaces the OpenCL buffer / queue / kernel plumbing that the reference does
 * inline in src/renderer/main.cpp (:227-260 context, :424-452 buffers, :489-509
 * kernel arguments, :533-539 uploads, :100-172 per-frame launch).  There is no
 * CPU fallback: without a GPU every entry point that needs one fails.
 *
 * Built with -ffp-contract=off: the scene pre-processing below evaluates the
 * ray-independent parts of the reference's primitive tests on the host with the
 * numeric contract, and the device consumes the results as they are.
 */
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdlib>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>

#include "../../include/bdpt.h"
#include "../../include/bdpt_clmath.h"
#include "bdpt_ctx_internal.h"
#include "bdpt_kernels.h"
#include "bdpt_bvh.h"
#include "bdpt_layout.h"

/* memory a context may take for the class order of a launch unless told otherwise (bdpt_set_order_scratch_limit) */
#define BDPT_ORDER_SCRATCH_DEFAULT ((size_t)8 << 30)

struct bdpt_ctx {
    int device = 0;
    int width = 0, height = 0;
    hipStream_t own_stream = nullptr;
    hipStream_t stream = nullptr;
    hipEvent_t ev_begin = nullptr, ev_end = nullptr;
    bool have_launch = false;

    bdpt_params params{};
    bdpt_camera camera{};
    bool have_camera = false, have_scene = false, have_seeds = false;
    int num_samples = 0;
    int sample_offset = 0;
    int rx0 = 0, ry0 = 0, rw = 0, rh = 0;
    int tile_w = 0, tile_h = 0, tile_stride = 1, tile_phase = 0;
    bool counters_on = false;
    bool auto_tiled = false;            /* AUTO prefers the tiled kernel for this scene even if it fits LDS */
    int last_variant = 0;
    int last_chunks = 1;

    /* host copies needed after upload */
    std::vector<bdpt_sphere> spheres;
    std::vector<bdpt_surface> surfaces;
    std::vector<uint32_t> seeds;       /* the caller's uint4 seeds, reference layout */
    int n_planes = 0, n_triangles = 0;

    /* device memory */
    float4 *d_sph = nullptr, *d_pln = nullptr, *d_tri = nullptr, *d_tri_bound = nullptr, *d_tri_cull = nullptr, *d_surf = nullptr;
    float cull_c = 0.0f;
    int cull_f4 = 0;
    uint32_t cull_group[BDPT_CULL_MAX_TRIANGLES / 32] = {};
    uint32_t *d_sph_surf = nullptr, *d_pln_surf = nullptr, *d_tri_surf = nullptr;
    /* hierarchy over the triangles' bounding spheres (bdpt_bvh.h), built by every scene upload that has triangles */
    float4 *d_bvh_nodes = nullptr;
    uint32_t *d_bvh_leaf_tri = nullptr, *d_bvh_brute = nullptr;
    int bvh_root = BDPT_BVH_NONE, bvh_n_brute = 0, bvh_depth = 0, bvh_nodes = 0;
    int bvh_stack_limit = 1;            /* entries of the walk's stack a launch reserves in LDS per lane: what the tree can need, at most BDPT_BVH_STACK */
    int bvh_stack_total = 1;            /* entries of the private-memory stack of a lane's second walk */
    int bvh_lds_cap = BDPT_BVH_STACK, bvh_stack_cap = BDPT_BVH_DEEP;
    int bvh_share_min = BDPT_BVH_SHARE_MIN_TRIANGLES;    /* BDPT_BVH_SHARE_MIN in the environment changes it (testing aid) */   /* BDPT_BVH_STACK_LIMIT in the environment lowers it (testing aid) */
    bool have_bvh = false;              /* false: the scene cannot have one (bdpt_bvh_build), BDPT_KERNEL_BVH is refused */
    uint32_t *d_seed_w = nullptr;
    float4 *d_accum_own = nullptr;
    float4 *d_accum = nullptr;
    int32_t *d_aov_eye = nullptr;
    uint32_t *d_aov_hash = nullptr, *d_aov_draws = nullptr, *d_aov_rays = nullptr;
    float4 *d_aov_rad = nullptr;
    unsigned long long *d_counters = nullptr;
    uint32_t *d_slice_sync = nullptr;  /* sliced launches: ticket counter + one word per workgroup tile */
    uint4 *d_slice_cursor = nullptr;   /* ... and a sample cursor per pixel */
    float *d_xt_tab = nullptr, *d_yt_tab = nullptr;   /* screen coordinate of every pixel column / row (kernel0.cl:600-604) */
    float4 *d_pool_state = nullptr;    /* pooled launches: path state that waits in device memory (bdpt_kargs.pool_state), allocated by the first launch that wants it */
    size_t pool_state_bytes = 0;
    uint32_t *d_class_masks = nullptr; /* class order from a table (bdpt_kargs.class_masks), allocated by the first launch that wants it */
    size_t class_mask_bytes = 0;       /* its size */
    size_t order_scratch_limit = BDPT_ORDER_SCRATCH_DEFAULT;   /* bdpt_set_order_scratch_limit */
    int slice_tiles = 0;               /* tiles d_slice_sync has room for */
    int launch_chunks = 0;             /* bdpt_set_launch_chunks: 0 = chosen per launch */
    /* One word of pinned host memory that a workgroup of a sliced launch sets when it gives up waiting for the run
     * before it (slice_begin); sticky across launches until the host has reported it.  Read directly, after any wait
     * for the stream, so every way of waiting sees it. */
    uint32_t *h_slice_abort = nullptr, *d_slice_abort = nullptr;
    double slice_wait_s = 0.0;         /* bdpt_set_slice_wait_limit: 0 = derived from the expected launch time */
    double steady_ticks_per_ms = 1.0e5; /* the device's steady counter (wall_clock64): 100 MHz unless the device says otherwise */
    int path_scheduling = BDPT_PATHS_AUTO;   /* bdpt_set_path_scheduling */
    int last_path_scheduling = 0;      /* BDPT_PATHS_PER_LANE, _POOLED or _BY_CLASS: how the last launch ran */
    int last_order_table = 0;          /* the last launch took its class order from the table (1) or worked it out as it went (0) */
    double ms_per_pixel_sample = 0.0;  /* measured on an earlier launch of this scene and these parameters; 0 = not yet */
    double last_launch_work = 0.0;     /* pixels x samples of the last launch, until its time has been read */

    std::string err;
};

/* automatic slicing: up to this many runs per launch, of at least this many samples for the small-scene kernels */
#ifndef BDPT_AUTO_CHUNKS
#define BDPT_AUTO_CHUNKS 16
#endif
#ifndef BDPT_AUTO_CHUNK_MIN_SPP
#define BDPT_AUTO_CHUNK_MIN_SPP 32
#endif
/* pooled launches (bdpt_pool.h): up to this many runs, of at least this many samples */
#ifndef BDPT_POOL_AUTO_CHUNKS
#define BDPT_POOL_AUTO_CHUNKS 8
#endif
#ifndef BDPT_POOL_AUTO_CHUNK_MIN_SPP
#define BDPT_POOL_AUTO_CHUNK_MIN_SPP 128
#endif
/* AUTO walks the hierarchy from this many triangles on; below, meeting every bounding sphere is faster (soups of 150 / 300 /
 * 1000 / 3000 triangles in the Cornell box, 64 samples at 1080p: exhaustive 39 / 54 / 129 / 344 ms, hierarchy 55 / 66 / 95 / 140) */
#ifndef BDPT_BVH_AUTO_MIN_TRIANGLES
#define BDPT_BVH_AUTO_MIN_TRIANGLES 512
#endif
/* launches in index order sort a wavefront's pixels by the length of their next sample (bdpt_class_pool.h) from this
 * many samples per pixel on, unless the caller chose (bdpt_set_path_scheduling) */
#ifndef BDPT_CPOOL_AUTO_MIN_SPP
#define BDPT_CPOOL_AUTO_MIN_SPP 32
#endif
/* what one more run costs per pixel (its state re-read and re-written): 1.06 ms per run of a 3840x2160 frame */
#define BDPT_RUN_COST_MS_PER_PIXEL 1.28e-7

static_assert(BDPT_QUERY_NOALLOC_MAX_RAYS == BDPT_QUERY_STATIC_MAX, "the documented limit is the size of the staging array");

static thread_local std::string g_null_ctx_error;

static int fail(bdpt_ctx *ctx, int code, const std::string &msg)
{
    if (ctx) ctx->err = msg;
    else g_null_ctx_error = msg;
    return code;
}

#define HIP_TRY(ctx, call)                                                                      \
    do {                                                                                        \
        hipError_t e_ = (call);                                                                 \
        if (e_ != hipSuccess)                                                                   \
            return fail(ctx, BDPT_ERR_HIP, std::string(#call) + ": " + hipGetErrorString(e_));  \
    } while (0)

template <class T>
static void dev_free(T *&p)
{
    if (p) (void)hipFree(p);
    p = nullptr;
}

template <class T>
static hipError_t dev_upload(T *&dst, const std::vector<T> &src)
{
    dev_free(dst);
    /* keep a valid pointer even for empty arrays: kernels never dereference it then */
    size_t bytes = (src.empty() ? 1 : src.size()) * sizeof(T);
    hipError_t e = hipMalloc((void **)&dst, bytes);
    if (e != hipSuccess) return e;
    if (!src.empty()) e = hipMemcpy(dst, src.data(), src.size() * sizeof(T), hipMemcpyHostToDevice);
    return e;
}

static clm_f3 v3(const float *p) { return clm_make3(p[0], p[1], p[2]); }

/* After the host has waited for the context's stream (or for the end event of its last launch): did a workgroup of a
 * sliced launch give up (slice_begin in bdpt_kernels.hip)?  Its tile and every later run of the launch were left
 * undone, so the image is incomplete.  Reported once; the mark is then cleared so that the caller can reset
 * (bdpt_set_camera) and render again. */
static int check_sliced(bdpt_ctx *ctx, const char *who)
{
    if (!ctx->h_slice_abort || *(volatile uint32_t *)ctx->h_slice_abort == 0u) return BDPT_OK;
    *(volatile uint32_t *)ctx->h_slice_abort = 0u;
    return fail(ctx, BDPT_ERR_HIP, std::string(who) + ": a sliced launch did not complete in order (a workgroup waited longer than the limit "
                                   "for the run before it and the launch was abandoned); the image, the seeds and the parity buffers are "
                                   "incomplete -- reset with bdpt_set_camera and render again (bdpt_set_launch_chunks(ctx, 1) renders unsliced)");
}

int bdpt_ctx_device(const bdpt_ctx *ctx) { return ctx->device; }
hipStream_t bdpt_ctx_stream(const bdpt_ctx *ctx) { return ctx->stream; }

extern "C" {

const char *bdpt_version(void) { return "bdpt-amd 0.1 (gfx950)"; }

const char *bdpt_last_error(const bdpt_ctx *ctx) { return ctx ? ctx->err.c_str() : g_null_ctx_error.c_str(); }

void bdpt_default_params(bdpt_params *p)
{
    if (!p) return;
    std::memset(p, 0, sizeof *p);
    p->max_light_bounces = 5;
    p->p_continue = 0.8f;
    p->blur_factor = 2.0f;
    p->indirect_only = 0;
    p->light_index = 0;
    p->rng_mode = BDPT_RNG_STREAM;
    p->light_origin = BDPT_LIGHT_ORIGIN_SELF;
    p->connection_falloff = BDPT_FALLOFF_INV_SQ;
    p->falloff_c = 1.0f;
    p->accum_mode = BDPT_ACCUM_MEAN;
    p->aov_enable = 0;
    p->kernel_variant = BDPT_KERNEL_AUTO;
    p->sample_order = BDPT_ORDER_AUTO;
    p->sampler = BDPT_SAMPLER_LAMBERT_CONE;
    p->imp_light_weight = 0.0f;
    p->imp_half_angle = (float)(0.2 * 3.14159265358979323846);
    p->integrator = BDPT_INTEGRATOR_BIDIRECTIONAL;
}

int bdpt_create(int device_ordinal, int width, int height, bdpt_ctx **out)
{
    if (!out) return fail(nullptr, BDPT_ERR_INVALID, "bdpt_create: out is null");
    *out = nullptr;
    if (width <= 0 || height <= 0 || (long long)width * height > (1LL << 28))
        return fail(nullptr, BDPT_ERR_INVALID, "bdpt_create: bad image size");
    int n_dev = 0;
    hipError_t e = hipGetDeviceCount(&n_dev);
    if (e != hipSuccess || n_dev <= 0)
        return fail(nullptr, BDPT_ERR_NO_DEVICE, "bdpt_create: no HIP device available (this library has no CPU path)");
    if (device_ordinal < 0 || device_ordinal >= n_dev)
        return fail(nullptr, BDPT_ERR_NO_DEVICE, "bdpt_create: device ordinal out of range");

    bdpt_ctx *ctx = new bdpt_ctx();
    ctx->device = device_ordinal;
    ctx->width = width;
    ctx->height = height;
    ctx->rw = width;
    ctx->rh = height;
    bdpt_default_params(&ctx->params);
    /* BDPT_LAUNCH_CHUNKS in the environment: the initial bdpt_set_launch_chunks of every context (to run an existing
     * program or test suite with sliced launches throughout) */
    if (const char *env = std::getenv("BDPT_LAUNCH_CHUNKS")) {
        const long v = std::strtol(env, nullptr, 10);
        if (v >= 0 && v <= 4096) ctx->launch_chunks = (int)v;
    }
    /* BDPT_PATH_SCHEDULING in the environment: the initial bdpt_set_path_scheduling of every context (to run an existing
     * program or test suite with the pooled launch form wherever it applies) */
    if (const char *env = std::getenv("BDPT_PATH_SCHEDULING")) {
        const long v = std::strtol(env, nullptr, 10);
        if (v >= BDPT_PATHS_AUTO && v <= BDPT_PATHS_BY_CLASS) ctx->path_scheduling = (int)v;
    }
    /* BDPT_BVH_STACK_LIMIT=a or a,b in the environment, for every context: the hierarchy walk's stack has at most `a`
     * entries in LDS and at most `b` (default a) in the private memory of a lane's second walk.  A testing aid: with
     * "2" nearly every walk runs over twice and finishes without the hierarchy, with "2,64" nearly every walk is
     * walked a second time. */
    if (const char *env = std::getenv("BDPT_BVH_STACK_LIMIT")) {
        char *end = nullptr;
        const long a = std::strtol(env, &end, 10);
        const long b = end && *end == ',' ? std::strtol(end + 1, nullptr, 10) : a;
        if (a >= 1 && a <= BDPT_BVH_STACK && b >= a && b <= BDPT_BVH_DEEP) { ctx->bvh_lds_cap = (int)a; ctx->bvh_stack_cap = (int)b; }
    }
    /* BDPT_BVH_SHARE_MIN=n in the environment, for every context: hierarchies of n triangles or more are walked with the
     * lanes of a wavefront sharing work (default 8192; a testing aid: 0 shares always, a large number never) */
    if (const char *env = std::getenv("BDPT_BVH_SHARE_MIN")) {
        const long v = std::strtol(env, nullptr, 10);
        if (v >= 0 && v <= 0x7FFFFFFF) ctx->bvh_share_min = (int)v;
    }
    const size_t npix = (size_t)width * height;
    auto bail = [&](hipError_t err, const char *what) {
        std::string msg = std::string("bdpt_create: ") + what + ": " + hipGetErrorString(err);
        bdpt_destroy(ctx);
        return fail(nullptr, BDPT_ERR_HIP, msg);
    };
    if ((e = hipSetDevice(device_ordinal)) != hipSuccess) return bail(e, "hipSetDevice");
    if ((e = hipStreamCreate(&ctx->own_stream)) != hipSuccess) return bail(e, "hipStreamCreate");
    ctx->stream = ctx->own_stream;
    if ((e = hipEventCreate(&ctx->ev_begin)) != hipSuccess) return bail(e, "hipEventCreate");
    if ((e = hipEventCreate(&ctx->ev_end)) != hipSuccess) return bail(e, "hipEventCreate");
    if ((e = hipMalloc((void **)&ctx->d_accum_own, npix * sizeof(float4))) != hipSuccess) return bail(e, "hipMalloc accum");
    if ((e = hipMemset(ctx->d_accum_own, 0, npix * sizeof(float4))) != hipSuccess) return bail(e, "hipMemset accum");
    ctx->d_accum = ctx->d_accum_own;
    if ((e = hipMalloc((void **)&ctx->d_seed_w, npix * sizeof(uint32_t))) != hipSuccess) return bail(e, "hipMalloc seeds");
    if ((e = hipMalloc((void **)&ctx->d_counters, BDPT_N_COUNTERS * sizeof(unsigned long long))) != hipSuccess) return bail(e, "hipMalloc counters");
    if ((e = hipMemset(ctx->d_counters, 0, BDPT_N_COUNTERS * sizeof(unsigned long long))) != hipSuccess) return bail(e, "hipMemset counters");
    /* room for the tiles of the whole frame plus one row and one column of split tiles that stick out of it */
    ctx->slice_tiles = ((width + 7) / 8 + 64) * ((height + 7) / 8 + 64);           /* tiles of 8x8 pixels at the smallest */
    const size_t slice_bytes = (size_t)(4 + ctx->slice_tiles) * sizeof(uint32_t) + 256;   /* header of 4 words; padded for the launcher's rounded clear */
    if ((e = hipMalloc((void **)&ctx->d_slice_sync, slice_bytes)) != hipSuccess) return bail(e, "hipMalloc slice words");
    if ((e = hipMemset(ctx->d_slice_sync, 0, slice_bytes)) != hipSuccess) return bail(e, "hipMemset slice words");
    if ((e = hipMalloc((void **)&ctx->d_slice_cursor, npix * sizeof(uint4))) != hipSuccess) return bail(e, "hipMalloc slice cursors");
    {
        /* kernel0.cl:600-604 for every column and row, with the arithmetic of pixel_coord (bdpt_device.h) */
        std::vector<float> xt((size_t)width), yt((size_t)height);
        const float aspect = clm_div((float)width * 1.0f, (float)height);
        for (int x = 0; x < width; x++) xt[x] = (float)(((double)clm_div((float)x, (float)width) * 2.0 - 1.0) * (double)aspect);
        for (int y = 0; y < height; y++) yt[y] = (float)((double)clm_div((float)y, (float)height) * 2.0 - 1.0);
        if ((e = dev_upload(ctx->d_xt_tab, xt)) != hipSuccess) return bail(e, "pixel column table");
        if ((e = dev_upload(ctx->d_yt_tab, yt)) != hipSuccess) return bail(e, "pixel row table");
    }
    if ((e = hipHostMalloc((void **)&ctx->h_slice_abort, sizeof(uint32_t), hipHostMallocMapped)) != hipSuccess) return bail(e, "hipHostMalloc give-up mark");
    *ctx->h_slice_abort = 0u;
    if ((e = hipHostGetDevicePointer((void **)&ctx->d_slice_abort, ctx->h_slice_abort, 0)) != hipSuccess) return bail(e, "hipHostGetDevicePointer");
    {
        int khz = 0;
        if (hipDeviceGetAttribute(&khz, hipDeviceAttributeWallClockRate, device_ordinal) == hipSuccess && khz > 0) ctx->steady_ticks_per_ms = (double)khz;
        else (void)hipGetLastError();
    }
    *out = ctx;
    return BDPT_OK;
}

void bdpt_destroy(bdpt_ctx *ctx)
{
    if (!ctx) return;
    (void)hipSetDevice(ctx->device);
    if (ctx->stream) (void)hipStreamSynchronize(ctx->stream);
    dev_free(ctx->d_sph); dev_free(ctx->d_pln); dev_free(ctx->d_tri); dev_free(ctx->d_tri_bound); dev_free(ctx->d_tri_cull); dev_free(ctx->d_surf);
    dev_free(ctx->d_sph_surf); dev_free(ctx->d_pln_surf); dev_free(ctx->d_tri_surf);
    dev_free(ctx->d_bvh_nodes); dev_free(ctx->d_bvh_leaf_tri); dev_free(ctx->d_bvh_brute);
    dev_free(ctx->d_seed_w); dev_free(ctx->d_accum_own);
    dev_free(ctx->d_aov_eye); dev_free(ctx->d_aov_hash); dev_free(ctx->d_aov_draws); dev_free(ctx->d_aov_rays);
    dev_free(ctx->d_aov_rad); dev_free(ctx->d_counters); dev_free(ctx->d_slice_sync); dev_free(ctx->d_slice_cursor);
    dev_free(ctx->d_class_masks); dev_free(ctx->d_pool_state); dev_free(ctx->d_xt_tab); dev_free(ctx->d_yt_tab);
    if (ctx->h_slice_abort) (void)hipHostFree(ctx->h_slice_abort);
    if (ctx->ev_begin) (void)hipEventDestroy(ctx->ev_begin);
    if (ctx->ev_end) (void)hipEventDestroy(ctx->ev_end);
    if (ctx->own_stream) (void)hipStreamDestroy(ctx->own_stream);
    delete ctx;
}

static int reset_aovs(bdpt_ctx *ctx)
{
    if (!ctx->d_aov_hash) return BDPT_OK;
    const size_t npix = (size_t)ctx->width * ctx->height;
    HIP_TRY(ctx, hipMemsetAsync(ctx->d_aov_eye, 0xFF, npix * 2 * sizeof(int32_t), ctx->stream));
    HIP_TRY(ctx, hipMemsetAsync(ctx->d_aov_hash, 0, npix * sizeof(uint32_t), ctx->stream));
    HIP_TRY(ctx, hipMemsetAsync(ctx->d_aov_draws, 0, npix * sizeof(uint32_t), ctx->stream));
    HIP_TRY(ctx, hipMemsetAsync(ctx->d_aov_rays, 0, npix * sizeof(uint32_t), ctx->stream));
    HIP_TRY(ctx, hipMemsetAsync(ctx->d_aov_rad, 0, npix * sizeof(float4), ctx->stream));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    return BDPT_OK;
}

static int ensure_aovs(bdpt_ctx *ctx)
{
    if (ctx->d_aov_hash) return BDPT_OK;
    const size_t npix = (size_t)ctx->width * ctx->height;
    hipError_t e = hipMalloc((void **)&ctx->d_aov_eye, npix * 2 * sizeof(int32_t));
    if (e == hipSuccess) e = hipMalloc((void **)&ctx->d_aov_draws, npix * sizeof(uint32_t));
    if (e == hipSuccess) e = hipMalloc((void **)&ctx->d_aov_rays, npix * sizeof(uint32_t));
    if (e == hipSuccess) e = hipMalloc((void **)&ctx->d_aov_rad, npix * sizeof(float4));
    if (e == hipSuccess) e = hipMalloc((void **)&ctx->d_aov_hash, npix * sizeof(uint32_t));   /* last: its presence means "all five exist" */
    if (e != hipSuccess) {
        dev_free(ctx->d_aov_eye); dev_free(ctx->d_aov_draws); dev_free(ctx->d_aov_rays); dev_free(ctx->d_aov_rad);
        dev_free(ctx->d_aov_hash);
        return fail(ctx, BDPT_ERR_HIP, std::string("AOV buffers: hipMalloc: ") + hipGetErrorString(e));
    }
    return reset_aovs(ctx);
}

int bdpt_set_params(bdpt_ctx *ctx, const bdpt_params *p)
{
    if (!ctx || !p) return fail(ctx, BDPT_ERR_INVALID, "bdpt_set_params: null argument");
    if (p->max_light_bounces < 1 || p->max_light_bounces > 64)
        return fail(ctx, BDPT_ERR_INVALID, "bdpt_set_params: max_light_bounces must be in [1, 64]");
    if (!(p->p_continue > 0.0f) || !(p->p_continue < 1.0f))
        return fail(ctx, BDPT_ERR_INVALID, "bdpt_set_params: p_continue must be in (0, 1)");
    if (!(p->blur_factor >= 0.0f)) return fail(ctx, BDPT_ERR_INVALID, "bdpt_set_params: blur_factor must be >= 0");
    if (p->light_index < 0) return fail(ctx, BDPT_ERR_INVALID, "bdpt_set_params: light_index must be >= 0");
    if (p->rng_mode != BDPT_RNG_STREAM && p->rng_mode != BDPT_RNG_STRIDED)
        return fail(ctx, BDPT_ERR_INVALID, "bdpt_set_params: unknown rng_mode");
    if (p->light_origin != BDPT_LIGHT_ORIGIN_SELF && p->light_origin != BDPT_LIGHT_ORIGIN_NONE)
        return fail(ctx, BDPT_ERR_INVALID, "bdpt_set_params: unknown light_origin");
    if (p->connection_falloff != BDPT_FALLOFF_INV_SQ && p->connection_falloff != BDPT_FALLOFF_INV_SQ_PLUS_C)
        return fail(ctx, BDPT_ERR_INVALID, "bdpt_set_params: unknown connection_falloff");
    if (p->accum_mode != BDPT_ACCUM_MEAN && p->accum_mode != BDPT_ACCUM_SUM)
        return fail(ctx, BDPT_ERR_INVALID, "bdpt_set_params: unknown accum_mode");
    if (p->kernel_variant < BDPT_KERNEL_AUTO || p->kernel_variant > BDPT_KERNEL_BVH || p->kernel_variant == 2)
        return fail(ctx, BDPT_ERR_INVALID, "bdpt_set_params: unknown kernel_variant");
    if (p->sample_order < BDPT_ORDER_AUTO || p->sample_order > BDPT_ORDER_ADAPTIVE)
        return fail(ctx, BDPT_ERR_INVALID, "bdpt_set_params: unknown sample_order");
    if ((p->sample_order == BDPT_ORDER_ROULETTE_CLASS || p->sample_order == BDPT_ORDER_ADAPTIVE) &&
        !(p->rng_mode == BDPT_RNG_STRIDED && p->accum_mode == BDPT_ACCUM_SUM))
        return fail(ctx, BDPT_ERR_INVALID, "bdpt_set_params: sample_order ROULETTE_CLASS / ADAPTIVE needs rng_mode STRIDED and accum_mode SUM");
    if (p->sampler < BDPT_SAMPLER_LAMBERT_CONE || p->sampler > BDPT_SAMPLER_LAMBERT_IMP)
        return fail(ctx, BDPT_ERR_INVALID, "bdpt_set_params: unknown sampler");
    if (p->integrator != BDPT_INTEGRATOR_BIDIRECTIONAL && p->integrator != BDPT_INTEGRATOR_PATH_TRACE)
        return fail(ctx, BDPT_ERR_INVALID, "bdpt_set_params: unknown integrator");
    if (p->integrator == BDPT_INTEGRATOR_PATH_TRACE && (p->rng_mode != BDPT_RNG_STREAM || p->accum_mode != BDPT_ACCUM_MEAN))
        return fail(ctx, BDPT_ERR_INVALID, "bdpt_set_params: the PATH_TRACE integrator needs rng_mode STREAM and accum_mode MEAN");
    /* STRIDED spaces the samples of a pixel 32 generator outputs apart.  A sample draws 5 + per_vertex * maxL
     * numbers, the first of which is discarded (kernel0.cl:592, :106-109, :787): with more than 33 draws the
     * last numbers of sample s would be the first used numbers of sample s+1. */
    {
        const int per_vertex = (p->sampler == BDPT_SAMPLER_REFLECTIVE || p->sampler == BDPT_SAMPLER_LAMBERT_IMP) ? 5 : 4;
        if (p->rng_mode == BDPT_RNG_STRIDED && 5 + per_vertex * p->max_light_bounces > 33)
            return fail(ctx, BDPT_ERR_INVALID,
                        "bdpt_set_params: rng_mode STRIDED needs max_light_bounces <= 7 (<= 5 with the REFLECTIVE and LAMBERT_IMP samplers): "
                        "longer light subpaths would reuse the random numbers of the next sample");
    }
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    ctx->params = *p;
    ctx->ms_per_pixel_sample = ctx->last_launch_work = 0.0;               /* other parameters, other cost per sample */
    if (p->aov_enable) return ensure_aovs(ctx);
    return BDPT_OK;
}

int bdpt_get_params(const bdpt_ctx *ctx, bdpt_params *p)
{
    if (!ctx || !p) return BDPT_ERR_INVALID;
    *p = ctx->params;
    return BDPT_OK;
}

int bdpt_upload_scene(bdpt_ctx *ctx, const bdpt_sphere *spheres, int n_spheres, const bdpt_plane *planes, int n_planes,
                      const bdpt_triangle *triangles, int n_triangles, const bdpt_surface *surfaces, int n_surfaces)
{
    if (!ctx) return fail(ctx, BDPT_ERR_INVALID, "bdpt_upload_scene: null context");
    if (n_spheres < 0 || n_planes < 0 || n_triangles < 0 || n_surfaces <= 0)
        return fail(ctx, BDPT_ERR_INVALID, "bdpt_upload_scene: negative count or no surfaces");
    if ((n_spheres && !spheres) || (n_planes && !planes) || (n_triangles && !triangles) || !surfaces)
        return fail(ctx, BDPT_ERR_INVALID, "bdpt_upload_scene: null array with non-zero count");
    if (n_spheres >= (1 << 28) || n_planes >= (1 << 28) || n_triangles >= (1 << 28))
        return fail(ctx, BDPT_ERR_INVALID, "bdpt_upload_scene: too many primitives");
    /* a surface id outside the table would be an out-of-bounds read on the device */
    for (int i = 0; i < n_spheres; i++)
        if (spheres[i].surface >= (uint32_t)n_surfaces) return fail(ctx, BDPT_ERR_INVALID, "bdpt_upload_scene: sphere surface id out of range");
    for (int i = 0; i < n_planes; i++)
        if (planes[i].surface >= (uint32_t)n_surfaces) return fail(ctx, BDPT_ERR_INVALID, "bdpt_upload_scene: plane surface id out of range");
    for (int i = 0; i < n_triangles; i++)
        if (triangles[i].surface >= (uint32_t)n_surfaces) return fail(ctx, BDPT_ERR_INVALID, "bdpt_upload_scene: triangle surface id out of range");
    /* The reference's dot, distance and length work on all four components (kernel0.cl:331, :336, :398, :714); its own
     * scene has w = 0 in every position, direction and normal (main.cpp:282-394), and so three-component arithmetic
     * gives its results exactly.  With any other w (homogeneous w = 1, say) the reference computes something else --
     * a sphere's |o - c|^2 grows by w^2 -- which this library does not reproduce: refuse instead of rendering another
     * image silently.  -0.0f counts as zero. */
    for (int i = 0; i < n_spheres; i++)
        if (spheres[i].center[3] != 0.0f) return fail(ctx, BDPT_ERR_INVALID, "bdpt_upload_scene: sphere " + std::to_string(i) + " has center[3] != 0; the fourth component of positions, normals and directions must be 0");
    for (int i = 0; i < n_planes; i++)
        if (planes[i].normal[3] != 0.0f || planes[i].point[3] != 0.0f) return fail(ctx, BDPT_ERR_INVALID, "bdpt_upload_scene: plane " + std::to_string(i) + " has normal[3] or point[3] != 0; the fourth component of positions, normals and directions must be 0");
    for (int i = 0; i < n_triangles; i++)
        if (triangles[i].p0[3] != 0.0f || triangles[i].p1[3] != 0.0f || triangles[i].p2[3] != 0.0f) return fail(ctx, BDPT_ERR_INVALID, "bdpt_upload_scene: triangle " + std::to_string(i) + " has a vertex with [3] != 0; the fourth component of positions, normals and directions must be 0");
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));

    /* the triangle array carries one zeroed record past the end: the search prefetches record i+1 */
    std::vector<float4> sph(n_spheres), pln((size_t)n_planes * BDPT_PLN_F4),
        tri(((size_t)n_triangles + 1) * BDPT_TRI_F4, make_float4(0.f, 0.f, 0.f, 0.f)),
        bound((size_t)n_triangles + 1, make_float4(0.f, 0.f, 0.f, 0.f)), cull((size_t)n_triangles * 3),
        surf((size_t)n_surfaces * 2);
    std::vector<uint32_t> sph_s(n_spheres), pln_s(n_planes), tri_s(n_triangles);
    std::vector<float> cull_s(n_triangles);
    std::vector<float4> cull_third(n_triangles);       /* third coordinate plane and longest edge of each triangle */
    std::vector<double> cull_len(n_triangles);
    std::vector<double> big_r2(n_triangles);           /* 1.05 R^2 of each bounding sphere, for AUTO's estimate */
    std::vector<double> normal_off(n_triangles);       /* |stored unit normal - true unit normal| of each triangle */
    double cull_c = 0.0;
    for (int i = 0; i < n_spheres; i++) {
        const bdpt_sphere &s = spheres[i];
        sph[i] = make_float4(s.center[0], s.center[1], s.center[2], s.radius * s.radius);   /* kernel0.cl:337 */
        sph_s[i] = s.surface;
    }
    for (int i = 0; i < n_planes; i++) {
        const bdpt_plane &p = planes[i];
        pln[(size_t)i * 2 + 0] = make_float4(p.normal[0], p.normal[1], p.normal[2], 0.0f);
        pln[(size_t)i * 2 + 1] = make_float4(p.point[0], p.point[1], p.point[2], 0.0f);
        pln_s[i] = p.surface;
    }
    for (int i = 0; i < n_triangles; i++) {
        const bdpt_triangle &t = triangles[i];
        const clm_f3 a = v3(t.p0), b = v3(t.p1), c = v3(t.p2);
        const clm_f3 e0 = clm_sub3(c, a), e1 = clm_sub3(b, c), e2 = clm_sub3(a, b);
        const clm_f3 n = clm_normalize3(clm_cross3(clm_sub3(b, a), e0));                     /* kernel0.cl:285 */
        const clm_f3 ns = clm_normalize3(clm_cross3(e2, e1));                                /* kernel0.cl:467 */
        float4 *q = &tri[(size_t)i * BDPT_TRI_F4];
        q[0] = make_float4(n.x, n.y, n.z, a.x);
        q[1] = make_float4(a.y, a.z, e0.x, e0.y);
        q[2] = make_float4(e0.z, c.x, c.y, c.z);
        q[3] = make_float4(e1.x, e1.y, e1.z, b.x);
        q[4] = make_float4(b.y, b.z, e2.x, e2.y);
        q[5] = make_float4(e2.z, ns.x, ns.y, ns.z);
        /* sphere around the vertices, in double and rounded outward; the factor 1.05 on the squared
         * radius is part of the margin the device test leaves for rounding (bdpt_kernels.hip) */
        const double mx = ((double)a.x + b.x + c.x) / 3.0, my = ((double)a.y + b.y + c.y) / 3.0, mz = ((double)a.z + b.z + c.z) / 3.0;
        const float cx = (float)mx, cy = (float)my, cz = (float)mz;
        double r2 = 0.0;
        for (const clm_f3 &v : {a, b, c}) {
            const double dx = (double)v.x - cx, dy = (double)v.y - cy, dz = (double)v.z - cz;
            r2 = std::max(r2, dx * dx + dy * dy + dz * dz);
        }
        /* The 5 % also covers a unit normal that is off by up to 1e-4 because it comes from a cross
         * product with cancellation.  For thinner slivers (and degenerate triangles, whose normal is
         * NaN) the region the edge tests accept need not resemble the triangle at all: no cull then. */
        const double ex[3][3] = {{(double)b.x - a.x, (double)b.y - a.y, (double)b.z - a.z},
                                 {(double)c.x - a.x, (double)c.y - a.y, (double)c.z - a.z},
                                 {(double)c.x - b.x, (double)c.y - b.y, (double)c.z - b.z}};
        const double crx = ex[0][1] * ex[1][2] - ex[0][2] * ex[1][1], cry = ex[0][2] * ex[1][0] - ex[0][0] * ex[1][2],
                     crz = ex[0][0] * ex[1][1] - ex[0][1] * ex[1][0];
        double longest2 = 0.0;
        for (const auto &e : ex) longest2 = std::max(longest2, e[0] * e[0] + e[1] * e[1] + e[2] * e[2]);
        const bool well_shaped = std::sqrt(crx * crx + cry * cry + crz * crz) >= 1.0e-3 * longest2;
        /* stored: w = (1 - 1e-5) |c|^2 - 1.05 R^2 - 2e-6 |c|^2, rounded down (the device test and the last term: bdpt_kernels.hip);
         * -inf = never out of reach, for slivers and for numbers so large that the device's float arithmetic could overflow */
        {
            const double cc = (double)cx * cx + (double)cy * cy + (double)cz * cz;
            const double wd = 0.99999 * cc - r2 * 1.05 - 2.0e-6 * cc;
            const bool tame = well_shaped && cc < 1.0e30 && r2 < 1.0e30;
            bound[i] = make_float4(cx, cy, cz, tame ? std::nextafter((float)wd, -INFINITY) : -INFINITY);
            big_r2[i] = well_shaped ? r2 * 1.05 : INFINITY;
        }
        tri_s[i] = t.surface;
        /* Records of the two-stage kernel's conservative test (closest_hit_lds2, where the margin is derived).
         * In exact arithmetic the reference's edge expressions (kernel0.cl:298-300) are, for a point p of the
         * triangle's plane,  dA = (p-a).((c-a) x n),  dB = (p-c).((b-c) x n),  dC = (p-b).((a-b) x n),  each equal
         * to |(b-a) x (c-a)| at the opposite vertex: divided by that they are barycentric coordinates g_j . p + h_j
         * that add up to 1.  They are stored times s = h / (104 * 2^-24), h the smallest height. */
        {
            const double A[3] = {a.x, a.y, a.z}, B[3] = {b.x, b.y, b.z}, C[3] = {c.x, c.y, c.z};
            const double area2 = std::sqrt(crx * crx + cry * cry + crz * crz);
            const double nt[3] = {crx / area2, cry / area2, crz / area2};
            const double theta = std::sqrt((n.x - nt[0]) * (n.x - nt[0]) + (n.y - nt[1]) * (n.y - nt[1]) + (n.z - nt[2]) * (n.z - nt[2]));
            normal_off[i] = theta;
            const double longest = std::sqrt(longest2);
            const double scale = (area2 / longest) / (104.0 * 0x1p-24);                    /* s */
            auto edge_plane = [&](const double *from, const double *to, const double *vertex, float4 &out) {
                const double e[3] = {to[0] - from[0], to[1] - from[1], to[2] - from[2]};     /* the reference's second cross operand */
                const double f = scale / area2;
                const double g[3] = {(e[1] * nt[2] - e[2] * nt[1]) * f, (e[2] * nt[0] - e[0] * nt[2]) * f, (e[0] * nt[1] - e[1] * nt[0]) * f};
                out = make_float4((float)g[0], (float)g[1], (float)g[2], (float)-(vertex[0] * g[0] + vertex[1] * g[1] + vertex[2] * g[2]));
            };
            float4 *q3 = &cull[(size_t)i * 3];
            q3[0] = make_float4(n.x, n.y, n.z, (float)(A[0] * n.x + A[1] * n.y + A[2] * n.z));
            edge_plane(A, C, A, q3[1]);                                   /* dA: (p-a) x (c-a) */
            edge_plane(C, B, C, q3[2]);                                   /* dB: (p-c) x (b-c); dC is what is left of s */
            edge_plane(B, A, B, cull_third[i]);                           /* dC: (p-b) x (a-b), for the pairing below */
            cull_len[i] = longest;
            double a1 = 0.0;
            for (const double *v : {A, B, C}) a1 = std::max(a1, std::fabs(v[0]) + std::fabs(v[1]) + std::fabs(v[2]));
            const double c_i = longest + a1 + theta * longest / 3.5e-6;
            const bool usable = well_shaped && theta < 1.0e-4 && std::isfinite(scale) && scale < 1.0e30 && std::isfinite(c_i) &&
                                std::isfinite(q3[1].x + q3[1].y + q3[1].z + q3[1].w + q3[2].x + q3[2].y + q3[2].z + q3[2].w);
            if (usable) {
                cull_s[i] = (float)scale;
                cull_c = std::max(cull_c, c_i);
            } else {                                                      /* every coordinate 0: never below the (negative) bound */
                cull_s[i] = 0.0f;
                q3[1] = q3[2] = cull_third[i] = make_float4(0.f, 0.f, 0.f, 0.f);
            }
        }
    }
    for (int i = 0; i < n_surfaces; i++) {
        const bdpt_surface &s = surfaces[i];
        surf[(size_t)i * 2 + 0] = make_float4(s.color[0], s.color[1], s.color[2], s.color[3]);
        surf[(size_t)i * 2 + 1] = make_float4(s.emissive[0], s.emissive[1], s.emissive[2], s.emissive[3]);
    }
    /* the old arrays are freed one by one below: until all eight uploads have succeeded the context has no scene */
    ctx->have_scene = false;
    ctx->spheres.clear();
    ctx->surfaces.clear();
    ctx->n_planes = ctx->n_triangles = 0;
    HIP_TRY(ctx, dev_upload(ctx->d_sph, sph));
    HIP_TRY(ctx, dev_upload(ctx->d_pln, pln));
    HIP_TRY(ctx, dev_upload(ctx->d_tri, tri));
    {
        /* the bounding spheres as the tiled kernel reads them: two spheres A, B per pair of float4, component by component,
         * (cx_A, cx_B, cy_A, cy_B), (cz_A, cz_B, w_A, w_B), so that one packed instruction works on both (bdpt_layout.h);
         * an odd count is completed by a sphere that is out of every ray's reach */
        const size_t pairs = ((size_t)n_triangles + 2) / 2;
        std::vector<float4> paired(pairs * 2);
        for (size_t j = 0; j < pairs; j++) {
            const float4 A = 2 * j < (size_t)n_triangles ? bound[2 * j] : make_float4(0.f, 0.f, 0.f, INFINITY);
            const float4 B = 2 * j + 1 < (size_t)n_triangles ? bound[2 * j + 1] : make_float4(0.f, 0.f, 0.f, INFINITY);
            paired[2 * j] = make_float4(A.x, B.x, A.y, B.y);
            paired[2 * j + 1] = make_float4(A.z, B.z, A.w, B.w);
        }
        HIP_TRY(ctx, dev_upload(ctx->d_tri_bound, paired));
    }
    /* The conservative test's records in the order the test walks them (bdpt_layout.h), per group of 32 triangles:
     * the pairs that share a plane (equal normal and a.n, wherever in the group they are) -- first those
     * that are the two halves of a parallelogram, whose second triangle needs no planes of its own -- then the rest. */
    std::vector<float4> cull_walk;
    std::memset(ctx->cull_group, 0, sizeof ctx->cull_group);
    if (n_triangles <= BDPT_CULL_MAX_TRIANGLES) {
        auto bit = [](int k) { const uint32_t u = 1u << k; float f; std::memcpy(&f, &u, 4); return f; };
        /* Triangles i and j halves of one parallelogram, cut along an edge they share?  Then, with u and v the
         * first one's coordinates that vanish on its two other edges (scaled: s at the opposite vertex), the
         * second one's coordinates are s - u, s - v and u + v - s.  Writes u's and v's planes. */
        auto parallelogram = [&](int i, int j, float4 &pu, float4 &pv) {
            if (cull_s[i] == 0.0f || std::memcmp(&cull_s[i], &cull_s[j], 4) != 0) return false;
            const float *vi[3] = {triangles[i].p0, triangles[i].p1, triangles[i].p2};   /* a, b, c */
            const float *vj[3] = {triangles[j].p0, triangles[j].p1, triangles[j].p2};
            int shared_i[3] = {0, 0, 0}, shared_j[3] = {0, 0, 0}, n_shared = 0;
            for (int x = 0; x < 3; x++)
                for (int y = 0; y < 3; y++)
                    if (!shared_j[y] && !shared_i[x] && std::memcmp(vi[x], vj[y], 3 * sizeof(float)) == 0) { shared_i[x] = shared_j[y] = 1; n_shared++; }
            if (n_shared != 2) return false;
            int oi = 0, oj = 0;                                             /* the vertices that are not shared */
            for (int x = 0; x < 3; x++) { if (!shared_i[x]) oi = x; if (!shared_j[x]) oj = x; }
            const float *s1 = vi[(oi + 1) % 3], *s2 = vi[(oi + 2) % 3];
            const double tol = 2.0e-7 * cull_len[i];
            for (int k = 0; k < 3; k++)
                if (std::fabs(((double)vi[oi][k] + vj[oj][k]) - ((double)s1[k] + s2[k])) > tol) return false;
            /* planes of triangle i: cull[3i+1] vanishes on edge a-c, cull[3i+2] on c-b, cull_third[i] on b-a; the
             * shared edge is the one opposite vertex oi (a: c-b, b: a-c, c: b-a): take the other two */
            const float4 on_ac = cull[(size_t)i * 3 + 1], on_cb = cull[(size_t)i * 3 + 2], on_ba = cull_third[i];
            if (oi == 0) { pu = on_ac; pv = on_ba; }
            else if (oi == 1) { pu = on_cb; pv = on_ba; }
            else { pu = on_ac; pv = on_cb; }
            /* check the prediction where it matters: at the second triangle's corners its coordinates are (s, 0, 0) in some order */
            const double sd = cull_s[i];
            for (int y = 0; y < 3; y++) {
                const double u = (double)pu.x * vj[y][0] + (double)pu.y * vj[y][1] + (double)pu.z * vj[y][2] + pu.w;
                const double v = (double)pv.x * vj[y][0] + (double)pv.y * vj[y][1] + (double)pv.z * vj[y][2] + pv.w;
                double co[3] = {sd - u, sd - v, u + v - sd};
                std::sort(co, co + 3);
                if (std::fabs(co[0]) > 1.0e-4 * sd || std::fabs(co[1]) > 1.0e-4 * sd || std::fabs(co[2] - sd) > 1.0e-4 * sd) return false;
            }
            return true;
        };
        /* Same normal and same a.n, as numbers: a zero may differ in sign between two triangles of one plane (it comes
         * out of a cross product), which changes d.n and k - o.n only where they are zero themselves -- a ray in the
         * plane, which hits neither triangle.  NaNs never compare equal. */
        auto same_plane = [](const float4 &a, const float4 &b) { return a.x == b.x && a.y == b.y && a.z == b.z && a.w == b.w; };
        for (int base = 0; base < n_triangles; base += 32) {
            const int cnt = std::min(32, n_triangles - base);
            const size_t off = cull_walk.size();
            int partner[32];
            for (int k = 0; k < cnt; k++) partner[k] = -1;
            struct Quad { int k, m; float4 c0, pu, pv; bool taken; };
            std::vector<Quad> quads;
            std::vector<float4> slab_rec, quad_rec, pair_rec;
            for (int k = 0; k < cnt; k++) {
                if (partner[k] >= 0) continue;
                for (int m = k + 1; m < cnt; m++)
                    if (partner[m] < 0 && same_plane(cull[(size_t)(base + k) * 3], cull[(size_t)(base + m) * 3])) {
                        partner[k] = m;
                        partner[m] = k;
                        const float4 *a = &cull[(size_t)(base + k) * 3], *b = &cull[(size_t)(base + m) * 3];
                        float4 pu, pv;
                        if (parallelogram(base + k, base + m, pu, pv))
                            quads.push_back({k, m, a[0], pu, pv, false});
                        else
                            pair_rec.insert(pair_rec.end(), {a[0], a[1], a[2], b[1], b[2], make_float4(cull_s[base + k], cull_s[base + m], bit(k), bit(m))});
                        break;
                    }
            }
            /* Two parallelograms that face each other or away from each other, one the other moved along the normal
             * and cut along the same diagonal: the normals are each other's negative and the coordinate planes
             * and s are the same numbers (in either order: the test is symmetric in u and v).  Signed zeros
             * compare equal; they change nothing but the sign of a zero (see same_plane). */
            for (size_t p = 0; p < quads.size(); p++) {
                Quad &P = quads[p];
                if (P.taken) continue;
                for (size_t q = p + 1; q < quads.size(); q++) {
                    Quad &Q = quads[q];
                    if (Q.taken || !(Q.c0.x == -P.c0.x && Q.c0.y == -P.c0.y && Q.c0.z == -P.c0.z)) continue;
                    if (!(cull_s[base + Q.k] == cull_s[base + P.k])) continue;
                    const bool direct = same_plane(Q.pu, P.pu) && same_plane(Q.pv, P.pv);
                    const bool swapped = same_plane(Q.pu, P.pv) && same_plane(Q.pv, P.pu);
                    if (!direct && !swapped) continue;
                    P.taken = Q.taken = true;
                    slab_rec.insert(slab_rec.end(), {P.c0, P.pu, P.pv, make_float4(cull_s[base + P.k], -Q.c0.w, 0.f, 0.f),
                                                     make_float4(bit(P.k), bit(P.m), bit(Q.k), bit(Q.m))});
                    break;
                }
            }
            for (const Quad &P : quads)
                if (!P.taken) quad_rec.insert(quad_rec.end(), {P.c0, P.pu, P.pv, make_float4(cull_s[base + P.k], bit(P.k), bit(P.m), 0.f)});
            cull_walk.insert(cull_walk.end(), slab_rec.begin(), slab_rec.end());
            cull_walk.insert(cull_walk.end(), quad_rec.begin(), quad_rec.end());
            cull_walk.insert(cull_walk.end(), pair_rec.begin(), pair_rec.end());
            int singles = 0;
            for (int k = 0; k < cnt; k++) {
                if (partner[k] >= 0) continue;
                const float4 *a = &cull[(size_t)(base + k) * 3];
                cull_walk.insert(cull_walk.end(), {a[0], a[1], a[2], make_float4(cull_s[base + k], bit(k), 0.f, 0.f)});
                singles++;
            }
            ctx->cull_group[base >> 5] = (uint32_t)off | ((uint32_t)(slab_rec.size() / BDPT_CULL_SLAB_F4) << 11) |
                                         ((uint32_t)(quad_rec.size() / BDPT_CULL_QUAD_F4) << 16) |
                                         ((uint32_t)(pair_rec.size() / BDPT_CULL_PAIR_F4) << 21) | ((uint32_t)singles << 26);
        }
    }
    ctx->cull_f4 = (int)cull_walk.size();
    HIP_TRY(ctx, dev_upload(ctx->d_tri_cull, cull_walk));
    ctx->cull_c = (float)(cull_c * 1.000001);
    HIP_TRY(ctx, dev_upload(ctx->d_surf, surf));
    HIP_TRY(ctx, dev_upload(ctx->d_sph_surf, sph_s));
    HIP_TRY(ctx, dev_upload(ctx->d_pln_surf, pln_s));
    HIP_TRY(ctx, dev_upload(ctx->d_tri_surf, tri_s));
    {
        bdpt_bvh_host bvh;
        auto vertex = [&](int i, int k) { return k == 0 ? triangles[i].p0 : k == 1 ? triangles[i].p1 : triangles[i].p2; };
        ctx->have_bvh = bdpt_bvh_build(vertex, bound.data(), big_r2.data(), normal_off.data(), n_triangles, bvh);
        if (!ctx->have_bvh) bvh = bdpt_bvh_host();
        HIP_TRY(ctx, dev_upload(ctx->d_bvh_nodes, bvh.nodes));
        HIP_TRY(ctx, dev_upload(ctx->d_bvh_leaf_tri, bvh.leaf_tri));
        HIP_TRY(ctx, dev_upload(ctx->d_bvh_brute, bvh.brute));
        ctx->bvh_root = bvh.root;
        ctx->bvh_n_brute = (int)bvh.brute.size();
        ctx->bvh_depth = bvh.depth;
        ctx->bvh_nodes = (int)(bvh.nodes.size() / BDPT_BVH_NODE_F4);
        ctx->bvh_stack_total = std::max(1, std::min(bvh.stack_need, ctx->bvh_stack_cap));
        ctx->bvh_stack_limit = std::min(ctx->bvh_stack_total, ctx->bvh_lds_cap);
    }
    /* AUTO's estimate, in instructions per ray: the LDS variant runs the full test (about 64) on every
     * triangle; the tiled variant runs the sphere test (13) on every triangle and the full test, with
     * scalar-load operands (about 110), on those some lane of a wavefront comes near -- practically
     * always for a triangle whose sphere is larger than an eighth of the scene, seldom for the rest. */
    {
        double lo[3] = {1e300, 1e300, 1e300}, hi[3] = {-1e300, -1e300, -1e300};
        auto grow = [&](const float *p3, double r) {
            for (int k = 0; k < 3; k++) { lo[k] = std::min(lo[k], (double)p3[k] - r); hi[k] = std::max(hi[k], (double)p3[k] + r); }
        };
        for (int i = 0; i < n_spheres; i++) grow(spheres[i].center, spheres[i].radius);
        for (int i = 0; i < n_triangles; i++) { grow(triangles[i].p0, 0); grow(triangles[i].p1, 0); grow(triangles[i].p2, 0); }
        double diag2 = 0.0;
        for (int k = 0; k < 3; k++) diag2 += hi[k] > lo[k] ? (hi[k] - lo[k]) * (hi[k] - lo[k]) : 0.0;
        const double big2 = diag2 / 4.0 / 64.0 * 1.05;                     /* (extent / 8)^2, against the stored 1.05 R^2 */
        long long n_big = 0;
        for (int i = 0; i < n_triangles; i++)
            if (!(big_r2[i] < big2)) n_big++;
        ctx->auto_tiled = n_triangles > 0 && 13LL * n_triangles + 110LL * n_big < 64LL * n_triangles;
    }
    ctx->spheres.assign(spheres, spheres + n_spheres);
    ctx->surfaces.assign(surfaces, surfaces + n_surfaces);
    ctx->n_planes = n_planes;
    ctx->n_triangles = n_triangles;
    ctx->have_scene = true;
    ctx->ms_per_pixel_sample = ctx->last_launch_work = 0.0;
    ctx->num_samples = 0;
    return reset_aovs(ctx);
}

int bdpt_upload_seeds(bdpt_ctx *ctx, const uint32_t *seeds)
{
    if (!ctx || !seeds) return fail(ctx, BDPT_ERR_INVALID, "bdpt_upload_seeds: null argument");
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    const int W = ctx->width, H = ctx->height;
    const size_t npix = (size_t)W * H;
    ctx->seeds.assign(seeds, seeds + npix * 4);
    std::vector<uint32_t> w(npix);
    for (int y = 0; y < H; y++)
        for (int x = 0; x < W; x++) w[(size_t)y * W + x] = seeds[((size_t)x * H + y) * 4 + 3];   /* kernel0.cl:590 */
    HIP_TRY(ctx, hipMemcpy(ctx->d_seed_w, w.data(), npix * sizeof(uint32_t), hipMemcpyHostToDevice));
    ctx->have_seeds = true;
    return reset_aovs(ctx);
}

int bdpt_read_seeds(bdpt_ctx *ctx, uint32_t *seeds)
{
    if (!ctx || !seeds) return fail(ctx, BDPT_ERR_INVALID, "bdpt_read_seeds: null argument");
    if (!ctx->have_seeds) return fail(ctx, BDPT_ERR_STATE, "bdpt_read_seeds: no seeds uploaded");
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    const int W = ctx->width, H = ctx->height;
    const size_t npix = (size_t)W * H;
    std::vector<uint32_t> w(npix);
    HIP_TRY(ctx, hipMemcpy(w.data(), ctx->d_seed_w, npix * sizeof(uint32_t), hipMemcpyDeviceToHost));
    for (int y = 0; y < H; y++)
        for (int x = 0; x < W; x++) ctx->seeds[((size_t)x * H + y) * 4 + 3] = w[(size_t)y * W + x];
    std::memcpy(seeds, ctx->seeds.data(), npix * 4 * sizeof(uint32_t));
    return check_sliced(ctx, "bdpt_read_seeds");
}

int bdpt_set_camera(bdpt_ctx *ctx, const bdpt_camera *cam)
{
    if (!ctx || !cam) return fail(ctx, BDPT_ERR_INVALID, "bdpt_set_camera: null argument");
    /* as for the scene: the reference normalises the ray direction over four components (kernel0.cl:215) and its
     * camera vectors have w = 0 (camera.cpp:137-139) */
    if (cam->pos[3] != 0.0f || cam->look_at[3] != 0.0f || cam->up[3] != 0.0f)
        return fail(ctx, BDPT_ERR_INVALID, "bdpt_set_camera: pos[3], look_at[3] and up[3] must be 0");
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    ctx->camera = *cam;
    ctx->have_camera = true;
    ctx->num_samples = 0;                                                 /* main.cpp:152 */
    return reset_aovs(ctx);
}

int bdpt_num_samples(const bdpt_ctx *ctx) { return ctx ? ctx->num_samples : BDPT_ERR_INVALID; }

int bdpt_set_num_samples(bdpt_ctx *ctx, int n)
{
    if (!ctx || n < 0) return fail(ctx, BDPT_ERR_INVALID, "bdpt_set_num_samples: bad argument");
    ctx->num_samples = n;
    return BDPT_OK;
}

int bdpt_set_region(bdpt_ctx *ctx, int x0, int y0, int w, int h)
{
    if (!ctx) return BDPT_ERR_INVALID;
    if (x0 < 0 || y0 < 0 || w <= 0 || h <= 0 || x0 + w > ctx->width || y0 + h > ctx->height)
        return fail(ctx, BDPT_ERR_INVALID, "bdpt_set_region: rectangle outside the image");
    ctx->rx0 = x0; ctx->ry0 = y0; ctx->rw = w; ctx->rh = h;
    ctx->ms_per_pixel_sample = ctx->last_launch_work = 0.0;               /* another part of the frame, another mix of paths */
    return BDPT_OK;
}

int bdpt_set_tile_interleave(bdpt_ctx *ctx, int tile_w, int tile_h, int stride, int phase)
{
    if (!ctx) return BDPT_ERR_INVALID;
    if (stride <= 1) {
        if (stride < 1 || phase != 0) return fail(ctx, BDPT_ERR_INVALID, "bdpt_set_tile_interleave: stride must be >= 1 and phase 0 when stride is 1");
        ctx->tile_stride = 1; ctx->tile_phase = 0;
        return BDPT_OK;
    }
    if (tile_w < 16 || tile_w % 16 || tile_h < 8 || tile_h % 8)
        return fail(ctx, BDPT_ERR_INVALID, "bdpt_set_tile_interleave: tile width must be a multiple of 16 pixels, tile height of 8");
    if (phase < 0 || phase >= stride) return fail(ctx, BDPT_ERR_INVALID, "bdpt_set_tile_interleave: phase must be in [0, stride)");
    ctx->tile_w = tile_w; ctx->tile_h = tile_h; ctx->tile_stride = stride; ctx->tile_phase = phase;
    return BDPT_OK;
}

int bdpt_set_sample_offset(bdpt_ctx *ctx, int first_sample)
{
    if (!ctx || first_sample < 0) return fail(ctx, BDPT_ERR_INVALID, "bdpt_set_sample_offset: bad argument");
    ctx->sample_offset = first_sample;
    return BDPT_OK;
}

int bdpt_set_stream(bdpt_ctx *ctx, void *hip_stream)
{
    if (!ctx) return BDPT_ERR_INVALID;
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    ctx->stream = hip_stream ? (hipStream_t)hip_stream : ctx->own_stream;
    ctx->have_launch = false;
    return BDPT_OK;
}

int bdpt_use_default_stream(bdpt_ctx *ctx)
{
    if (!ctx) return BDPT_ERR_INVALID;
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    ctx->stream = nullptr;                                                /* HIP's default stream */
    ctx->have_launch = false;
    return BDPT_OK;
}

int bdpt_bind_accum(bdpt_ctx *ctx, void *device_rgba)
{
    if (!ctx) return BDPT_ERR_INVALID;
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    if (device_rgba && ((uintptr_t)device_rgba & 15u))
        return fail(ctx, BDPT_ERR_INVALID, "bdpt_bind_accum: pointer must be 16-byte aligned");
    ctx->d_accum = device_rgba ? (float4 *)device_rgba : ctx->d_accum_own;
    return BDPT_OK;
}

void *bdpt_accum_device_ptr(bdpt_ctx *ctx) { return ctx ? (void *)ctx->d_accum : nullptr; }

int bdpt_enable_counters(bdpt_ctx *ctx, int on)
{
    if (!ctx) return BDPT_ERR_INVALID;
    ctx->counters_on = on != 0;
    return BDPT_OK;
}

static int fill_kargs(bdpt_ctx *ctx, int spp, bdpt_kargs &K)
{
    const bdpt_params &P = ctx->params;
    if (!ctx->have_scene) return fail(ctx, BDPT_ERR_STATE, "bdpt_render: no scene uploaded");
    if (!ctx->have_seeds) return fail(ctx, BDPT_ERR_STATE, "bdpt_render: no seeds uploaded");
    if (!ctx->have_camera) return fail(ctx, BDPT_ERR_STATE, "bdpt_render: no camera set");
    if (spp < 1 || spp > (1 << 24)) return fail(ctx, BDPT_ERR_INVALID, "bdpt_render: spp must be in [1, 2^24]");
    if (P.light_index >= (int)ctx->spheres.size())
        return fail(ctx, BDPT_ERR_STATE, "bdpt_render: light_index is not a sphere of the uploaded scene");
    std::memset(&K, 0, sizeof K);
    K.scene.sph = ctx->d_sph; K.scene.sph_surf = ctx->d_sph_surf;
    K.scene.pln = ctx->d_pln; K.scene.pln_surf = ctx->d_pln_surf;
    K.scene.tri = ctx->d_tri; K.scene.tri_bound = ctx->d_tri_bound; K.scene.tri_cull = ctx->d_tri_cull; K.scene.cull_f4 = ctx->cull_f4; K.scene.cull_c = ctx->cull_c; std::memcpy(K.scene.cull_group, ctx->cull_group, sizeof K.scene.cull_group); K.scene.tri_surf = ctx->d_tri_surf;
    K.scene.bvh_nodes = ctx->d_bvh_nodes; K.scene.bvh_leaf_tri = ctx->d_bvh_leaf_tri; K.scene.bvh_brute = ctx->d_bvh_brute;
    K.scene.bvh_root = ctx->bvh_root; K.scene.bvh_n_brute = ctx->bvh_n_brute; K.scene.bvh_stack_limit = ctx->bvh_stack_limit; K.scene.bvh_stack_total = ctx->bvh_stack_total;
    K.scene.bvh_share = ctx->n_triangles - ctx->bvh_n_brute >= ctx->bvh_share_min ? 1 : 0;
    K.scene.surf = ctx->d_surf;
    K.scene.n_spheres = (int)ctx->spheres.size();
    K.scene.n_planes = ctx->n_planes;
    K.scene.n_triangles = ctx->n_triangles;
    K.scene.n_surfaces = (int)ctx->surfaces.size();
    K.seed_w = ctx->d_seed_w;
    K.accum = ctx->d_accum;
    K.aov_eye_hit = ctx->d_aov_eye; K.aov_hit_hash = ctx->d_aov_hash; K.aov_radiance = ctx->d_aov_rad;
    K.aov_rng_draws = ctx->d_aov_draws; K.aov_rays = ctx->d_aov_rays;
    K.counters = ctx->d_counters;
    K.slice_sync = ctx->d_slice_sync;
    K.slice_cursor = ctx->d_slice_cursor;
    K.slice_abort = ctx->d_slice_abort;
    K.xt_tab = ctx->d_xt_tab; K.yt_tab = ctx->d_yt_tab;
    K.n_tiles = ctx->slice_tiles;                                       /* room; the launcher puts the launch's count here */
    K.n_chunks = 1;
    K.chunk_spp = spp;
    K.wg_x = 0;
    K.block = 0;
    K.width = ctx->width; K.height = ctx->height;
    K.rx0 = ctx->rx0; K.ry0 = ctx->ry0; K.rw = ctx->rw; K.rh = ctx->rh;
    K.tile_stride = ctx->tile_stride; K.tile_phase = ctx->tile_phase;
    K.tile_w = ctx->tile_w; K.tile_h = ctx->tile_h;
    K.tiles_x = ctx->tile_stride > 1 ? (ctx->rw + ctx->tile_w - 1) / ctx->tile_w : 1;
    K.num_samples = ctx->num_samples;
    K.spp = spp;
    K.first_sample = (uint32_t)(ctx->sample_offset + ctx->num_samples);

    const bdpt_camera &C = ctx->camera;
    const clm_f3 right = clm_cross3(v3(C.up), v3(C.look_at));            /* kernel0.cl:212 */
    for (int k = 0; k < 3; k++) { K.cam_pos[k] = C.pos[k]; K.cam_look[k] = C.look_at[k]; K.cam_up[k] = C.up[k]; }
    K.cam_right[0] = right.x; K.cam_right[1] = right.y; K.cam_right[2] = right.z;
    K.aspect = clm_div((float)ctx->width * 1.0f, (float)ctx->height);   /* kernel0.cl:603 */
    K.delta_x = clm_div(P.blur_factor, (float)ctx->width);               /* kernel0.cl:196 */
    K.delta_y = clm_div(P.blur_factor, (float)ctx->height);              /* kernel0.cl:197 */

    const bdpt_sphere &L = ctx->spheres[P.light_index];
    for (int k = 0; k < 3; k++) K.light_c[k] = L.center[k];
    K.light_r = L.radius;
    for (int k = 0; k < 4; k++) K.light_emissive[k] = ctx->surfaces[L.surface].emissive[k];   /* kernel0.cl:563 */
    if (P.light_origin == BDPT_LIGHT_ORIGIN_SELF) { K.light_origin_type = 0; K.light_origin_index = P.light_index; }
    else { K.light_origin_type = -1; K.light_origin_index = -1; }

    K.max_l = P.max_light_bounces;
    K.p_con = P.p_continue;
    K.pre_factor0 = clm_div(1.0f - P.p_continue, 1.0f - clm_powi(P.p_continue, P.max_light_bounces));   /* kernel0.cl:694 */
    {
        float pre = K.pre_factor0;
        for (int k = 0; k < BDPT_INV_PRE_N; k++) { K.inv_pre[k] = clm_div(1.0f, pre); pre *= P.p_continue; }   /* kernel0.cl:741, :745 */
        /* sqrt is monotone and correctly rounded, so "sqrt(x) < 0.0001f" holds exactly below one threshold */
        uint32_t lo = 0u, hi = 0x7F800000u;                               /* smallest bit pattern whose root is >= 0.0001f */
        while (lo < hi) {
            const uint32_t mid = lo + (hi - lo) / 2u;
            float x; std::memcpy(&x, &mid, 4);
            if (clm_sqrt(x) >= 0.0001f) hi = mid; else lo = mid + 1u;
        }
        std::memcpy(&K.same_point_d2, &lo, 4);
    }
    K.indirect_only = P.indirect_only;
    K.rng_mode = P.rng_mode;
    K.falloff = P.connection_falloff;
    K.falloff_c = P.falloff_c;
    K.accum_mode = P.accum_mode;
    K.sampler = P.sampler;
    K.imp_light_weight = P.imp_light_weight;
    K.imp_half_angle = P.imp_half_angle;
    K.reorder = 0;
    if (P.rng_mode == BDPT_RNG_STRIDED && P.accum_mode == BDPT_ACCUM_SUM && P.sample_order != BDPT_ORDER_INDEX)
        K.reorder = P.sample_order == BDPT_ORDER_ROULETTE_CLASS ? 1 : 2;   /* AUTO and ADAPTIVE: decided per pixel */
    /* (float)x * 2^-32 > p_con, the roulette test of kernel0.cl:788, as an integer test x > fate_above: the
     * conversion is monotone in x and the scaling exact, so there is one such threshold */
    {
        uint64_t lo = 0u, hi = 0xFFFFFFFFu;                               /* largest x whose draw does not exceed p_con */
        while (lo < hi) {
            const uint64_t mid = lo + (hi - lo + 1u) / 2u;
            if ((float)(uint32_t)mid * 0x1p-32f > P.p_continue) hi = mid - 1u; else lo = mid;
        }
        K.fate_above = (uint32_t)lo;
    }
    return BDPT_OK;
}

int bdpt_render_async(bdpt_ctx *ctx, int spp)
{
    if (!ctx) return fail(ctx, BDPT_ERR_INVALID, "bdpt_render: null context");
    bdpt_kargs K;
    int rc = fill_kargs(ctx, spp, K);
    if (rc != BDPT_OK) return rc;
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    /* a launch that was abandoned since the host last looked: do not build on its image */
    if ((rc = check_sliced(ctx, "bdpt_render")) != BDPT_OK) return rc;
    /* how long a sample of this scene takes, from the last launch if it has ended (no waiting; a stream under
     * capture is left alone) */
    if (ctx->have_launch && ctx->last_launch_work > 0.0) {
        hipStreamCaptureStatus capturing = hipStreamCaptureStatusNone;
        if (hipStreamIsCapturing(ctx->stream, &capturing) != hipSuccess) (void)hipGetLastError();
        if (capturing == hipStreamCaptureStatusNone) {
            float ms = 0.0f;
            if (hipEventQuery(ctx->ev_end) == hipSuccess && hipEventElapsedTime(&ms, ctx->ev_begin, ctx->ev_end) == hipSuccess && ms > 0.0f) {
                ctx->ms_per_pixel_sample = ms / ctx->last_launch_work;
                ctx->last_launch_work = 0.0;
            } else (void)hipGetLastError();              /* not ready is not an error */
        }
    }
    const bool aov = ctx->params.aov_enable != 0;
    if (aov && (rc = ensure_aovs(ctx)) != BDPT_OK) return rc;
    int variant = ctx->params.kernel_variant;
    if ((variant == BDPT_KERNEL_LDS || variant == BDPT_KERNEL_LDS2) &&
        bdpt_lds_scene_bytes(K.scene, variant == BDPT_KERNEL_LDS2) > BDPT_LDS_SCENE_MAX)
        return fail(ctx, BDPT_ERR_INVALID, "bdpt_render: scene does not fit the LDS-resident kernel");
    if (variant == BDPT_KERNEL_BVH && !ctx->have_bvh)
        return fail(ctx, BDPT_ERR_INVALID, "bdpt_render: the scene has no triangle hierarchy (too many triangles that no sphere bounds, or too deep for the walk's stack)");
    /* AUTO: see the estimate in bdpt_upload_scene; scenes for the many-triangle kernels walk the hierarchy when they have
     * one and enough triangles for it to pay */
    if (variant == BDPT_KERNEL_AUTO)
        variant = !(bdpt_lds_scene_bytes(K.scene, true) <= BDPT_LDS_SCENE_MAX && K.scene.n_triangles <= BDPT_CULL_MAX_TRIANGLES && !ctx->auto_tiled)
                      ? (ctx->have_bvh && K.scene.n_triangles >= BDPT_BVH_AUTO_MIN_TRIANGLES ? BDPT_KERNEL_BVH : BDPT_KERNEL_TILED)
                  : K.scene.n_triangles > 0 ? BDPT_KERNEL_LDS2 : BDPT_KERNEL_LDS;    /* nothing for the two-stage test to do without triangles */
    /* Pixels sorted by the class of their next sample (bdpt_class_pool.h): launches in index order.  Left to itself the
     * library runs them so from BDPT_CPOOL_AUTO_MIN_SPP samples on (1080p Cornell box, sequential generator, running
     * mean: 1 sample 0.25 against 0.20 ms, 4 samples 0.79 against 0.89, 64 samples 9.3 against 9.6, 1024 samples 118
     * against 134).  Its runs cost less to start than the per-lane kernels' (the image is updated in place), so a
     * launch may be cut into twice as many.  Not over the hierarchy walk unless asked for: that kernel lives on the
     * number of resident wavefronts, and the pool's LDS costs it three of 24 (100 000 triangles, 64 samples: 1344
     * against 1310 ms). */
    const bool by_class = ctx->params.integrator != BDPT_INTEGRATOR_PATH_TRACE && bdpt_class_pool_supported(K, variant) &&
                          (ctx->path_scheduling == BDPT_PATHS_BY_CLASS ||
                           (ctx->path_scheduling == BDPT_PATHS_AUTO && spp >= BDPT_CPOOL_AUTO_MIN_SPP && variant != BDPT_KERNEL_BVH));
    /* Sliced launch (slice_begin in bdpt_kernels.hip): runs short enough that the launch does not end with a long
     * drain, long enough that the per-run reload of the pixel state does not show (1080p Cornell box: 64 samples
     * 10.4 -> 9.8 ms as 8 runs, 256 samples 47.5 -> 42.2 ms as 16, 1024 samples 175 -> 158 ms as 16, no gain from
     * 32 or 64; 8 samples lose 4 % as 2 runs).  The many-triangle kernel spends so long on a sample that one sample is a run already
     * (100 000 triangles, 16 samples: 3.59 -> 3.27 s as 8 or 16 runs). */
    if (ctx->params.integrator != BDPT_INTEGRATOR_PATH_TRACE) {
        int chunks = ctx->launch_chunks;
        if (chunks == 0) {
            /* a run should last a few times what it costs to start; how long a sample takes is known from the last
             * launch of this scene (the Cornell box: 6e-8 ms per pixel, runs of 7 samples; six spheres under an open
             * sky: 1.2e-8, runs of 34), else assumed short */
            int min_spp = variant == BDPT_KERNEL_TILED ? 1 : BDPT_AUTO_CHUNK_MIN_SPP;
            if (ctx->ms_per_pixel_sample > 0.0)
                min_spp = (int)std::min(64.0, std::max(1.0, std::ceil(3.0 * BDPT_RUN_COST_MS_PER_PIXEL / ctx->ms_per_pixel_sample)));
            chunks = std::min(by_class ? 2 * BDPT_AUTO_CHUNKS : BDPT_AUTO_CHUNKS, spp / min_spp);
        }
        chunks = std::max(1, std::min(chunks, spp));
        K.chunk_spp = (spp + chunks - 1) / chunks;
        K.n_chunks = (spp + K.chunk_spp - 1) / K.chunk_spp;
    }
    /* Class order from a table: one mask word per class and 32 samples of every pixel of the frame.  The table is
     * kept from launch to launch and grows when a launch needs more; a launch whose table would exceed the limit, or
     * that could not get the memory (or may not ask for it: a stream under capture), classifies on the fly. */
    if (K.reorder && ctx->params.integrator != BDPT_INTEGRATOR_PATH_TRACE && variant != BDPT_KERNEL_TILED && K.max_l <= 7) {
        const size_t words = ((size_t)spp + 31) / 32;
        const size_t bytes = (size_t)ctx->width * ctx->height * (size_t)K.max_l * words * sizeof(uint32_t);
        if (bytes <= ctx->order_scratch_limit) {
            if (bytes > ctx->class_mask_bytes) {
                hipStreamCaptureStatus capturing = hipStreamCaptureStatusNone;
                if (hipStreamIsCapturing(ctx->stream, &capturing) != hipSuccess) (void)hipGetLastError();
                if (capturing == hipStreamCaptureStatusNone) {
                    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));          /* an earlier launch may still read the old table */
                    dev_free(ctx->d_class_masks);
                    ctx->class_mask_bytes = 0;
                    /* not more than half of what is free now: other contexts and ranks on this GPU want memory too */
                    size_t free_b = 0, total_b = 0;
                    if (hipMemGetInfo(&free_b, &total_b) != hipSuccess) { free_b = 0; (void)hipGetLastError(); }
                    if (bytes <= free_b / 2 && hipMalloc((void **)&ctx->d_class_masks, bytes) == hipSuccess) ctx->class_mask_bytes = bytes;
                    else { ctx->d_class_masks = nullptr; (void)hipGetLastError(); }
                }
            }
            if (bytes <= ctx->class_mask_bytes) { K.class_masks = ctx->d_class_masks; K.mask_words = (int)words; }
        }
    }
    ctx->last_order_table = K.class_masks ? 1 : 0;
    /* How long a workgroup of a sliced launch may wait for the run before it.  A legitimate wait is at most one run
     * of one tile plus scheduling, far below the duration of the whole launch; the limit is a hundred times the
     * expected duration of the launch (from the measured time per sample when there is one, else from a generous
     * 1e-5 ms per pixel-sample and triangle), at least a minute. */
    {
        double wait_ms = ctx->slice_wait_s * 1e3;
        if (!(ctx->slice_wait_s > 0.0)) {
            const double work = (double)K.rw * K.rh / std::max(1, K.tile_stride) * spp;
            const double per = ctx->ms_per_pixel_sample > 0.0 ? ctx->ms_per_pixel_sample : 1.0e-5 * (1.0 + K.scene.n_triangles);
            wait_ms = std::max(60.0e3, 100.0 * work * per);
        }
        K.slice_wait_ticks = (unsigned long long)std::min(1.0e18, wait_ms * ctx->steady_ticks_per_ms);
    }
    /* Paths sorted by phase inside a wavefront (bdpt_pool.h), where the launch allows it.  Part of a path's state waits
     * in device memory, every run of a tile in a region of its own: kept from launch to launch, grown when a launch needs
     * more (never while the stream is being captured), and a launch that cannot have it runs one lane per pixel. */
    bool pooled = ctx->params.integrator != BDPT_INTEGRATOR_PATH_TRACE && ctx->path_scheduling == BDPT_PATHS_POOLED &&
                  bdpt_pool_supported(K, variant);
    auto pool_state_for = [&](size_t bytes) -> int {
        if (bytes > ctx->pool_state_bytes && bytes <= ctx->order_scratch_limit) {
            hipStreamCaptureStatus capturing = hipStreamCaptureStatusNone;
            if (hipStreamIsCapturing(ctx->stream, &capturing) != hipSuccess) (void)hipGetLastError();
            if (capturing == hipStreamCaptureStatusNone) {
                HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));              /* an earlier launch may still use the old one */
                dev_free(ctx->d_pool_state);
                ctx->pool_state_bytes = 0;
                size_t free_b = 0, total_b = 0;
                if (hipMemGetInfo(&free_b, &total_b) != hipSuccess) { free_b = 0; (void)hipGetLastError(); }
                if (bytes <= free_b / 2 && hipMalloc((void **)&ctx->d_pool_state, bytes) == hipSuccess) ctx->pool_state_bytes = bytes;
                else { ctx->d_pool_state = nullptr; (void)hipGetLastError(); }
            }
        }
        return BDPT_OK;
    };
    if (pooled) {
        int n_chunks = K.n_chunks, chunk_spp = K.chunk_spp;
        auto cut = [&](int chunks) {
            chunk_spp = (spp + chunks - 1) / chunks;
            n_chunks = (spp + chunk_spp - 1) / chunk_spp;
        };
        if (ctx->launch_chunks == 0)                     /* long runs keep a pool full; few runs keep the state small */
            cut(std::max(1, std::min(BDPT_POOL_AUTO_CHUNKS, spp / BDPT_POOL_AUTO_CHUNK_MIN_SPP)));
        size_t bytes = bdpt_pool_state_bytes(K, variant, n_chunks);
        while (n_chunks > 1 && bytes > ctx->order_scratch_limit) {
            cut(n_chunks / 2);
            bytes = bdpt_pool_state_bytes(K, variant, n_chunks);
        }
        if ((rc = pool_state_for(bytes)) != BDPT_OK) return rc;
        if (bytes == 0 || bytes > ctx->pool_state_bytes) pooled = false;     /* the per-lane launch keeps its own cut */
        else { K.pool_state = ctx->d_pool_state; K.n_chunks = n_chunks; K.chunk_spp = chunk_spp; }
    }
    if (ctx->counters_on) HIP_TRY(ctx, hipMemsetAsync(ctx->d_counters, 0, BDPT_N_COUNTERS * sizeof(unsigned long long), ctx->stream));
    HIP_TRY(ctx, hipEventRecord(ctx->ev_begin, ctx->stream));
    ctx->last_path_scheduling = pooled ? BDPT_PATHS_POOLED : by_class ? BDPT_PATHS_BY_CLASS : BDPT_PATHS_PER_LANE;
    if (ctx->params.integrator == BDPT_INTEGRATOR_PATH_TRACE) {
        variant = BDPT_KERNEL_DIRECT;
        HIP_TRY(ctx, bdpt_launch_path_trace(K, aov, ctx->stream));
    } else if (pooled) HIP_TRY(ctx, bdpt_launch_pool(K, variant, aov, ctx->counters_on, ctx->stream));
    else if (by_class) HIP_TRY(ctx, bdpt_launch_class_pool(K, variant, aov, ctx->counters_on, ctx->stream));
    else if (variant == BDPT_KERNEL_DIRECT) HIP_TRY(ctx, bdpt_launch_direct(K, aov, ctx->counters_on, ctx->stream));
    else if (variant == BDPT_KERNEL_LDS) HIP_TRY(ctx, bdpt_launch_lds(K, aov, ctx->counters_on, ctx->stream));
    else if (variant == BDPT_KERNEL_LDS2) HIP_TRY(ctx, bdpt_launch_lds2(K, aov, ctx->counters_on, ctx->stream));
    else if (variant == BDPT_KERNEL_BVH) HIP_TRY(ctx, bdpt_launch_bvh(K, aov, ctx->counters_on, ctx->stream));
    else HIP_TRY(ctx, bdpt_launch_tiled(K, aov, ctx->counters_on, ctx->stream));
    HIP_TRY(ctx, hipEventRecord(ctx->ev_end, ctx->stream));
    ctx->have_launch = true;
    ctx->last_launch_work = (double)K.rw * K.rh / std::max(1, K.tile_stride) * spp;
    ctx->last_variant = variant;
    ctx->last_chunks = K.n_chunks;
    ctx->num_samples += spp;                                              /* main.cpp:131 */
    return BDPT_OK;
}

int bdpt_sync(bdpt_ctx *ctx)
{
    if (!ctx) return BDPT_ERR_INVALID;
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    return check_sliced(ctx, "bdpt_sync");
}

int bdpt_render(bdpt_ctx *ctx, int spp)
{
    int rc = bdpt_render_async(ctx, spp);
    if (rc != BDPT_OK) return rc;
    return bdpt_sync(ctx);                                                /* main.cpp:121-124 */
}

int bdpt_set_launch_chunks(bdpt_ctx *ctx, int chunks)
{
    if (!ctx) return BDPT_ERR_INVALID;
    if (chunks < 0 || chunks > 4096) return fail(ctx, BDPT_ERR_INVALID, "bdpt_set_launch_chunks: 0 (automatic) to 4096");
    ctx->launch_chunks = chunks;
    return BDPT_OK;
}

int bdpt_set_slice_wait_limit(bdpt_ctx *ctx, double seconds)
{
    if (!ctx) return BDPT_ERR_INVALID;
    if (!(seconds >= 0.0) || seconds > 1.0e6) return fail(ctx, BDPT_ERR_INVALID, "bdpt_set_slice_wait_limit: 0 (automatic) to 1e6 seconds");
    ctx->slice_wait_s = seconds;
    return BDPT_OK;
}

int bdpt_last_launch_order_table(const bdpt_ctx *ctx) { return ctx ? ctx->last_order_table : 0; }

int bdpt_set_path_scheduling(bdpt_ctx *ctx, int mode)
{
    if (!ctx) return BDPT_ERR_INVALID;
    if (mode < BDPT_PATHS_AUTO || mode > BDPT_PATHS_BY_CLASS) return fail(ctx, BDPT_ERR_INVALID, "bdpt_set_path_scheduling: unknown mode");
    ctx->path_scheduling = mode;
    ctx->ms_per_pixel_sample = ctx->last_launch_work = 0.0;               /* another launch form, another cost per sample */
    return BDPT_OK;
}

int bdpt_last_launch_path_scheduling(const bdpt_ctx *ctx) { return ctx ? ctx->last_path_scheduling : 0; }

int bdpt_set_order_scratch_limit(bdpt_ctx *ctx, size_t bytes)
{
    if (!ctx) return BDPT_ERR_INVALID;
    ctx->order_scratch_limit = bytes;
    return BDPT_OK;
}

int bdpt_last_launch_variant(const bdpt_ctx *ctx) { return ctx ? ctx->last_variant : 0; }

int bdpt_hierarchy_info(const bdpt_ctx *ctx, int info[4])
{
    if (!ctx || !info || !ctx->have_scene || !ctx->have_bvh) return 0;
    info[0] = ctx->bvh_nodes;
    info[1] = ctx->bvh_depth;
    info[2] = ctx->bvh_stack_limit;
    info[3] = ctx->bvh_n_brute;
    return 1;
}
int bdpt_last_launch_chunks(const bdpt_ctx *ctx) { return ctx ? ctx->last_chunks : 0; }

int bdpt_two_stage_record_counts(const bdpt_ctx *ctx, int out[4])
{
    if (!ctx || !out) return BDPT_ERR_INVALID;
    out[0] = out[1] = out[2] = out[3] = 0;
    for (int g = 0; g * 32 < ctx->n_triangles && g < BDPT_CULL_MAX_TRIANGLES / 32; g++) {
        out[3] += (int)((ctx->cull_group[g] >> 11) & 0xfu);
        out[0] += (int)((ctx->cull_group[g] >> 16) & 0x1fu);
        out[1] += (int)((ctx->cull_group[g] >> 21) & 0x1fu);
        out[2] += (int)(ctx->cull_group[g] >> 26);
    }
    return BDPT_OK;
}

int bdpt_last_launch_ms(bdpt_ctx *ctx, float *ms)
{
    if (!ctx || !ms) return fail(ctx, BDPT_ERR_INVALID, "bdpt_last_launch_ms: null argument");
    if (!ctx->have_launch) return fail(ctx, BDPT_ERR_STATE, "bdpt_last_launch_ms: nothing launched yet");
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    HIP_TRY(ctx, hipEventSynchronize(ctx->ev_end));
    HIP_TRY(ctx, hipEventElapsedTime(ms, ctx->ev_begin, ctx->ev_end));
    return check_sliced(ctx, "bdpt_last_launch_ms");
}

int bdpt_read_accum(bdpt_ctx *ctx, float *rgba)
{
    if (!ctx || !rgba) return fail(ctx, BDPT_ERR_INVALID, "bdpt_read_accum: null argument");
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    HIP_TRY(ctx, hipMemcpy(rgba, ctx->d_accum, (size_t)ctx->width * ctx->height * sizeof(float4), hipMemcpyDeviceToHost));
    return check_sliced(ctx, "bdpt_read_accum");
}

int bdpt_write_accum(bdpt_ctx *ctx, const float *rgba)
{
    if (!ctx || !rgba) return fail(ctx, BDPT_ERR_INVALID, "bdpt_write_accum: null argument");
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    HIP_TRY(ctx, hipMemcpy(ctx->d_accum, rgba, (size_t)ctx->width * ctx->height * sizeof(float4), hipMemcpyHostToDevice));
    return BDPT_OK;
}

int bdpt_read_aov(bdpt_ctx *ctx, int kind, void *out)
{
    if (!ctx || !out) return fail(ctx, BDPT_ERR_INVALID, "bdpt_read_aov: null argument");
    if (!ctx->d_aov_hash) return fail(ctx, BDPT_ERR_STATE, "bdpt_read_aov: params.aov_enable was never set");
    const size_t npix = (size_t)ctx->width * ctx->height;
    const void *src = nullptr;
    size_t bytes = 0;
    switch (kind) {
    case BDPT_AOV_EYE_HIT: src = ctx->d_aov_eye; bytes = npix * 2 * sizeof(int32_t); break;
    case BDPT_AOV_HIT_HASH: src = ctx->d_aov_hash; bytes = npix * sizeof(uint32_t); break;
    case BDPT_AOV_RADIANCE: src = ctx->d_aov_rad; bytes = npix * sizeof(float4); break;
    case BDPT_AOV_RNG_DRAWS: src = ctx->d_aov_draws; bytes = npix * sizeof(uint32_t); break;
    case BDPT_AOV_RAYS: src = ctx->d_aov_rays; bytes = npix * sizeof(uint32_t); break;
    default: return fail(ctx, BDPT_ERR_INVALID, "bdpt_read_aov: unknown kind");
    }
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    HIP_TRY(ctx, hipMemcpy(out, src, bytes, hipMemcpyDeviceToHost));
    return check_sliced(ctx, "bdpt_read_aov");
}

int bdpt_read_counters(bdpt_ctx *ctx, uint64_t out[BDPT_N_COUNTERS])
{
    if (!ctx || !out) return fail(ctx, BDPT_ERR_INVALID, "bdpt_read_counters: null argument");
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    unsigned long long c[BDPT_N_COUNTERS];
    HIP_TRY(ctx, hipMemcpy(c, ctx->d_counters, sizeof c, hipMemcpyDeviceToHost));
    for (int i = 0; i < BDPT_N_COUNTERS; i++) out[i] = c[i];
    return check_sliced(ctx, "bdpt_read_counters");
}

int bdpt_query_rays(bdpt_ctx *ctx, const float *rays, const int32_t *origin_ids, int n, int32_t *out_ids, float *out_pos,
                    float *out_normal)
{
    if (!ctx || n < 0 || (n && (!rays || !origin_ids || !out_ids || !out_pos || !out_normal)))
        return fail(ctx, BDPT_ERR_INVALID, "bdpt_query_rays: bad argument");
    if (!ctx->have_scene) return fail(ctx, BDPT_ERR_STATE, "bdpt_query_rays: no scene uploaded");
    if (n == 0) return BDPT_OK;
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    float *d_rays = nullptr, *d_pos = nullptr, *d_n = nullptr;
    int *d_org = nullptr, *d_id = nullptr;
    int rc = BDPT_OK;
    auto step = [&](hipError_t e, const char *what) {
        if (e != hipSuccess && rc == BDPT_OK) rc = fail(ctx, BDPT_ERR_HIP, std::string("bdpt_query_rays: ") + what + ": " + hipGetErrorString(e));
        return e == hipSuccess;
    };
    bdpt_dev_scene S{};
    S.sph = ctx->d_sph; S.sph_surf = ctx->d_sph_surf; S.pln = ctx->d_pln; S.pln_surf = ctx->d_pln_surf;
    S.tri = ctx->d_tri; S.tri_bound = ctx->d_tri_bound; S.tri_surf = ctx->d_tri_surf; S.surf = ctx->d_surf;
    S.n_spheres = (int)ctx->spheres.size(); S.n_planes = ctx->n_planes; S.n_triangles = ctx->n_triangles;
    S.n_surfaces = (int)ctx->surfaces.size();
    const size_t N = (size_t)n;
    if (step(hipMalloc((void **)&d_rays, N * 6 * sizeof(float)), "hipMalloc") &&
        step(hipMalloc((void **)&d_org, N * 2 * sizeof(int)), "hipMalloc") &&
        step(hipMalloc((void **)&d_id, N * 2 * sizeof(int)), "hipMalloc") &&
        step(hipMalloc((void **)&d_pos, N * 3 * sizeof(float)), "hipMalloc") &&
        step(hipMalloc((void **)&d_n, N * 3 * sizeof(float)), "hipMalloc") &&
        step(hipMemcpy(d_rays, rays, N * 6 * sizeof(float), hipMemcpyHostToDevice), "copy rays") &&
        step(hipMemcpy(d_org, origin_ids, N * 2 * sizeof(int), hipMemcpyHostToDevice), "copy origins") &&
        step(bdpt_launch_query(S, d_rays, d_org, n, d_id, d_pos, d_n, ctx->stream), "launch") &&
        step(hipStreamSynchronize(ctx->stream), "sync") &&
        step(hipMemcpy(out_ids, d_id, N * 2 * sizeof(int), hipMemcpyDeviceToHost), "copy ids") &&
        step(hipMemcpy(out_pos, d_pos, N * 3 * sizeof(float), hipMemcpyDeviceToHost), "copy pos")) {
        step(hipMemcpy(out_normal, d_n, N * 3 * sizeof(float), hipMemcpyDeviceToHost), "copy normals");
    }
    dev_free(d_rays); dev_free(d_org); dev_free(d_id); dev_free(d_pos); dev_free(d_n);
    return rc;
}

int bdpt_query_rays_noalloc(bdpt_ctx *ctx, const float *rays, const int32_t *origin_ids, int n, int32_t *out_ids,
                            float *out_pos, float *out_normal)
{
    if (!ctx || n < 0 || (n && (!rays || !origin_ids || !out_ids || !out_pos || !out_normal)))
        return fail(ctx, BDPT_ERR_INVALID, "bdpt_query_rays_noalloc: bad argument");
    if (n > BDPT_QUERY_STATIC_MAX)
        return fail(ctx, BDPT_ERR_INVALID, "bdpt_query_rays_noalloc: at most " + std::to_string(BDPT_QUERY_STATIC_MAX) +
                                               " rays per call (the staging array has a fixed size); use bdpt_query_rays or split the batch");
    if (!ctx->have_scene) return fail(ctx, BDPT_ERR_STATE, "bdpt_query_rays_noalloc: no scene uploaded");
    if (n == 0) return BDPT_OK;
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    void *base = nullptr;
    HIP_TRY(ctx, bdpt_query_scratch(ctx->device, &base));
    /* fixed offsets for the full capacity, so that every part stays 256-byte aligned */
    const size_t M = BDPT_QUERY_STATIC_MAX;
    uint32_t *w = (uint32_t *)base;
    float *d_rays = (float *)w;
    int *d_org = (int *)(w + 6 * M), *d_id = (int *)(w + 8 * M);
    float *d_pos = (float *)(w + 10 * M), *d_n = (float *)(w + 13 * M);
    bdpt_dev_scene S{};
    S.sph = ctx->d_sph; S.sph_surf = ctx->d_sph_surf; S.pln = ctx->d_pln; S.pln_surf = ctx->d_pln_surf;
    S.tri = ctx->d_tri; S.tri_bound = ctx->d_tri_bound; S.tri_surf = ctx->d_tri_surf; S.surf = ctx->d_surf;
    S.n_spheres = (int)ctx->spheres.size(); S.n_planes = ctx->n_planes; S.n_triangles = ctx->n_triangles;
    S.n_surfaces = (int)ctx->surfaces.size();
    const size_t N = (size_t)n;
    /* everything on the context's stream, in order; the kernel writes all three outputs of every ray i < n,
     * so nothing a previous (possibly larger) call left in the array can reach the caller */
    HIP_TRY(ctx, hipMemcpyAsync(d_rays, rays, N * 6 * sizeof(float), hipMemcpyHostToDevice, ctx->stream));
    HIP_TRY(ctx, hipMemcpyAsync(d_org, origin_ids, N * 2 * sizeof(int), hipMemcpyHostToDevice, ctx->stream));
    HIP_TRY(ctx, bdpt_launch_query(S, d_rays, d_org, n, d_id, d_pos, d_n, ctx->stream));
    HIP_TRY(ctx, hipMemcpyAsync(out_ids, d_id, N * 2 * sizeof(int), hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(ctx, hipMemcpyAsync(out_pos, d_pos, N * 3 * sizeof(float), hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(ctx, hipMemcpyAsync(out_normal, d_n, N * 3 * sizeof(float), hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    return BDPT_OK;
}

int bdpt_clmath_eval(bdpt_ctx *ctx, int op, const double *in, double *out, int n)
{
    if (!ctx || n < 0 || op < 0 || op > 7 || (n && (!in || !out))) return fail(ctx, BDPT_ERR_INVALID, "bdpt_clmath_eval: bad argument");
    if (n == 0) return BDPT_OK;
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    double *d_in = nullptr, *d_out = nullptr;
    int rc = BDPT_OK;
    auto step = [&](hipError_t e, const char *what) {
        if (e != hipSuccess && rc == BDPT_OK) rc = fail(ctx, BDPT_ERR_HIP, std::string("bdpt_clmath_eval: ") + what + ": " + hipGetErrorString(e));
        return e == hipSuccess;
    };
    const size_t N = (size_t)n;
    if (step(hipMalloc((void **)&d_in, N * 3 * sizeof(double)), "hipMalloc") &&
        step(hipMalloc((void **)&d_out, N * sizeof(double)), "hipMalloc") &&
        step(hipMemcpy(d_in, in, N * 3 * sizeof(double), hipMemcpyHostToDevice), "copy in") &&
        step(bdpt_launch_clmath(op, d_in, d_out, n, ctx->stream), "launch") &&
        step(hipStreamSynchronize(ctx->stream), "sync")) {
        step(hipMemcpy(out, d_out, N * sizeof(double), hipMemcpyDeviceToHost), "copy out");
    }
    dev_free(d_in); dev_free(d_out);
    return rc;
}

}  /* extern "C" */

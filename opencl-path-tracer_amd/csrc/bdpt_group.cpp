/*
 * bdpt_group.cpp -- several GPUs behind one handle, inside one process.
 *
 * The reference drives one device from one thread (src/renderer/main.cpp:242-260, :522); nothing here
 * has a counterpart in it.  A group owns one context per GPU.  Every pixel-sample is independent, so a
 * render call deals the samples of each pixel to the members (member r renders its own range of sample
 * indices of every pixel, STRIDED generator, SUM accumulation) and the image is the sum of the members'
 * partial sums divided by the sample count.  The sum is one reduce to member 0: RCCL over xGMI when the
 * library can be loaded (looked up at run time, so that a host that already holds an RCCL -- torch's --
 * keeps using its own), else device-to-device copies into member 0 and an add kernel.  Launches are
 * asynchronous on each member's stream; one host thread drives all of them.
 */
#include <dlfcn.h>
#include <hip/hip_runtime.h>

#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "../../include/bdpt.h"
#include "bdpt_ctx_internal.h"
#include "bdpt_kernels.h"

namespace {

/* the few RCCL entry points used, resolved with dlsym (types as in rccl.h: ncclFloat = 7, ncclSum = 0) */
struct Rccl {
    void *lib = nullptr;
    int (*CommInitAll)(void **comms, int ndev, const int *devlist) = nullptr;
    int (*CommDestroy)(void *comm) = nullptr;
    int (*Reduce)(const void *send, void *recv, size_t count, int dtype, int op, int root, void *comm, hipStream_t stream) = nullptr;
    int (*GroupStart)() = nullptr;
    int (*GroupEnd)() = nullptr;
    const char *(*GetErrorString)(int) = nullptr;
    bool load()
    {
        for (const char *name : {"librccl.so.1", "librccl.so"}) {
            lib = dlopen(name, RTLD_NOW | RTLD_LOCAL);
            if (lib) break;
        }
        if (!lib) return false;
        CommInitAll = (decltype(CommInitAll))dlsym(lib, "ncclCommInitAll");
        CommDestroy = (decltype(CommDestroy))dlsym(lib, "ncclCommDestroy");
        Reduce = (decltype(Reduce))dlsym(lib, "ncclReduce");
        GroupStart = (decltype(GroupStart))dlsym(lib, "ncclGroupStart");
        GroupEnd = (decltype(GroupEnd))dlsym(lib, "ncclGroupEnd");
        GetErrorString = (decltype(GetErrorString))dlsym(lib, "ncclGetErrorString");
        return CommInitAll && CommDestroy && Reduce && GroupStart && GroupEnd && GetErrorString;
    }
};

}  // namespace

struct bdpt_group {
    std::vector<bdpt_ctx *> members;
    std::vector<int> devices;
    int width = 0, height = 0;
    int total_samples = 0;                 /* samples per pixel rendered since the last reset, all members together */
    std::vector<int> member_samples;       /* of which by each member */
    int next_member = 0;                   /* the member that takes the first of the next call's left-over samples: calls of
                                              fewer samples than members (one sample per frame) go round the members */
    bool needs_reset = false;              /* a render call failed half-way: the members' counters no longer add up */
    bool use_rccl = false;
    Rccl rccl;
    std::vector<void *> comms;
    float4 *d_total = nullptr;             /* on member 0: the combined sum */
    float4 *d_stage = nullptr;             /* on member 0: one member's partial sum in transit (copy path) */
    std::string err;
};

static thread_local std::string g_group_error;

static int gfail(bdpt_group *g, int code, const std::string &msg)
{
    if (g) g->err = msg;
    else g_group_error = msg;
    return code;
}

#define G_HIP(g, call)                                                                              \
    do {                                                                                            \
        hipError_t e_ = (call);                                                                     \
        if (e_ != hipSuccess) return gfail(g, BDPT_ERR_HIP, std::string(#call) + ": " + hipGetErrorString(e_)); \
    } while (0)

/* every member, stopping at the first failure */
#define G_EACH(g, expr)                                                                             \
    do {                                                                                            \
        for (size_t i_ = 0; i_ < (g)->members.size(); i_++) {                                       \
            bdpt_ctx *m = (g)->members[i_];                                                         \
            int rc_ = (expr);                                                                       \
            if (rc_ != BDPT_OK) return gfail(g, rc_, "member " + std::to_string(i_) + ": " + bdpt_last_error(m)); \
        }                                                                                           \
    } while (0)

extern "C" {

const char *bdpt_group_last_error(const bdpt_group *g) { return g ? g->err.c_str() : g_group_error.c_str(); }

void bdpt_group_destroy(bdpt_group *g)
{
    if (!g) return;
    for (size_t i = 0; i < g->comms.size(); i++)
        if (g->comms[i]) {
            (void)hipSetDevice(g->devices[i]);
            g->rccl.CommDestroy(g->comms[i]);
        }
    if (!g->devices.empty()) (void)hipSetDevice(g->devices[0]);
    if (g->d_total) (void)hipFree(g->d_total);
    if (g->d_stage) (void)hipFree(g->d_stage);
    for (bdpt_ctx *m : g->members) bdpt_destroy(m);
    delete g;
}

int bdpt_group_create(const int *device_ordinals, int n_devices, int width, int height, int combine, bdpt_group **out)
{
    if (!out) return gfail(nullptr, BDPT_ERR_INVALID, "bdpt_group_create: out is null");
    *out = nullptr;
    if (!device_ordinals || n_devices < 1 || n_devices > 64) return gfail(nullptr, BDPT_ERR_INVALID, "bdpt_group_create: need 1..64 devices");
    if (combine < BDPT_COMBINE_AUTO || combine > BDPT_COMBINE_COPY) return gfail(nullptr, BDPT_ERR_INVALID, "bdpt_group_create: unknown combine");
    bdpt_group *g = new bdpt_group();
    g->width = width;
    g->height = height;
    auto bail = [&](int code, const std::string &msg) {
        bdpt_group_destroy(g);
        return gfail(nullptr, code, msg);
    };
    bool distinct = true;
    for (int i = 0; i < n_devices; i++) {
        for (int j = 0; j < i; j++) distinct = distinct && device_ordinals[i] != device_ordinals[j];
        bdpt_ctx *m = nullptr;
        int rc = bdpt_create(device_ordinals[i], width, height, &m);
        if (rc != BDPT_OK) return bail(rc, std::string("bdpt_group_create: member ") + std::to_string(i) + ": " + bdpt_last_error(nullptr));
        g->members.push_back(m);
        g->devices.push_back(device_ordinals[i]);
        g->member_samples.push_back(0);
        bdpt_params p;
        bdpt_default_params(&p);
        p.rng_mode = BDPT_RNG_STRIDED;                /* any member can render any sample range */
        p.accum_mode = BDPT_ACCUM_SUM;
        if ((rc = bdpt_set_params(m, &p)) != BDPT_OK) return bail(rc, std::string("bdpt_group_create: ") + bdpt_last_error(m));
    }
    const size_t bytes = (size_t)width * height * sizeof(float4);
    hipError_t e = hipSetDevice(g->devices[0]);
    if (e == hipSuccess) e = hipMalloc((void **)&g->d_total, bytes);
    if (e == hipSuccess && n_devices > 1) e = hipMalloc((void **)&g->d_stage, bytes);
    if (e != hipSuccess) return bail(BDPT_ERR_HIP, std::string("bdpt_group_create: hipMalloc: ") + hipGetErrorString(e));
    /* RCCL wants one communicator per distinct GPU; a group that names a GPU twice (a rehearsal on one GPU) copies */
    if (combine != BDPT_COMBINE_COPY && distinct && g->rccl.load()) {
        g->comms.assign(n_devices, nullptr);
        int rc = g->rccl.CommInitAll(g->comms.data(), n_devices, g->devices.data());
        if (rc != 0) {
            const std::string why = g->rccl.GetErrorString(rc);
            g->comms.clear();
            if (combine == BDPT_COMBINE_RCCL) return bail(BDPT_ERR_HIP, "bdpt_group_create: ncclCommInitAll: " + why);
        } else {
            g->use_rccl = true;
        }
    } else if (combine == BDPT_COMBINE_RCCL) {
        return bail(BDPT_ERR_INVALID, distinct ? "bdpt_group_create: RCCL (librccl.so.1) cannot be loaded"
                                               : "bdpt_group_create: RCCL needs distinct GPUs");
    }
    *out = g;
    return BDPT_OK;
}

int bdpt_group_size(const bdpt_group *g) { return g ? (int)g->members.size() : BDPT_ERR_INVALID; }

bdpt_ctx *bdpt_group_member(bdpt_group *g, int i) { return (g && i >= 0 && i < (int)g->members.size()) ? g->members[i] : nullptr; }

int bdpt_group_combine(const bdpt_group *g) { return g ? (g->use_rccl ? BDPT_COMBINE_RCCL : BDPT_COMBINE_COPY) : BDPT_ERR_INVALID; }

int bdpt_group_num_samples(const bdpt_group *g) { return g ? g->total_samples : BDPT_ERR_INVALID; }

static void group_reset(bdpt_group *g)
{
    g->total_samples = 0;
    for (int &n : g->member_samples) n = 0;
    g->next_member = 0;
    g->needs_reset = false;
}

int bdpt_group_set_params(bdpt_group *g, const bdpt_params *p)
{
    if (!g || !p) return gfail(g, BDPT_ERR_INVALID, "bdpt_group_set_params: null argument");
    if (p->rng_mode != BDPT_RNG_STRIDED || p->accum_mode != BDPT_ACCUM_SUM)
        return gfail(g, BDPT_ERR_INVALID, "bdpt_group_set_params: a group splits samples across GPUs, which needs rng_mode STRIDED and accum_mode SUM");
    G_EACH(g, bdpt_set_params(m, p));
    return BDPT_OK;
}

int bdpt_group_upload_scene(bdpt_group *g, const bdpt_sphere *spheres, int n_spheres, const bdpt_plane *planes, int n_planes,
                            const bdpt_triangle *triangles, int n_triangles, const bdpt_surface *surfaces, int n_surfaces)
{
    if (!g) return gfail(g, BDPT_ERR_INVALID, "bdpt_group_upload_scene: null group");
    G_EACH(g, bdpt_upload_scene(m, spheres, n_spheres, planes, n_planes, triangles, n_triangles, surfaces, n_surfaces));
    group_reset(g);
    return BDPT_OK;
}

int bdpt_group_upload_seeds(bdpt_group *g, const uint32_t *seeds)
{
    if (!g) return gfail(g, BDPT_ERR_INVALID, "bdpt_group_upload_seeds: null group");
    G_EACH(g, bdpt_upload_seeds(m, seeds));         /* the same seeds everywhere: a sample is named by (pixel seed, sample index) */
    return BDPT_OK;
}

int bdpt_group_set_camera(bdpt_group *g, const bdpt_camera *cam)
{
    if (!g) return gfail(g, BDPT_ERR_INVALID, "bdpt_group_set_camera: null group");
    G_EACH(g, bdpt_set_camera(m, cam));             /* zeroes every member's counter, as main.cpp:152 does */
    group_reset(g);
    return BDPT_OK;
}

int bdpt_group_render(bdpt_group *g, int spp)
{
    if (!g || spp < 1) return gfail(g, BDPT_ERR_INVALID, "bdpt_group_render: bad argument");
    if (g->needs_reset) return gfail(g, BDPT_ERR_STATE, "bdpt_group_render: an earlier render call failed part-way; reset with bdpt_group_set_camera");
    const int n = (int)g->members.size();
    const int base = spp / n, rem = spp % n;
    /* Members in the order next_member, next_member + 1, ...: each takes a contiguous range of the call's sample
     * indices, the first `rem` of them one sample more.  The counters change only when every member has launched
     * and finished; a failure before that leaves them inconsistent with what the members hold, and the group says
     * so until it is reset. */
    std::vector<int> add(n, 0);
    int first = g->total_samples;
    for (int k = 0; k < n; k++) {                                         /* enqueue on every GPU, then wait for all */
        const int r = (g->next_member + k) % n;
        const int count = base + (k < rem ? 1 : 0);
        if (count == 0) continue;
        bdpt_ctx *m = g->members[r];
        /* the kernel numbers a launch's samples from sample_offset + (the member's own counter) */
        int rc = bdpt_set_sample_offset(m, first - g->member_samples[r]);
        if (rc == BDPT_OK) rc = bdpt_render_async(m, count);
        if (rc != BDPT_OK) {
            g->needs_reset = true;
            return gfail(g, rc, "member " + std::to_string(r) + ": " + bdpt_last_error(m));
        }
        add[r] = count;
        first += count;
    }
    for (size_t i = 0; i < g->members.size(); i++) {
        const int rc = bdpt_sync(g->members[i]);
        if (rc != BDPT_OK) {
            g->needs_reset = true;
            return gfail(g, rc, "member " + std::to_string(i) + ": " + bdpt_last_error(g->members[i]));
        }
    }
    for (int r = 0; r < n; r++) g->member_samples[r] += add[r];
    g->total_samples += spp;
    g->next_member = (g->next_member + rem) % n;
    return BDPT_OK;
}

int bdpt_group_read_image(bdpt_group *g, float *rgba)
{
    if (!g || !rgba) return gfail(g, BDPT_ERR_INVALID, "bdpt_group_read_image: null argument");
    if (g->needs_reset) return gfail(g, BDPT_ERR_STATE, "bdpt_group_read_image: an earlier render call failed part-way; reset with bdpt_group_set_camera");
    if (g->total_samples < 1) return gfail(g, BDPT_ERR_STATE, "bdpt_group_read_image: nothing rendered since the last reset");
    const int n = (int)g->members.size();
    const size_t npix = (size_t)g->width * g->height, bytes = npix * sizeof(float4);
    const hipStream_t s0 = bdpt_ctx_stream(g->members[0]);
    /* a member that has rendered nothing since the reset (fewer samples than GPUs so far) holds an old image */
    std::vector<int> live;
    for (int r = 0; r < n; r++)
        if (g->member_samples[r] > 0) live.push_back(r);
    if (g->use_rccl && (int)live.size() == n) {
        if (g->rccl.GroupStart() != 0) return gfail(g, BDPT_ERR_HIP, "ncclGroupStart failed");
        for (int r = 0; r < n; r++) {
            G_HIP(g, hipSetDevice(g->devices[r]));
            const int rc = g->rccl.Reduce(bdpt_accum_device_ptr(g->members[r]), r == 0 ? (void *)g->d_total : nullptr, npix * 4, /*ncclFloat*/ 7,
                                          /*ncclSum*/ 0, 0, g->comms[r], bdpt_ctx_stream(g->members[r]));
            if (rc != 0) {
                (void)g->rccl.GroupEnd();
                return gfail(g, BDPT_ERR_HIP, std::string("ncclReduce: ") + g->rccl.GetErrorString(rc));
            }
        }
        const int rc = g->rccl.GroupEnd();
        if (rc != 0) return gfail(g, BDPT_ERR_HIP, std::string("ncclGroupEnd: ") + g->rccl.GetErrorString(rc));
        G_EACH(g, bdpt_sync(m));
        G_HIP(g, hipSetDevice(g->devices[0]));
    } else {
        G_HIP(g, hipSetDevice(g->devices[0]));
        bool first = true;
        for (int r : live) {
            const void *src = bdpt_accum_device_ptr(g->members[r]);
            float4 *dst = first ? g->d_total : g->d_stage;
            if (g->devices[r] == g->devices[0]) G_HIP(g, hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToDevice, s0));
            else G_HIP(g, hipMemcpyPeerAsync(dst, g->devices[0], src, g->devices[r], bytes, s0));
            if (!first) G_HIP(g, bdpt_launch_add(g->d_total, g->d_stage, npix, s0));
            first = false;
        }
    }
    G_HIP(g, bdpt_launch_scale(g->d_total, npix, 1.0f / (float)g->total_samples, s0));
    G_HIP(g, hipStreamSynchronize(s0));
    G_HIP(g, hipMemcpy(rgba, g->d_total, bytes, hipMemcpyDeviceToHost));
    return BDPT_OK;
}

}  /* extern "C" */

/*
 * bdpt_kernels.h -- host-callable launchers of the kernels in bdpt_kernels.hip.
 * Every launcher only enqueues on `stream`; none of them allocates or waits.
 */
#ifndef BDPT_KERNELS_H
#define BDPT_KERNELS_H

#include "bdpt_layout.h"

/* spp samples for every pixel of the region in K; one lane per pixel. */
hipError_t bdpt_launch_direct(const bdpt_kargs &K, bool aov, bool count, hipStream_t stream);

/* The same work with the whole scene resident in LDS (scenes up to BDPT_LDS_SCENE_MAX bytes of records). */
#define BDPT_LDS_SCENE_MAX (24u * 1024u)
size_t bdpt_lds_scene_bytes(const bdpt_dev_scene &S, bool with_cull_records);
hipError_t bdpt_launch_lds(const bdpt_kargs &K, bool aov, bool count, hipStream_t stream);

/* LDS-resident scene, triangles in two stages: a conservative test for all lanes together, then the full test
 * per lane on what is left (the scene's cull records go to LDS as well). */
hipError_t bdpt_launch_lds2(const bdpt_kargs &K, bool aov, bool count, hipStream_t stream);

/* Scenes of many triangles: a hierarchy over the triangles' bounding spheres (bdpt_bvh.h, K.scene.bvh_*), walked by
 * every lane for itself; same hits as the exhaustive searches. */
hipError_t bdpt_launch_bvh(const bdpt_kargs &K, bool aov, bool count, hipStream_t stream);

/* The same work for scenes of any size: triangle records staged through wave-private LDS tiles. */
hipError_t bdpt_launch_tiled(const bdpt_kargs &K, bool aov, bool count, hipStream_t stream);

/* The same work with a wavefront's paths sorted by phase (bdpt_pool.h): a wavefront owns more pixels than lanes and runs,
 * every round, the phase most of their paths wait in.  variant: BDPT_KERNEL_LDS or BDPT_KERNEL_LDS2 (how the search reads
 * the scene).  Not every launch can run this way; ask first. */
bool bdpt_pool_supported(const bdpt_kargs &K, int variant);
/* bytes of K.pool_state such a launch of K's region needs when it is cut into `chunks` runs per tile */
size_t bdpt_pool_state_bytes(const bdpt_kargs &K, int variant, int chunks);
hipError_t bdpt_launch_pool(const bdpt_kargs &K, int variant, bool aov, bool count, hipStream_t stream);

/* Launches that take every pixel's samples in index order, with a wavefront's pixels sorted by the class of their next
 * sample (bdpt_class_pool.h): a wavefront owns BDPT_CPOOL_W x BDPT_CPOOL_ROWS pixels and runs one whole sample per round for 64 of them whose light
 * subpaths roulette will cut to about the same length.  Needs no memory beyond the per-pixel launches'.  variant: BDPT_KERNEL_DIRECT, _LDS or _LDS2. */
bool bdpt_class_pool_supported(const bdpt_kargs &K, int variant);
hipError_t bdpt_launch_class_pool(const bdpt_kargs &K, int variant, bool aov, bool count, hipStream_t stream);

/* The reference's older unidirectional kernel (kernel0.cl:872-1001): spp samples per pixel of the region in one launch. */
hipError_t bdpt_launch_path_trace(const bdpt_kargs &K, bool aov, hipStream_t stream);

/* dst[i] += src[i] and dst[i] *= factor over n float4 on one device (combining partial sums of several GPUs). */
hipError_t bdpt_launch_add(float4 *dst, const float4 *src, size_t n, hipStream_t stream);
hipError_t bdpt_launch_scale(float4 *dst, size_t n, float factor, hipStream_t stream);

/* Closest hit of n caller-supplied rays (6 floats each: origin, direction;
 * origin[2*i], origin[2*i+1] = primitive to skip or (4,0)). Device pointers. */
hipError_t bdpt_launch_query(const bdpt_dev_scene &S, const float *rays, const int *origin, int n, int *out_id,
                             float *out_pos, float *out_n, hipStream_t stream);

/* Device address, on GPU `device` (which must be the current device), of a staging array that is part of
 * the loaded code object: BDPT_QUERY_STATIC_MAX * 16 32-bit words, shared by every caller on that GPU.
 * Looked up on first use per device; never allocates. */
#define BDPT_QUERY_STATIC_MAX 65536
#define BDPT_MAX_DEVICES 64
hipError_t bdpt_query_scratch(int device, void **out);

/* Numeric-contract built-ins evaluated on the device (see the kernel). */
hipError_t bdpt_launch_clmath(int op, const double *in, double *out, int n, hipStream_t stream);

#endif

/*
 * bdpt_oracle.h -- CPU oracle of the reference's BDPT sample kernel.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under oracle/ is part of the product:
 * only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
 * load it, and only as the checker or as a timed CPU baseline.
 *
 * PARITY UNPINNED: the reference ships no tests, fixtures or golden images, and
 * its kernel cannot be executed in this environment (no OpenCL device; building
 * it for the host would need stand-ins for the OpenCL C library).  The oracle
 * is therefore a hand restatement of src/kernel/kernel0.cl, function by
 * function, with the OpenCL built-ins instantiated by include/bdpt_clmath.h.
 */
#ifndef BDPT_ORACLE_H
#define BDPT_ORACLE_H

#include <stdint.h>
#include "../include/bdpt.h"

#ifdef __cplusplus
extern "C" {
#endif

typedef struct orc_job {
    int width, height;
    const bdpt_camera *camera;
    const bdpt_sphere *spheres;     int n_spheres;
    const bdpt_plane *planes;       int n_planes;
    const bdpt_triangle *triangles; int n_triangles;
    const bdpt_surface *surfaces;   int n_surfaces;
    const bdpt_params *params;
    int x0, y0, w, h;               /* pixel rectangle to render */
    int num_samples;                /* imgDesc.numSamples before the first sample */
    int spp;                        /* consecutive samples to render */
    int sample_offset;              /* STRIDED mode: index of sample 0 */
    uint32_t *seeds;                /* in/out, width*height*4, index x*height+y */
    float *accum;                   /* in/out, width*height*4, row-major */
    /* optional outputs, width*height entries each, row-major; NULL to skip */
    int32_t *aov_eye_hit;           /* 2 per pixel */
    uint32_t *aov_hit_hash;         /* in/out chain */
    float *aov_radiance;            /* 4 per pixel */
    uint32_t *aov_rng_draws;        /* in/out */
    uint32_t *aov_rays;             /* in/out */
    uint64_t counters[3];           /* out: rays, light-loop samples, eye-ray-only samples */
    int n_threads;                  /* <=1: single thread */
} orc_job;

/* Renders job->spp samples for every pixel of the rectangle. 0 on success. */
int orc_render(orc_job *job);

/* Single closest-hit query (kernel0.cl:365-478) for known-answer tests.
 * out_id = {type, index}, out_pos/out_normal 3 floats. Returns hit flag. */
int orc_ray_intersect(const bdpt_sphere *spheres, int n_spheres,
                      const bdpt_plane *planes, int n_planes,
                      const bdpt_triangle *triangles, int n_triangles,
                      const bdpt_surface *surfaces,
                      const float origin[3], const float dir[3],
                      int origin_type, int origin_index,
                      int out_id[2], float out_pos[3], float out_normal[3]);

/* Individual intersectors (kernel0.cl:232-361). Return hit flag, write pos. */
int orc_ray_sphere(const bdpt_sphere *s, const float o[3], const float d[3], float pos[3]);
int orc_ray_plane(const bdpt_plane *p, const float o[3], const float d[3], float pos[3]);
int orc_ray_triangle(const bdpt_triangle *t, const float o[3], const float d[3], float pos[3]);

/* Generator (kernel0.cl:106-112): advances *w, returns the float in [0,1]. */
float orc_rng_float(uint32_t *w);
/* random_sample_cone (kernel0.cl:155-185). */
void orc_sample_cone(uint32_t *w, const float n[3], float half_angle, float out[3]);
/* sampleSphereSurface (kernel0.cl:547-569): pos and normal of the sampled point. */
void orc_sample_sphere(uint32_t *w, const bdpt_sphere *s, float pos[3], float nrm[3]);
/* shoot_ray (kernel0.cl:206-217). */
void orc_shoot_ray(const bdpt_camera *cam, float sx, float sy, float out[3]);

/* Built-in instantiations, exported so tests can bound them against libm. */
void orc_sincos(float x, float *s, float *c);
float orc_acosf(float x);
double orc_acos_f64(double x);

#ifdef __cplusplus
}
#endif
#endif

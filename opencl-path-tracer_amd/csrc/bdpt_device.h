/*
 * bdpt_device.h -- per-lane device functions of the BDPT sample: generator,
 * camera ray, light and cone sampling, the three primitive tests and the
 * connection term.  Path state stays in registers; nothing here touches memory
 * except the uniform-address scene reads.
 *
 * Each function states the reference lines whose result it must reproduce
 * (kernel0.cl = /root/reference/src/kernel/kernel0.cl).  Arithmetic follows the
 * numeric contract of include/bdpt_clmath.h; this translation unit is built
 * with -ffp-contract=off.
 */
#ifndef BDPT_DEVICE_H
#define BDPT_DEVICE_H

#include "bdpt_layout.h"
#include "../../include/bdpt.h"
#include "../../include/bdpt_clmath.h"

namespace bdpt {

#define BDPT_DEV __device__ __forceinline__

/* 48271^32 mod 2^32: the generator's stride between samples in STRIDED mode */
constexpr uint32_t kSampleStride = 1820286465u;

BDPT_DEV clm_f3 f3(const float *p) { return clm_make3(p[0], p[1], p[2]); }

/* kernel0.cl:106-112.  (float)UINT_MAX is 2^32, and dividing by it is exact. */
BDPT_DEV float rng_float(uint32_t &w, uint32_t &draws)
{
    w *= 48271u;
    draws++;
    return (float)w * 0x1p-32f;
}

constexpr uint32_t pow_u32_c(uint32_t base, uint32_t e)
{
    uint32_t r = 1u;
    for (uint32_t i = 0; i < e; i++) r *= base;
    return r;
}

BDPT_DEV uint32_t pow_u32(uint32_t base, uint32_t e)
{
    uint32_t r = 1u;
    while (e) {
        if (e & 1u) r *= base;
        base *= base;
        e >>= 1;
    }
    return r;
}

/* kernel0.cl:155-185 random_sample_cone */
BDPT_DEV clm_f3 sample_cone(uint32_t &w, uint32_t &draws, clm_f3 axis, float half_angle)
{
    float p0 = rng_float(w, draws);
    float p1 = rng_float(w, draws);
    float p2 = rng_float(w, draws);
    clm_f3 helper = clm_make3((float)((double)p2 * 2.0 - 1.0), 0.0f, 1.0f);          /* :163 */
    clm_f3 t1 = clm_normalize3(clm_cross3(axis, helper));                            /* :168 */
    clm_f3 t2 = clm_cross3(axis, t1);                                                /* :170 */
    float sweep = p1 * 6.283185307f;                                                 /* :177 */
    float lift = (float)((1.0 - ((double)(clm_acosf(p0) * 2.0f) * CLM_1_PI_F64)) * (double)half_angle);  /* :178 */
    float sl, cl, ss, cs;
    clm_sincos(lift, &sl, &cl);
    clm_sincos(sweep, &ss, &cs);
    clm_f3 ring = clm_mad3(ss, t2, clm_scale3(cs, t1));                              /* :182 */
    return clm_mad3(sl, ring, clm_scale3(cl, axis));
}

/* kernel0.cl:117-153 random_sample_hemisphere */
BDPT_DEV clm_f3 sample_hemisphere(uint32_t &w, uint32_t &draws, clm_f3 axis)
{
    float p0 = rng_float(w, draws);
    float p1 = rng_float(w, draws);
    float p2 = rng_float(w, draws);
    clm_f3 helper = clm_make3((float)((double)p2 * 2.0 - 1.0), 0.0f, 1.0f);          /* :127 */
    clm_f3 t1 = clm_normalize3(clm_cross3(axis, helper));                            /* :132 */
    clm_f3 t2 = clm_cross3(axis, t1);                                                /* :134 */
    float sweep = p1 * 6.283185307f;                                                 /* :141 */
    float lift = p0 * 1.570796326f;                                                  /* :142 */
    float sl, cl, ss, cs;
    clm_sincos(lift, &sl, &cl);
    clm_sincos(sweep, &ss, &cs);
    clm_f3 ring = clm_mad3(ss, t2, clm_scale3(cs, t1));                              /* :150 */
    return clm_mad3(sl, ring, clm_scale3(cl, axis));
}

/* samplers that draw one number before the three of the cone */
BDPT_DEV bool sampler_draws_four(int sampler)
{
    return sampler == BDPT_SAMPLER_REFLECTIVE || sampler == BDPT_SAMPLER_LAMBERT_IMP;
}

/* the samplers other than the live one (kernel0.cl:767): hemisphere, brdf_reflective :492-508, or
 * brdf_lambertian_imp :512-543; `from` is the point the bounce ray leaves */
BDPT_DEV clm_f3 sample_bounce_alt(const bdpt_kargs &K, uint32_t &w, uint32_t &draws, clm_f3 n, clm_f3 incoming, clm_f3 from)
{
    if (K.sampler == BDPT_SAMPLER_HEMISPHERE) return sample_hemisphere(w, draws, n);
    if (K.sampler == BDPT_SAMPLER_LAMBERT_IMP) {
        const float p4 = rng_float(w, draws);                                        /* :513 */
        const bool wide = p4 > K.imp_light_weight;                                   /* :517 */
        const clm_f3 to_light = clm_normalize3(clm_sub3(f3(K.light_c), from));       /* the argument at :945 */
        return sample_cone(w, draws, wide ? n : to_light, wide ? (float)(0.50f * CLM_PI_F64) : K.imp_half_angle);   /* :518-525 */
    }
    const float p4 = rng_float(w, draws);                                            /* :493 */
    const bool mirror = p4 > 0.6f;                                                   /* :500 */
    const clm_f3 r = clm_reflect3(n, incoming);                                      /* :495 */
    const clm_f3 axis = mirror ? r : n;
    return sample_cone(w, draws, axis, mirror ? 3.1415f * 0.02f : 3.1415f * 0.50f);  /* :501, :503 */
}

/* kernel0.cl:547-569 sampleSphereSurface: point and outward normal on the light */
BDPT_DEV void sample_light(uint32_t &w, uint32_t &draws, const bdpt_kargs &K, clm_f3 &pos, clm_f3 &nrm)
{
    float u0 = rng_float(w, draws);
    float u1 = rng_float(w, draws);
    float a0 = (float)((double)u0 * CLM_PI_F64);                                     /* :553 */
    float theta = (float)clm_acos_f64(1.0 - (((double)2.0f * CLM_1_PI_F64) * (double)a0));   /* :554 */
    float phi = (float)((double)(u1 * 2.0f) * CLM_PI_F64);                           /* :556 */
    float st, ct, sp, cp;
    clm_sincos(theta, &st, &ct);
    clm_sincos(phi, &sp, &cp);
    nrm = clm_make3(st * sp, ct, st * cp);                                           /* :558 */
    pos = clm_mad3(K.light_r, nrm, f3(K.light_c));                                   /* :559 */
}

/* kernel0.cl:600-604: the pixel's normalised screen coordinate (fp64 at two sites) */
BDPT_DEV void pixel_coord(const bdpt_kargs &K, int x, int y, float &xt, float &yt)
{
    float xf = clm_div((float)x, (float)K.width);
    float yf = clm_div((float)y, (float)K.height);
    xt = (float)(((double)xf * 2.0 - 1.0) * (double)K.aspect);
    yt = (float)((double)yf * 2.0 - 1.0);
}

/* kernel0.cl:190-217: sub-pixel jitter (2 draws) and the camera ray */
BDPT_DEV clm_f3 camera_ray(uint32_t &w, uint32_t &draws, const bdpt_kargs &K, float xt, float yt)
{
    float jx = rng_float(w, draws);
    float jy = rng_float(w, draws);
    float sx = xt + jx * K.delta_x;                                                  /* :198 */
    float sy = yt + jy * K.delta_y;                                                  /* :199 */
    clm_f3 look = f3(K.cam_look);
    clm_f3 cy = clm_mad3(sy, f3(K.cam_up), look);                                    /* :211 */
    clm_f3 cx = clm_mad3(sx, f3(K.cam_right), look);                                 /* :212 */
    return clm_normalize3(clm_add3(cx, cy));                                         /* :215 */
}

/*
 * Scene records are read with a wave-uniform address.  Reading them through the
 * constant address space tells the compiler the bytes never change during the
 * launch, so it always uses scalar loads (s_load_dwordx4/x8/x16 through the
 * constant cache, results in SGPRs that the VALU consumes directly) whatever
 * other code surrounds the load.
 */
typedef float bdpt_v4f __attribute__((ext_vector_type(4)));
typedef const __attribute__((address_space(4))) bdpt_v4f *const_v4f_ptr;
BDPT_DEV float4 uload(const float4 *base, size_t i)
{
    const bdpt_v4f v = ((const_v4f_ptr)(uintptr_t)base)[i];
    return make_float4(v.x, v.y, v.z, v.w);
}

/* closest-hit record; type < 0 is a miss */
struct Hit {
    int type, idx;
    clm_f3 pos;
    clm_f3 n;          /* shading normal, flipped to face the ray (kernel0.cl:461-473) */
    uint32_t surf;
};

/*
 * Running best of the exhaustive search.  The reference keeps the hit whose
 * distance sqrt(d2) is strictly smaller than the best so far, starting from
 * 10000 (kernel0.cl:374, :398-400, :422-424, :441-442), so ties go to the
 * earliest (type, index).  sqrt is monotone, so the same decisions can be made
 * on squared distances and the square root is only needed when two squared
 * distances are so close that their roots might round to the same float:
 *   d2 >= best.d2             -> sqrt(d2) >= sqrt(best.d2): rejected, exactly as the reference
 *   d2 <  best.d2 * (1-1e-6)  -> the roots differ by > 3 ulp: accepted without a root
 *   in between (rare)         -> both roots are taken and compared, as the reference does.
 * best.d2 starts at 1e8 = 10000^2, whose root is exactly the reference's 10000.
 */
struct Best {
    float d2, d2_lo;
    int type, idx;
    clm_f3 pos;
};

BDPT_DEV void best_init(Best &b)
{
    b.d2 = 1.0e8f;
    b.d2_lo = 9.99999e7f;
    b.type = -1;
    b.idx = -1;
    b.pos = clm_make3(0.0f, 0.0f, 0.0f);
}

BDPT_DEV void best_consider(Best &b, clm_f3 p, clm_f3 o, int type, int i)
{
    const clm_f3 v = clm_sub3(p, o);
    const float d2 = clm_dot3(v, v);                       /* distance(p, o)^2 before the root */
    bool take = d2 < b.d2_lo;
    const bool close_call = !take && d2 < b.d2;
    /* a wave-uniform branch on the vote keeps the two roots out of the common path
     * (a plain 'if' is flattened into straight-line code by the compiler) */
    if (__builtin_amdgcn_ballot_w64(close_call) != 0) {
        asm volatile("");                /* not speculatable: keeps this block a real branch */
        if (close_call) take = clm_sqrt(d2) < clm_sqrt(b.d2);
    }
    if (take) {
        b.d2 = d2;
        b.d2_lo = d2 > 1.0e-30f ? d2 * 0.999999f : 0.0f;  /* below that every later candidate takes the exact branch */
        b.type = type;
        b.idx = i;
        b.pos = p;
    }
}

/* kernel0.cl:314-361 + :396-406.  s = (c.xyz, r*r) */
BDPT_DEV void test_sphere(Best &b, float4 s, int i, clm_f3 o, clm_f3 d)
{
    clm_f3 oc = clm_sub3(o, clm_make3(s.x, s.y, s.z));
    float dot1 = clm_dot3(d, oc);                                                    /* :331 */
    float d2 = clm_dot3(oc, oc);                                                     /* :336 */
    float disc = dot1 * dot1 - d2 + s.w;                                             /* :337 */
    /* one branch for the two culls: a wavefront nearly always has a lane that passes the first, and each
     * branch level costs about as much as the six operations it would skip */
    if (dot1 > 0 || disc < 0) return;                                                /* :333, :341 */
    float tA = -dot1 - clm_sqrt(disc);                                               /* :344-347 */
    if (tA < 0) return;                                                              /* :350 */
    clm_f3 p = clm_mad3(tA, d, o);                                                   /* :354 */
    best_consider(b, p, o, 0, i);                                                    /* :398-406 */
}

/* kernel0.cl:232-271 + :420-431.  q0 = (n.xyz,-), q1 = (pt.xyz,-) */
BDPT_DEV void test_plane(Best &b, float4 q0, float4 q1, int i, clm_f3 o, clm_f3 d)
{
    clm_f3 n = clm_make3(q0.x, q0.y, q0.z);
    float angle = clm_dot3(d, n);                                                    /* :249 */
    if (angle >= 0) return;                                                          /* :252 */
    float temp = clm_dot3(clm_sub3(clm_make3(q1.x, q1.y, q1.z), o), n);              /* :256 */
    float desc = clm_dot3(clm_scale3(temp, n), d);                                   /* :257 */
    if (!(desc > 0)) return;                                                         /* :258 */
    clm_f3 p = clm_mad3(clm_div(temp, angle), d, o);                                 /* :263-264 */
    best_consider(b, p, o, 1, i);                                                    /* :422-430 */
}

/* kernel0.cl:273-312 + :440-449 on the pre-laid-out record (bdpt_layout.h) */
BDPT_DEV void test_triangle(Best &b, float4 q0, float4 q1, float4 q2, float4 q3, float4 q4, float e2z,
                            int i, clm_f3 o, clm_f3 d)
{
    clm_f3 n = clm_make3(q0.x, q0.y, q0.z);
    float val = clm_dot3(d, n);                                                      /* :286 */
    if (val > 0) return;                                                             /* :288 */
    clm_f3 a = clm_make3(q0.w, q1.x, q1.y);
    float t = clm_div(clm_dot3(clm_sub3(a, o), n), val);                             /* :291 */
    if (t < 0) return;                                                               /* :292 */
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
    best_consider(b, p, o, 2, i);                                                    /* :441-448 */
}

/*
 * test_triangle for rays that scatter in all directions (connection and bounce
 * rays).  Two changes of form, none of result:
 *  - the back-face cull (:288) and the behind-the-origin cull (:292) are taken
 *    together after the division: for such rays some lane of the wavefront
 *    passes :288 for practically every triangle, so a separate branch there
 *    only costs a compare and an exec-mask update per triangle;
 *  - the origin primitive (:436) is not skipped up front but refused when it
 *    would be accepted, which moves that compare out of the per-triangle path
 *    into the one or two triangles per ray that are actually hit.
 */
BDPT_DEV void test_triangle_scattered(Best &b, float4 q0, float4 q1, float4 q2, float4 q3, float4 q4, float e2z, int i,
                                      int skip, clm_f3 o, clm_f3 d)
{
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
    if (i != skip) best_consider(b, p, o, 2, i);                                     /* :436, :441-448 */
}

/*
 * test_triangle_scattered without branches, for the per-lane second stage of the two-stage kernel, where
 * nearly every lane that runs the test has a hit: all culls become one predicate and the running best is
 * updated by selects.  `wanted` = this lane has a triangle to test at all.
 */
BDPT_DEV void test_triangle_flat(Best &b, float4 q0, float4 q1, float4 q2, float4 q3, float4 q4, float e2z, int i,
                                 bool wanted, clm_f3 o, clm_f3 d)
{
    clm_f3 n = clm_make3(q0.x, q0.y, q0.z);
    float val = clm_dot3(d, n);                                                      /* :286 */
    clm_f3 a = clm_make3(q0.w, q1.x, q1.y);
    float t = clm_div(clm_dot3(clm_sub3(a, o), n), val);                             /* :291 */
    clm_f3 p = clm_mad3(t, d, o);                                                    /* :295 */
    clm_f3 e0 = clm_make3(q1.z, q1.w, q2.x);
    clm_f3 c = clm_make3(q2.y, q2.z, q2.w);
    clm_f3 e1 = clm_make3(q3.x, q3.y, q3.z);
    clm_f3 bb = clm_make3(q3.w, q4.x, q4.y);
    clm_f3 e2 = clm_make3(q4.z, q4.w, e2z);
    float dA = clm_dot3(clm_cross3(clm_sub3(p, a), e0), n);                          /* :298 */
    float dB = clm_dot3(clm_cross3(clm_sub3(p, c), e1), n);                          /* :299 */
    float dC = clm_dot3(clm_cross3(clm_sub3(p, bb), e2), n);                         /* :300 */
    const bool hit = wanted && !(val > 0) && !(t < 0) && !(dA < 0) && !(dB < 0) && !(dC < 0);   /* :288, :292, :304 */
    const clm_f3 v = clm_sub3(p, o);
    const float d2 = clm_dot3(v, v);
    bool take = hit && d2 < b.d2_lo;
    const bool close_call = hit && !take && d2 < b.d2;
    if (__builtin_amdgcn_ballot_w64(close_call) != 0) {
        asm volatile("");
        if (close_call) take = clm_sqrt(d2) < clm_sqrt(b.d2);
    }
    b.d2 = take ? d2 : b.d2;
    b.d2_lo = take ? (d2 > 1.0e-30f ? d2 * 0.999999f : 0.0f) : b.d2_lo;
    b.type = take ? 2 : b.type;
    b.idx = take ? i : b.idx;
    b.pos.x = take ? p.x : b.pos.x;
    b.pos.y = take ? p.y : b.pos.y;
    b.pos.z = take ? p.z : b.pos.z;
}

/* kernel0.cl:452-473: surface id, shading normal of the winner, flip toward the ray */
BDPT_DEV Hit finish_hit(const bdpt_dev_scene &S, const Best &b, clm_f3 d)
{
    Hit h;
    h.type = b.type;
    h.idx = b.idx;
    h.pos = b.pos;
    h.n = clm_make3(0.0f, 0.0f, 0.0f);
    h.surf = 0;
    if (b.type == 0) {
        float4 s = S.sph[b.idx];
        h.n = clm_normalize3(clm_sub3(clm_make3(s.x, s.y, s.z), b.pos));             /* :462 */
        h.surf = S.sph_surf[b.idx];
    } else if (b.type == 1) {
        float4 q = S.pln[b.idx * BDPT_PLN_F4];
        h.n = clm_make3(q.x, q.y, q.z);                                              /* :464 */
        h.surf = S.pln_surf[b.idx];
    } else if (b.type == 2) {
        float4 q = S.tri[b.idx * BDPT_TRI_F4 + 5];
        h.n = clm_make3(q.y, q.z, q.w);                                              /* :467 */
        h.surf = S.tri_surf[b.idx];
    }
    if (clm_dot3(d, h.n) > 0) h.n = clm_neg3(h.n);                                   /* :471 */
    return h;
}

/* id word chained into the hit hash: type<<28 | index, all ones on a miss */
BDPT_DEV uint32_t hit_id(int type, int idx)
{
    return type < 0 ? 0xFFFFFFFFu : (((uint32_t)type << 28) | (uint32_t)idx);
}

/* 1 / preFactor of light-subpath vertex lti (kernel0.cl:694, :745): the first entries come from the host's table */
BDPT_DEV float inv_pre_factor(const bdpt_kargs &K, int lti, float pre)
{
    return lti < BDPT_INV_PRE_N ? K.inv_pre[lti] : clm_div(1.0f, pre);
}

/*
 * kernel0.cl:712-744: contribution of one light-subpath vertex to the eye
 * vertex, all four channels.  `vis` is the closest hit of the ray from the
 * light vertex toward the eye vertex; inv_imp is 1 / preFactor (:741), see
 * inv_pre_factor().
 */
BDPT_DEV void connection(const bdpt_kargs &K, const Hit &eye, const Hit &vis, clm_f3 vert_pos, clm_f3 vert_n,
                         clm_f3 test_dir, const float acc[4], float inv_imp, float out[4])
{
    out[0] = out[1] = out[2] = out[3] = 0.0f;
    const clm_f3 gap = clm_sub3(eye.pos, vis.pos);
    /* one branch for the three conditions: each is a few operations, a branch level costs about as much */
    if (eye.type != vis.type || eye.idx != vis.idx ||                                /* :712 */
        !(clm_dot3(gap, gap) < K.same_point_d2) ||                                   /* :714, length < 0.0001f without the root */
        !(clm_dot3(eye.n, vis.n) > 0)) return;
    clm_f3 pv = clm_sub3(vis.pos, vert_pos);                                         /* :716 */
    float r2 = pv.x * pv.x + pv.y * pv.y + pv.z * pv.z;
    float sf = (K.falloff == BDPT_FALLOFF_INV_SQ_PLUS_C) ? clm_div(1.0f, K.falloff_c + r2)
                                                         : clm_div(1.0f, r2);        /* :719 */
    float df = clm_dot3(eye.n, test_dir) * clm_dot3(vert_n, test_dir);               /* :732 */
    df *= clm_sign(df);                                                              /* :733 */
#pragma unroll
    for (int k = 0; k < 4; k++) out[k] = acc[k] * sf * df * inv_imp;                 /* :741 */
}

/*
 * Order in which a launch takes the samples of a pixel.
 *
 * STREAM mode and MEAN accumulation are sequential by definition.  In STRIDED
 * mode every sample has its own start state, and the Russian-roulette draws of
 * a sample (the draw after each cone sample, kernel0.cl:787) depend on that
 * state alone, so the number of light-subpath vertices roulette allows -- its
 * class, 1..max_l -- is known before the sample is traced.  A wavefront walks
 * the light subpaths of its 64 pixels in lock-step and pays for the longest of
 * them; taking samples class by class (longest first, sample index ascending
 * within a class) lets all lanes run subpaths of equal length at the same time.
 * With SUM accumulation this order is simply the order of the additions; it is
 * part of the definition of a STRIDED + SUM launch (the CPU oracle adds in the
 * same order) and changes no sample.
 */
struct SampleCursor {
    int cls;                /* class being collected (reorder); negative: this pixel takes its samples in index order */
    int head;               /* next sample index to examine */
    uint32_t st_head;       /* STRIDED: start state of sample head */
    uint32_t pending;       /* reorder: bit i set = sample head-1-i is of class cls and not traced yet */
};

/* 48271^e mod 2^32 */
BDPT_DEV constexpr uint32_t lcg_pow(int e)
{
    uint32_t m = 1u;
    for (int i = 0; i < e; i++) m *= 48271u;
    return m;
}

/* bit k set = roulette draw k of the sample (k = 0 .. N-1) ends the light subpath there; x = that draw of k = 0 */
template <int N>
BDPT_DEV uint32_t roulette_stops(uint32_t x, uint32_t step, uint32_t fate_above)
{
    uint32_t stops = 0u;
#pragma unroll
    for (int k = 0; k < N; k++) {
        stops |= x > fate_above ? (1u << k) : 0u;                                    /* :788 */
        x *= step;
    }
    return stops;
}

/* Number of light-subpath vertices roulette allows a sample; same cost for every lane.  A sample
 * draws 8 numbers before its first roulette draw (1 discarded, 2 pixel jitter, 2 light point, 3 first
 * cone sample) and 3 between one roulette draw and the next, so roulette draw k is the generator's
 * output number 9+4k: the start state times 48271^(9+4k).
 * N > 0: straight-line code that looks at N draws, for max_l <= N + 1 (draws past max_l - 1 set bits
 * above the one that caps the walk and are never seen); N = 0: a loop, any max_l. */
template <int N>
BDPT_DEV int roulette_class(const bdpt_kargs &K, uint32_t start_state)
{
    const bool five = sampler_draws_four(K.sampler);           /* one more draw per vertex: 10 + 5k */
    const uint32_t step = five ? lcg_pow(5) : lcg_pow(4);
    const uint32_t x = start_state * (five ? lcg_pow(10) : lcg_pow(9));
    uint32_t stops = 0u;
    if (N > 0) {
        stops = roulette_stops<N>(x, step, K.fate_above);
    } else {
        uint32_t y = x;
        for (int k = 0; k + 1 < K.max_l && k < 31; k++) {
            stops |= y > K.fate_above ? (1u << k) : 0u;
            y *= step;
        }
    }
    if (K.max_l <= 32) stops |= 1u << (K.max_l - 1);           /* the walk never has more than max_l vertices */
    return stops ? __builtin_ctz(stops) + 1 : K.max_l;
}

/* start state of the first sample of this launch (STRIDED) */
BDPT_DEV uint32_t cursor_first_state(const bdpt_kargs &K, uint32_t seed)
{
    return (K.rng_mode == BDPT_RNG_STRIDED) ? seed * pow_u32(kSampleStride, K.first_sample) : 0u;
}

BDPT_DEV void cursor_init(SampleCursor &c, const bdpt_kargs &K, uint32_t seed)
{
    c.cls = K.max_l;
    c.head = 0;
    c.st_head = cursor_first_state(K, seed);
    c.pending = 0u;
}

/* multiplicative inverse of kSampleStride mod 2^32, raised to 2^j */
BDPT_DEV constexpr uint32_t stride_inv_pow2(int j)
{
    uint32_t inv = kSampleStride;                       /* Newton: inv <- inv * (2 - stride * inv) doubles the correct low bits */
    for (int i = 0; i < 5; i++) inv *= 2u - kSampleStride * inv;
    for (int i = 0; i < j; i++) inv *= inv;
    return inv;
}
static_assert(stride_inv_pow2(0) * kSampleStride == 1u, "stride inverse");

/* powers 0..32 of that inverse */
struct StrideInvPow { uint32_t v[33]; };
BDPT_DEV constexpr StrideInvPow make_stride_inv_pow()
{
    StrideInvPow t{};
    uint32_t p = 1u;
    for (int i = 0; i < 33; i++) { t.v[i] = p; p *= stride_inv_pow2(0); }
    return t;
}
static __device__ const StrideInvPow kStrideInvPow = make_stride_inv_pow();

/*
 * The reordered case of cursor_next.  A lane finds its next sample of the current class by classifying
 * sample indices ahead of the last one it traced.  The lanes of a wavefront do this together, one index
 * per lane per round, until every lane has a sample to trace; a lane that already has one keeps
 * classifying (up to 32 indices past its oldest untraced sample) and remembers what it finds in
 * `pending`, so that it seldom is the lane the others wait for.  A round is selects and two votes;
 * what depends on max_l alone is decided outside the loop (N as in roulette_class).
 */
template <int N>
BDPT_DEV bool cursor_rounds(SampleCursor &c, const bdpt_kargs &K, const uint32_t *seed_w, uint32_t &start)
{
    for (;;) {
        const bool starved = c.pending == 0u && c.cls >= 1;
        if (__builtin_amdgcn_ballot_w64(starved) == 0) break;
        const bool next_class = starved && c.head >= K.spp;  /* no more samples of this class */
        if (__builtin_amdgcn_ballot_w64(next_class) != 0) {
            asm volatile("");                                /* a real branch: a handful of times per launch */
            if (next_class) {
                c.cls--;
                c.head = 0;
                c.st_head = cursor_first_state(K, *seed_w);
            }
            continue;                                        /* the lane may have run out of classes */
        }
        const bool advance = c.cls >= 1 && c.head < K.spp && (c.pending >> 31) == 0u;
        const uint32_t found = roulette_class<N>(K, c.st_head) == c.cls ? 1u : 0u;
        c.pending = advance ? ((c.pending << 1) | found) : c.pending;
        c.head += advance ? 1 : 0;
        c.st_head = advance ? c.st_head * kSampleStride : c.st_head;
    }
    if (c.cls < 1) return false;
    const int i = 31 - __builtin_clz(c.pending);             /* oldest pending sample: index head-1-i */
    c.pending &= ~(1u << i);
    start = c.st_head * kStrideInvPow.v[i + 1];              /* back from sample head to sample head-1-i */
    return true;
}

/*
 * false when the launch has no more samples for this pixel; otherwise start = start state (STRIDED).
 * seed_w = the pixel's seed word, wanted again when a lane restarts at sample 0 for the next class.
 */
BDPT_DEV bool cursor_next(SampleCursor &c, const bdpt_kargs &K, const uint32_t *seed_w, uint32_t &start)
{
    if (!K.reorder || c.cls < 0) {
        if (c.head >= K.spp) return false;
        start = c.st_head;
        c.st_head *= kSampleStride;
        c.head++;
        return true;
    }
    if (K.max_l <= 5) return cursor_rounds<4>(c, K, seed_w, start);
    if (K.max_l <= 7) return cursor_rounds<6>(c, K, seed_w, start);
    return cursor_rounds<0>(c, K, seed_w, start);
}

/* one link of the per-pixel hit-id chain (the additive constant keeps a run of
 * zero ids from pinning the chain at zero) */
BDPT_DEV uint32_t hash_step(uint32_t h, uint32_t id) { return (h ^ id) * 16777619u + 0x9E3779B9u; }

}  // namespace bdpt

#endif

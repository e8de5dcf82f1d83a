/*
 * bdpt_device2.h -- the per-lane functions of bdpt_device.h for TWO paths per
 * lane, one in each half of a 64-bit register pair.
 *
 * Why: gfx950 issues one FP32 vector instruction per SIMD every 4 cycles
 * whether it is v_fma_f32 or v_pk_fma_f32, and the packed form does two lanes'
 * worth of work per lane (the 157 TFLOP/s vector peak is the packed rate; plain
 * FP32 instructions top out at half of it).  A lane that carries two
 * independent paths has a twin for every add, mul and fma, so the whole
 * geometric part of the sample packs 2-wide.
 *
 * Every function here performs, per component, exactly the operations of its
 * one-path counterpart (same order, same roundings, same comparison forms), so
 * results stay bit-identical to the numeric contract (include/bdpt_clmath.h).
 * Data-dependent branches of the one-path code become per-component predicates
 * (B2); a block is skipped only when neither component needs it.
 */
#ifndef BDPT_DEVICE2_H
#define BDPT_DEVICE2_H

#include "bdpt_device.h"

namespace bdpt {

typedef float f2 __attribute__((ext_vector_type(2)));
typedef int i2 __attribute__((ext_vector_type(2)));
typedef unsigned int u2 __attribute__((ext_vector_type(2)));

/* one predicate per component */
struct B2 {
    bool a, b;
};
BDPT_DEV B2 mk2(bool a, bool b) { B2 r; r.a = a; r.b = b; return r; }
BDPT_DEV bool any2(B2 c) { return c.a || c.b; }
BDPT_DEV B2 and2(B2 x, B2 y) { return mk2(x.a && y.a, x.b && y.b); }
BDPT_DEV B2 andnot2(B2 x, B2 y) { return mk2(x.a && !y.a, x.b && !y.b); }
BDPT_DEV B2 gt2(f2 x, f2 y) { return mk2(x.x > y.x, x.y > y.y); }
BDPT_DEV B2 lt2(f2 x, f2 y) { return mk2(x.x < y.x, x.y < y.y); }
BDPT_DEV B2 ge2(f2 x, f2 y) { return mk2(x.x >= y.x, x.y >= y.y); }
BDPT_DEV B2 not2(B2 x) { return mk2(!x.a, !x.b); }

BDPT_DEV f2 splat(float s) { f2 r; r.x = s; r.y = s; return r; }
BDPT_DEV f2 sel2(B2 c, f2 t, f2 f) { f2 r; r.x = c.a ? t.x : f.x; r.y = c.b ? t.y : f.y; return r; }
BDPT_DEV i2 sel2(B2 c, i2 t, i2 f) { i2 r; r.x = c.a ? t.x : f.x; r.y = c.b ? t.y : f.y; return r; }
BDPT_DEV u2 sel2(B2 c, u2 t, u2 f) { u2 r; r.x = c.a ? t.x : f.x; r.y = c.b ? t.y : f.y; return r; }

BDPT_DEV f2 fma2(f2 a, f2 b, f2 c) { return __builtin_elementwise_fma(a, b, c); }
BDPT_DEV f2 sqrt2(f2 x) { f2 r; r.x = clm_sqrt(x.x); r.y = clm_sqrt(x.y); return r; }
BDPT_DEV f2 div2(f2 a, f2 b) { f2 r; r.x = clm_div(a.x, b.x); r.y = clm_div(a.y, b.y); return r; }

/* two 3-vectors */
struct V3 {
    f2 x, y, z;
};
BDPT_DEV V3 mk3(f2 x, f2 y, f2 z) { V3 r; r.x = x; r.y = y; r.z = z; return r; }
BDPT_DEV V3 splat3(clm_f3 v) { return mk3(splat(v.x), splat(v.y), splat(v.z)); }
BDPT_DEV clm_f3 comp3(const V3 &v, int k) { return k == 0 ? clm_make3(v.x.x, v.y.x, v.z.x) : clm_make3(v.x.y, v.y.y, v.z.y); }
BDPT_DEV V3 pack3(clm_f3 a, clm_f3 b)
{
    V3 r;
    r.x.x = a.x; r.y.x = a.y; r.z.x = a.z;
    r.x.y = b.x; r.y.y = b.y; r.z.y = b.z;
    return r;
}
BDPT_DEV V3 add3(V3 a, V3 b) { return mk3(a.x + b.x, a.y + b.y, a.z + b.z); }
BDPT_DEV V3 sub3(V3 a, V3 b) { return mk3(a.x - b.x, a.y - b.y, a.z - b.z); }
BDPT_DEV V3 neg3(V3 a) { return mk3(-a.x, -a.y, -a.z); }
BDPT_DEV V3 scale3(f2 s, V3 a) { return mk3(s * a.x, s * a.y, s * a.z); }
BDPT_DEV V3 mad3(f2 s, V3 b, V3 a) { return mk3(fma2(s, b.x, a.x), fma2(s, b.y, a.y), fma2(s, b.z, a.z)); }
BDPT_DEV V3 sel3(B2 c, V3 t, V3 f) { return mk3(sel2(c, t.x, f.x), sel2(c, t.y, f.y), sel2(c, t.z, f.z)); }
BDPT_DEV f2 dot3(V3 a, V3 b) { return fma2(a.z, b.z, fma2(a.y, b.y, a.x * b.x)); }
BDPT_DEV V3 cross3(V3 a, V3 b)
{
    return mk3(fma2(a.y, b.z, -(a.z * b.y)), fma2(a.z, b.x, -(a.x * b.z)), fma2(a.x, b.y, -(a.y * b.x)));
}
BDPT_DEV V3 normalize3(V3 a)
{
    f2 inv = div2(splat(1.0f), sqrt2(dot3(a, a)));
    return scale3(inv, a);
}

/* clm_sincos, two angles */
BDPT_DEV void sincos2(f2 x, f2 &s_out, f2 &c_out)
{
    f2 t = x * splat(0x1.45f306p-1f);
    f2 kf;
    kf.x = __builtin_rintf(t.x);
    kf.y = __builtin_rintf(t.y);
    f2 r = fma2(-kf, splat(0x1.921fb6p+0f), x);
    r = fma2(-kf, splat(-0x1.777a5cp-25f), r);
    r = fma2(-kf, splat(-0x1.ee59dap-50f), r);
    f2 z = r * r;
    f2 ps = fma2(splat(-1.9515295891e-4f), z, splat(8.3321608736e-3f));
    ps = fma2(ps, z, splat(-1.6666654611e-1f));
    f2 sn = fma2(ps * z, r, r);
    f2 pc = fma2(splat(2.443315711809948e-5f), z, splat(-1.388731625493765e-3f));
    pc = fma2(pc, z, splat(4.166664568298827e-2f));
    f2 cs = fma2(pc, z * z, fma2(splat(-0.5f), z, splat(1.0f)));
    const int qa = ((int)kf.x) & 3, qb = ((int)kf.y) & 3;
    f2 s, c;
    s.x = (qa & 1) ? cs.x : sn.x;  c.x = (qa & 1) ? sn.x : cs.x;
    s.y = (qb & 1) ? cs.y : sn.y;  c.y = (qb & 1) ? sn.y : cs.y;
    if (qa == 1 || qa == 2) c.x = -c.x;
    if (qb == 1 || qb == 2) c.y = -c.y;
    if (qa >= 2) s.x = -s.x;
    if (qb >= 2) s.y = -s.y;
    s_out = s;
    c_out = c;
}

/* clm_acosf, two arguments: the three ranges of the one-argument form as selects */
BDPT_DEV f2 acosf2(f2 x)
{
    const B2 hi = gt2(x, splat(0.5f)), lo = lt2(x, splat(-0.5f));
    const B2 outer = mk2(hi.a || lo.a, hi.b || lo.b);
    const f2 zo = splat(0.5f) * sel2(hi, splat(1.0f) - x, splat(1.0f) + x);
    const f2 z = sel2(outer, zo, x * x);
    const f2 s = sel2(outer, sqrt2(z), x);
    f2 p = fma2(splat(4.2163199048e-2f), z, splat(2.4181311049e-2f));
    p = fma2(p, z, splat(4.5470025998e-2f));
    p = fma2(p, z, splat(7.4953002686e-2f));
    p = fma2(p, z, splat(1.6666752422e-1f));
    const f2 core = fma2(s * z, p, s);
    const f2 two_core = splat(2.0f) * core;
    const f2 r_lo = splat(0x1.921fb6p+1f) - two_core;
    const f2 r_mid = splat(0x1.921fb6p+0f) - core;
    return sel2(hi, two_core, sel2(lo, r_lo, r_mid));
}

/* kernel0.cl:106-112 for the components in `act`; the others keep their state */
BDPT_DEV f2 rng_float2(u2 &w, u2 &draws, B2 act)
{
    const u2 nw = w * 48271u;
    w = sel2(act, nw, w);
    draws.x += act.a ? 1u : 0u;
    draws.y += act.b ? 1u : 0u;
    f2 r;
    r.x = (float)nw.x * 0x1p-32f;
    r.y = (float)nw.y * 0x1p-32f;
    return r;
}

/* kernel0.cl:155-185 random_sample_cone, two axes */
BDPT_DEV V3 sample_cone2(u2 &w, u2 &draws, B2 act, V3 axis, float half_angle)
{
    const f2 p0 = rng_float2(w, draws, act);
    const f2 p1 = rng_float2(w, draws, act);
    const f2 p2 = rng_float2(w, draws, act);
    f2 hx;
    hx.x = (float)((double)p2.x * 2.0 - 1.0);                                        /* :163, fp64 */
    hx.y = (float)((double)p2.y * 2.0 - 1.0);
    const V3 helper = mk3(hx, splat(0.0f), splat(1.0f));
    const V3 t1 = normalize3(cross3(axis, helper));                                  /* :168 */
    const V3 t2 = cross3(axis, t1);                                                  /* :170 */
    const f2 sweep = p1 * splat(6.283185307f);                                       /* :177 */
    const f2 ac2 = acosf2(p0) * splat(2.0f);
    f2 lift;                                                                         /* :178, fp64 */
    lift.x = (float)((1.0 - ((double)ac2.x * CLM_1_PI_F64)) * (double)half_angle);
    lift.y = (float)((1.0 - ((double)ac2.y * CLM_1_PI_F64)) * (double)half_angle);
    f2 sl, cl, ss, cs;
    sincos2(lift, sl, cl);
    sincos2(sweep, ss, cs);
    const V3 ring = mad3(ss, t2, scale3(cs, t1));                                    /* :182 */
    return mad3(sl, ring, scale3(cl, axis));
}

/* kernel0.cl:547-569 sampleSphereSurface, two samples (fp64 sites evaluated per component) */
BDPT_DEV void sample_light2(u2 &w, u2 &draws, B2 act, const bdpt_kargs &K, V3 &pos, V3 &nrm)
{
    const f2 u0 = rng_float2(w, draws, act);
    const f2 u1 = rng_float2(w, draws, act);
    f2 theta, phi;
    theta.x = (float)clm_acos_f64(1.0 - (((double)2.0f * CLM_1_PI_F64) * (double)(float)((double)u0.x * CLM_PI_F64)));
    theta.y = (float)clm_acos_f64(1.0 - (((double)2.0f * CLM_1_PI_F64) * (double)(float)((double)u0.y * CLM_PI_F64)));
    const f2 u1x2 = u1 * splat(2.0f);
    phi.x = (float)((double)u1x2.x * CLM_PI_F64);
    phi.y = (float)((double)u1x2.y * CLM_PI_F64);
    f2 st, ct, sp, cp;
    sincos2(theta, st, ct);
    sincos2(phi, sp, cp);
    nrm = mk3(st * sp, ct, st * cp);                                                 /* :558 */
    pos = mad3(splat(K.light_r), nrm, splat3(f3(K.light_c)));                        /* :559 */
}

/* kernel0.cl:190-217: jitter and camera ray for two pixels */
BDPT_DEV V3 camera_ray2(u2 &w, u2 &draws, B2 act, const bdpt_kargs &K, f2 xt, f2 yt)
{
    const f2 jx = rng_float2(w, draws, act);
    const f2 jy = rng_float2(w, draws, act);
    const f2 sx = xt + jx * splat(K.delta_x);                                        /* :198 */
    const f2 sy = yt + jy * splat(K.delta_y);                                        /* :199 */
    const V3 look = splat3(f3(K.cam_look));
    const V3 cy = mad3(sy, splat3(f3(K.cam_up)), look);                              /* :211 */
    const V3 cx = mad3(sx, splat3(f3(K.cam_right)), look);                           /* :212 */
    return normalize3(add3(cx, cy));                                                 /* :215 */
}

/* ---- exhaustive closest hit for two rays ---------------------------------- */

struct Hit2 {
    i2 type, idx;
    V3 pos, n;
    u2 surf;
};

struct Best2 {
    f2 d2, d2_lo;
    i2 type, idx;
    V3 pos;
};

BDPT_DEV void best2_init(Best2 &b)
{
    b.d2 = splat(1.0e8f);
    b.d2_lo = splat(9.99999e7f);
    b.type.x = b.type.y = -1;
    b.idx.x = b.idx.y = -1;
    b.pos = mk3(splat(0.0f), splat(0.0f), splat(0.0f));
}

/* best_consider (bdpt_device.h) for the components in `ok` */
BDPT_DEV void best2_consider(Best2 &b, B2 ok, V3 p, V3 o, int type, int i)
{
    const V3 v = sub3(p, o);
    const f2 d2 = dot3(v, v);
    B2 take = and2(ok, lt2(d2, b.d2_lo));
    const B2 close_call = and2(andnot2(ok, take), lt2(d2, b.d2));
    if (__builtin_amdgcn_ballot_w64(any2(close_call)) != 0) {
        asm volatile("");                /* not speculatable: keeps the two roots out of the common path */
        if (close_call.a) take.a = clm_sqrt(d2.x) < clm_sqrt(b.d2.x);
        if (close_call.b) take.b = clm_sqrt(d2.y) < clm_sqrt(b.d2.y);
    }
    if (any2(take)) {
        f2 lo;
        lo.x = d2.x > 1.0e-30f ? d2.x * 0.999999f : 0.0f;
        lo.y = d2.y > 1.0e-30f ? d2.y * 0.999999f : 0.0f;
        i2 ty, ix;
        ty.x = ty.y = type;
        ix.x = ix.y = i;
        b.d2 = sel2(take, d2, b.d2);
        b.d2_lo = sel2(take, lo, b.d2_lo);
        b.type = sel2(take, ty, b.type);
        b.idx = sel2(take, ix, b.idx);
        b.pos = sel3(take, p, b.pos);
    }
}

/* kernel0.cl:314-361 + :396-406 */
BDPT_DEV void test_sphere2(Best2 &b, B2 en, float4 s, int i, V3 o, V3 d)
{
    const V3 oc = mk3(o.x - s.x, o.y - s.y, o.z - s.z);
    const f2 dot1 = dot3(d, oc);                                                     /* :331 */
    B2 ok = andnot2(en, gt2(dot1, splat(0.0f)));                                     /* :333 */
    if (!any2(ok)) return;
    const f2 dd = dot3(oc, oc);                                                      /* :336 */
    const f2 disc = dot1 * dot1 - dd + s.w;                                          /* :337 */
    ok = andnot2(ok, lt2(disc, splat(0.0f)));                                        /* :341 */
    if (!any2(ok)) return;
    const f2 tA = -dot1 - sqrt2(disc);                                               /* :344-347 */
    ok = andnot2(ok, lt2(tA, splat(0.0f)));                                          /* :350 */
    if (!any2(ok)) return;
    best2_consider(b, ok, mad3(tA, d, o), o, 0, i);                                  /* :354, :398-406 */
}

/* kernel0.cl:232-271 + :420-431 */
BDPT_DEV void test_plane2(Best2 &b, B2 en, float4 q0, float4 q1, int i, V3 o, V3 d)
{
    const f2 angle = fma2(d.z, splat(q0.z), fma2(d.y, splat(q0.y), d.x * q0.x));     /* :249 */
    B2 ok = andnot2(en, ge2(angle, splat(0.0f)));                                    /* :252 */
    if (!any2(ok)) return;
    const V3 n = mk3(splat(q0.x), splat(q0.y), splat(q0.z));
    const V3 po = mk3(q1.x - o.x, q1.y - o.y, q1.z - o.z);
    const f2 temp = dot3(po, n);                                                     /* :256 */
    const f2 desc = dot3(scale3(temp, n), d);                                        /* :257 */
    ok = and2(ok, gt2(desc, splat(0.0f)));                                           /* :258 */
    if (!any2(ok)) return;
    best2_consider(b, ok, mad3(div2(temp, angle), d, o), o, 1, i);                   /* :263-264, :422-430 */
}

/* kernel0.cl:273-312 + :440-449 on the pre-laid-out record */
BDPT_DEV void test_triangle2(Best2 &b, B2 en, float4 q0, float4 q1, float4 q2, float4 q3, float4 q4, float e2z, int i,
                             V3 o, V3 d)
{
    const V3 n = mk3(splat(q0.x), splat(q0.y), splat(q0.z));
    const f2 val = dot3(d, n);                                                       /* :286 */
    B2 ok = andnot2(en, gt2(val, splat(0.0f)));                                      /* :288 */
    if (!any2(ok)) return;
    const V3 a = mk3(splat(q0.w), splat(q1.x), splat(q1.y));
    const f2 t = div2(dot3(sub3(a, o), n), val);                                     /* :291 */
    ok = andnot2(ok, lt2(t, splat(0.0f)));                                           /* :292 */
    if (!any2(ok)) return;
    const V3 p = mad3(t, d, o);                                                      /* :295 */
    const V3 e0 = mk3(splat(q1.z), splat(q1.w), splat(q2.x));
    const V3 c = mk3(splat(q2.y), splat(q2.z), splat(q2.w));
    const V3 e1 = mk3(splat(q3.x), splat(q3.y), splat(q3.z));
    const V3 bb = mk3(splat(q3.w), splat(q4.x), splat(q4.y));
    const V3 e2 = mk3(splat(q4.z), splat(q4.w), splat(e2z));
    const f2 dA = dot3(cross3(sub3(p, a), e0), n);                                   /* :298 */
    const f2 dB = dot3(cross3(sub3(p, c), e1), n);                                   /* :299 */
    const f2 dC = dot3(cross3(sub3(p, bb), e2), n);                                  /* :300 */
    const B2 out = mk2(dA.x < 0 || dB.x < 0 || dC.x < 0, dA.y < 0 || dB.y < 0 || dC.y < 0);   /* :304 */
    ok = andnot2(ok, out);
    if (!any2(ok)) return;
    best2_consider(b, ok, p, o, 2, i);                                               /* :441-448 */
}

/* kernel0.cl:365-478 for two rays that leave primitives (otype, oidx); `act` masks the components */
BDPT_DEV Hit2 closest_hit2(const bdpt_dev_scene &S, B2 act, V3 o, V3 d, i2 otype, i2 oidx)
{
    Best2 b;
    best2_init(b);
    i2 skip_s, skip_p, skip_t;
    skip_s.x = otype.x == 0 ? oidx.x : -1;  skip_s.y = otype.y == 0 ? oidx.y : -1;
    skip_p.x = otype.x == 1 ? oidx.x : -1;  skip_p.y = otype.y == 1 ? oidx.y : -1;
    skip_t.x = otype.x == 2 ? oidx.x : -1;  skip_t.y = otype.y == 2 ? oidx.y : -1;
    for (int i = 0; i < S.n_spheres; i++) {
        const float4 s = uload(S.sph, i);
        const B2 en = mk2(act.a && i != skip_s.x, act.b && i != skip_s.y);
        if (any2(en)) test_sphere2(b, en, s, i, o, d);
    }
    for (int i = 0; i < S.n_planes; i++) {
        const float4 q0 = uload(S.pln, (size_t)i * BDPT_PLN_F4 + 0), q1 = uload(S.pln, (size_t)i * BDPT_PLN_F4 + 1);
        const B2 en = mk2(act.a && i != skip_p.x, act.b && i != skip_p.y);
        if (any2(en)) test_plane2(b, en, q0, q1, i, o, d);
    }
    for (int i = 0; i < S.n_triangles; i++) {
        const size_t r = (size_t)i * BDPT_TRI_F4;
        const float4 q0 = uload(S.tri, r), q1 = uload(S.tri, r + 1), q2 = uload(S.tri, r + 2), q3 = uload(S.tri, r + 3),
                     q4 = uload(S.tri, r + 4);
        const float e2z = uload(S.tri, r + 5).x;
        const B2 en = mk2(act.a && i != skip_t.x, act.b && i != skip_t.y);
        if (any2(en)) test_triangle2(b, en, q0, q1, q2, q3, q4, e2z, i, o, d);
    }
    /* kernel0.cl:452-473 per component (gathers with per-lane addresses) */
    Hit2 h;
    Best b0, b1;
    b0.d2 = b.d2.x; b0.d2_lo = b.d2_lo.x; b0.type = b.type.x; b0.idx = b.idx.x; b0.pos = comp3(b.pos, 0);
    b1.d2 = b.d2.y; b1.d2_lo = b.d2_lo.y; b1.type = b.type.y; b1.idx = b.idx.y; b1.pos = comp3(b.pos, 1);
    const Hit h0 = finish_hit(S, b0, comp3(d, 0));
    const Hit h1 = finish_hit(S, b1, comp3(d, 1));
    h.type.x = h0.type; h.type.y = h1.type;
    h.idx.x = h0.idx;   h.idx.y = h1.idx;
    h.surf.x = h0.surf; h.surf.y = h1.surf;
    h.pos = pack3(h0.pos, h1.pos);
    h.n = pack3(h0.n, h1.n);
    return h;
}

}  // namespace bdpt

#endif

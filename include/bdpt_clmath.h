/*
 * bdpt_clmath.h -- the numeric contract: one deterministic instantiation of the
 * OpenCL C built-ins that the reference kernel calls.
 *
 * The reference (src/kernel/kernel0.cl) calls dot, cross, distance, length,
 * normalize, fast_normalize, native_sin, native_cos, native_divide, half_sqrt,
 * acos(float), acos(double), pow and sign.  OpenCL leaves the precision of the
 * native_/half_/fast_ forms to the implementation, allows 2.5-4 ulp on the rest
 * and lets the compiler contract a*b+c into fma, so no two OpenCL runtimes agree
 * bit for bit.  This header fixes ONE valid choice, written only with operations
 * that round identically on x86-64 and on gfx950: IEEE-754 binary32/binary64
 * add, sub, mul, div, sqrt(binary32), fused multiply-add, round-to-nearest-even
 * integer rounding and conversions.  Every translation unit that includes it
 * must be built with -ffp-contract=off so that the only fused operations are the
 * ones written here.
 *
 * It is included by the CPU oracle (plain C), by the host-side scene
 * pre-processing and by the HIP device code, so that a ray that is bit-identical
 * on both sides meets bit-identical primitives.
 *
 * Contraction rule used by every caller (stated once, applied everywhere):
 *   - dot(a,b)   = fma(a.z,b.z, fma(a.y,b.y, a.x*b.x))
 *   - cross(a,b) = per component fma(p,q, -(r*s)) with r*s rounded first
 *   - a vector expression "A + s*B" (s scalar, A possibly itself a rounded
 *     product) is fma(s, B, A) per component (clm_mad3)
 *   - every other expression rounds after each operation, as written.
 * Vectors are the xyz part of the reference's float4; its w is 0 for every
 * geometric quantity, so w contributes nothing to dot/length.
 */
#ifndef BDPT_CLMATH_H
#define BDPT_CLMATH_H

#include <stdint.h>

#if defined(__HIPCC__)
#define CLM_FN __host__ __device__ static inline __attribute__((always_inline))
#else
#define CLM_FN static inline
#endif

typedef struct clm_f3 { float x, y, z; } clm_f3;

#define CLM_PI_F64 3.14159265358979323846   /* M_PI   as the reference's fp64 literal */
#define CLM_1_PI_F64 0.318309886183790671538 /* M_1_PI */

CLM_FN float clm_fma(float a, float b, float c) { return __builtin_fmaf(a, b, c); }
CLM_FN float clm_sqrt(float x) { return __builtin_sqrtf(x); }           /* IEEE, also half_sqrt */
CLM_FN float clm_div(float a, float b) { return a / b; }               /* IEEE, also native_divide and '/' */

CLM_FN clm_f3 clm_make3(float x, float y, float z) { clm_f3 r; r.x = x; r.y = y; r.z = z; return r; }
CLM_FN clm_f3 clm_add3(clm_f3 a, clm_f3 b) { return clm_make3(a.x + b.x, a.y + b.y, a.z + b.z); }
CLM_FN clm_f3 clm_sub3(clm_f3 a, clm_f3 b) { return clm_make3(a.x - b.x, a.y - b.y, a.z - b.z); }
CLM_FN clm_f3 clm_neg3(clm_f3 a) { return clm_make3(-a.x, -a.y, -a.z); }
CLM_FN clm_f3 clm_scale3(float s, clm_f3 a) { return clm_make3(s * a.x, s * a.y, s * a.z); }
/* a + s*b, fused per component */
CLM_FN clm_f3 clm_mad3(float s, clm_f3 b, clm_f3 a)
{
    return clm_make3(clm_fma(s, b.x, a.x), clm_fma(s, b.y, a.y), clm_fma(s, b.z, a.z));
}

CLM_FN float clm_dot3(clm_f3 a, clm_f3 b)
{
    return clm_fma(a.z, b.z, clm_fma(a.y, b.y, a.x * b.x));
}

CLM_FN clm_f3 clm_cross3(clm_f3 a, clm_f3 b)
{
    return clm_make3(clm_fma(a.y, b.z, -(a.z * b.y)),
                     clm_fma(a.z, b.x, -(a.x * b.z)),
                     clm_fma(a.x, b.y, -(a.y * b.x)));
}

CLM_FN float clm_length3(clm_f3 a) { return clm_sqrt(clm_dot3(a, a)); }
CLM_FN float clm_distance3(clm_f3 a, clm_f3 b) { return clm_length3(clm_sub3(a, b)); }

/* normalize and fast_normalize: one reciprocal of the length, three multiplies. */
CLM_FN clm_f3 clm_normalize3(clm_f3 a)
{
    float inv = clm_div(1.0f, clm_length3(a));
    return clm_scale3(inv, a);
}

/* (-2 * n * dot(n, v)) + v, evaluated left to right, each operation rounded (kernel0.cl:495). */
CLM_FN clm_f3 clm_reflect3(clm_f3 n, clm_f3 v)
{
    float dn = clm_dot3(n, v);
    clm_f3 r;
    r.x = ((-2.0f * n.x) * dn) + v.x;
    r.y = ((-2.0f * n.y) * dn) + v.y;
    r.z = ((-2.0f * n.z) * dn) + v.z;
    return r;
}

/* sign(): 1, -1, the zero itself, or 0 for NaN. */
CLM_FN float clm_sign(float x)
{
    if (x > 0.0f) return 1.0f;
    if (x < 0.0f) return -1.0f;
    if (x == 0.0f) return x;
    return 0.0f;
}

/* pow(base, n) for the small non-negative integer exponents the kernel uses. */
CLM_FN float clm_powi(float base, int n)
{
    float r = 1.0f;
    for (int i = 0; i < n; i++) r = r * base;
    return r;
}

/*
 * native_sin / native_cos, computed together.  Cody-Waite reduction by pi/2 in
 * three fused steps, then the classic 3-term minimax polynomials on
 * [-pi/4, pi/4].  About 1-2 ulp for |x| up to a few thousand; the kernel only
 * ever passes [0, 2*pi].
 */
CLM_FN void clm_sincos(float x, float *s_out, float *c_out)
{
    const float TWO_OVER_PI = 0x1.45f306p-1f;
    const float PIO2_HI = 0x1.921fb6p+0f;
    const float PIO2_MID = -0x1.777a5cp-25f;
    const float PIO2_LO = -0x1.ee59dap-50f;
    float kf = __builtin_rintf(x * TWO_OVER_PI);
    float r = clm_fma(-kf, PIO2_HI, x);
    r = clm_fma(-kf, PIO2_MID, r);
    r = clm_fma(-kf, PIO2_LO, r);
    float z = r * r;
    float ps = clm_fma(-1.9515295891e-4f, z, 8.3321608736e-3f);
    ps = clm_fma(ps, z, -1.6666654611e-1f);
    float sn = clm_fma(ps * z, r, r);
    float pc = clm_fma(2.443315711809948e-5f, z, -1.388731625493765e-3f);
    pc = clm_fma(pc, z, 4.166664568298827e-2f);
    float cs = clm_fma(pc, z * z, clm_fma(-0.5f, z, 1.0f));
    int q = ((int)kf) & 3;
    float s = (q & 1) ? cs : sn;
    float c = (q & 1) ? sn : cs;
    if (q == 1 || q == 2) c = -c;
    if (q >= 2) s = -s;
    *s_out = s;
    *c_out = c;
}

/* asin(s) for |s| <= 0.5 given z = s*s: s + s*z*P(z). */
CLM_FN float clm_asin_core(float s, float z)
{
    float p = clm_fma(4.2163199048e-2f, z, 2.4181311049e-2f);
    p = clm_fma(p, z, 4.5470025998e-2f);
    p = clm_fma(p, z, 7.4953002686e-2f);
    p = clm_fma(p, z, 1.6666752422e-1f);
    return clm_fma(s * z, p, s);
}

/* acos(float) on [-1, 1]: beyond +-0.5 through asin(sqrt((1 - |x|) / 2)).  One evaluation of the root and of the
 * polynomial whatever the range (the arguments are selected, not the code path): on a GPU the lanes of a
 * wavefront fall into all three ranges at once. */
CLM_FN float clm_acosf(float x)
{
    const float PI_F = 0x1.921fb6p+1f;
    const float PIO2_F = 0x1.921fb6p+0f;
    const float ax = x < 0.0f ? -x : x;
    const int outer = ax > 0.5f;
    const float z = outer ? 0.5f * (1.0f - ax) : x * x;
    const float s = outer ? clm_sqrt(z) : x;
    const float a = clm_asin_core(s, z);
    if (!outer) return PIO2_F - a;
    return x > 0.0f ? 2.0f * a : PI_F - 2.0f * a;
}

/* sqrt(binary64) from a binary32 seed and two Newton steps (only + * /). */
CLM_FN double clm_sqrt_f64(double z)
{
    float zf = (float)z;
    if (!(zf > 0.0f)) return 0.0;
    double s = (double)clm_sqrt(zf);
    s = 0.5 * (s + z / s);
    s = 0.5 * (s + z / s);
    return s;
}

/* acos(double) on [-1, 1]: rational approximation of asin on [0, 0.5]. */
CLM_FN double clm_acos_f64(double x)
{
    const double PIO2_HI = 1.57079632679489655800e+00;
    const double PIO2_LO = 6.12323399573676603587e-17;
    const double PI_D = 3.14159265358979311600e+00;
    const double pS0 = 1.66666666666666657415e-01, pS1 = -3.25565818622400915405e-01,
                 pS2 = 2.01212532134862925881e-01, pS3 = -4.00555345006794114027e-02,
                 pS4 = 7.91534994289814532176e-04, pS5 = 3.47933107596021167570e-05;
    const double qS1 = -2.40339491173441421878e+00, qS2 = 2.02094576023350569471e+00,
                 qS3 = -6.88283971605453293030e-01, qS4 = 7.70381505559019352791e-02;
    double ax = x < 0.0 ? -x : x;
    double z = (ax < 0.5) ? x * x : (1.0 - ax) * 0.5;
    double p = z * (pS0 + z * (pS1 + z * (pS2 + z * (pS3 + z * (pS4 + z * pS5)))));
    double q = 1.0 + z * (qS1 + z * (qS2 + z * (qS3 + z * qS4)));
    double r = p / q;
    if (ax < 0.5) return PIO2_HI - (x - (PIO2_LO - x * r));
    double s = clm_sqrt_f64(z);
    double w = s + s * r;
    if (x < 0.0) return PI_D - 2.0 * (w - PIO2_LO);
    return 2.0 * w;
}

#endif /* BDPT_CLMATH_H */

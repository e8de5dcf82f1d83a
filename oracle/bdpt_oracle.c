/*
 * bdpt_oracle.c -- CPU restatement of the reference's BDPT sample kernel.
 *
 * TEST INFRASTRUCTURE ONLY (see bdpt_oracle.h).  PARITY UNPINNED: the reference
 * holds no tests or golden data and cannot be executed here, so this file is
 * checked only against itself (committed fixtures) and against analytic
 * known-answer cases in tests/.
 *
 * Every function follows one function of /root/reference/src/kernel/kernel0.cl
 * (cited as kernel0.cl:LINE) statement by statement, in the same order, with
 * the same comparison operators.  The OpenCL built-ins are the instantiations
 * of include/bdpt_clmath.h and floating-point contraction follows the rule
 * stated there.  Build with -ffp-contract=off.
 *
 * Resolutions of behaviours the reference leaves undefined:
 *   - the primitive skipped by the first light-subpath rays (uninitialised in
 *     sampleSphereSurface, kernel0.cl:547-569, used at :705, :793) is a
 *     parameter: (0, light_index) or none;
 *   - primType/index are ints, -1 on a miss (the reference keeps them as
 *     floats and casts -1.0f to uint, kernel0.cl:384-385, :454-455); on a miss
 *     no surface or normal is fetched;
 *   - the accumulator is fp32 RGBA, unclamped.
 */
#include "bdpt_oracle.h"
#include "../include/bdpt_clmath.h"

#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

/* ---- scene view ----------------------------------------------------------- */

typedef struct {
    const bdpt_sphere *spheres;     int n_spheres;
    const bdpt_plane *planes;       int n_planes;
    const bdpt_triangle *triangles; int n_triangles;
    const bdpt_surface *surfaces;
} Scene;

/* Intersection, kernel0.cl:40-46 (vDir is never read downstream) */
typedef struct {
    clm_f3 pos;
    int hit;
} Inter;

/* FullIntersection, kernel0.cl:48-58 */
typedef struct {
    Inter k;
    int type;
    int index;
    clm_f3 n;          /* vDir: shading normal */
    float diffuse[4];
    float emissive[4];
} Full;

typedef struct {
    uint32_t w;        /* state.w, the only live word of the uint4 state */
    uint32_t draws;
    uint32_t rays;
    uint32_t hash;
} Lane;

static clm_f3 v3(const float *p) { return clm_make3(p[0], p[1], p[2]); }

/* ---- generator, kernel0.cl:106-112 ---------------------------------------- */

static uint32_t rng_next(Lane *L)
{
    L->w = 48271u * L->w;
    L->draws++;
    return L->w;
}

static float get_rng_float(Lane *L)
{
    /* (float)UINT_MAX rounds to 2^32 */
    return clm_div((float)rng_next(L), 4294967296.0f);
}

/* ---- random_sample_cone, kernel0.cl:155-185 ------------------------------- */

static clm_f3 random_sample_cone(Lane *L, clm_f3 vDir, float halfAngle)
{
    float p0 = get_rng_float(L);
    float p1 = get_rng_float(L);
    float p2 = get_rng_float(L);

    /* :163  p2*2.0 - 1.0f is evaluated in fp64 and narrowed by the float4 constructor */
    clm_f3 vecRand = clm_make3((float)((double)p2 * 2.0 - (double)1.0f), 0.0f, 1.0f);
    clm_f3 vPerp = clm_normalize3(clm_cross3(vDir, vecRand));           /* :168 */
    clm_f3 vPerp2 = clm_cross3(vDir, vPerp);                            /* :170 */

    p1 = p1 * 6.283185307f;                                             /* :177 */
    /* :178  acos(float)*2.0f in fp32, then * M_1_PI, 1.0f - .., * halfAngle in fp64 */
    p0 = (float)(((double)1.0f - ((double)(clm_acosf(p0) * 2.0f) * CLM_1_PI_F64)) * (double)halfAngle);

    float s0, c0, s1, c1;
    clm_sincos(p0, &s0, &c0);
    clm_sincos(p1, &s1, &c1);
    /* :182  cos(p0)*vDir + sin(p0)*(cos(p1)*vPerp + sin(p1)*vPerp2) */
    clm_f3 inner = clm_mad3(s1, vPerp2, clm_scale3(c1, vPerp));
    return clm_mad3(s0, inner, clm_scale3(c0, vDir));
}

/* ---- random_sample_hemisphere, kernel0.cl:117-153 --------------------------- */

static clm_f3 random_sample_hemisphere(Lane *L, clm_f3 vDir)
{
    float p0 = get_rng_float(L);
    float p1 = get_rng_float(L);
    float p2 = get_rng_float(L);
    clm_f3 vecRand = clm_make3((float)((double)p2 * 2.0 - (double)1.0f), 0.0f, 1.0f);   /* :127 */
    clm_f3 vPerp = clm_normalize3(clm_cross3(vDir, vecRand));           /* :132 */
    clm_f3 vPerp2 = clm_cross3(vDir, vPerp);                            /* :134 */
    p1 = p1 * 6.283185307f;                                             /* :141 */
    p0 = p0 * 1.570796326f;                                             /* :142 */
    float s0, c0, s1, c1;
    clm_sincos(p0, &s0, &c0);
    clm_sincos(p1, &s1, &c1);
    clm_f3 inner = clm_mad3(s1, vPerp2, clm_scale3(c1, vPerp));         /* :150 */
    return clm_mad3(s0, inner, clm_scale3(c0, vDir));
}

/* ---- brdf_reflective, kernel0.cl:492-508 ------------------------------------ */

static clm_f3 brdf_reflective(Lane *L, clm_f3 normal, clm_f3 vDir)
{
    float p4 = get_rng_float(L);                                        /* :493 */
    clm_f3 reflected = clm_reflect3(normal, vDir);                      /* :495 */
    if (p4 > 0.6f) return random_sample_cone(L, reflected, 3.1415f * 0.02f);   /* :500-501 */
    return random_sample_cone(L, normal, 3.1415f * 0.50f);              /* :503 */
}

/* ---- brdf_lambertian_imp, kernel0.cl:512-543 -------------------------------- */
/* Only the direction: the density it also returns in .w (:527-540) is zeroed unread by the live kernel (:772-778). */

static clm_f3 brdf_lambertian_imp(Lane *L, clm_f3 normal, clm_f3 vDirLight, float lightWeight, float halfAngle)
{
    float p4 = get_rng_float(L);                                        /* :513 */
    clm_f3 dir;
    float angle;
    if (p4 > lightWeight) {                                             /* :517 */
        dir = normal;                                                   /* :518 */
        angle = (float)((double)0.50f * CLM_PI_F64);                    /* :519, fp64 product narrowed by the assignment */
    } else {
        dir = vDirLight;                                                /* :521 */
        angle = halfAngle;                                              /* :522 */
    }
    return random_sample_cone(L, dir, angle);                           /* :525 */
}

/* ---- findRandomPoint, kernel0.cl:190-202 ----------------------------------- */

static void findRandomPoint(Lane *L, float blur, int W, int H, float pixx, float pixy,
                            float *outx, float *outy)
{
    float scaleX = get_rng_float(L);
    float scaleY = get_rng_float(L);
    float deltaX = clm_div(blur, (float)W);
    float deltaY = clm_div(blur, (float)H);
    *outx = pixx + scaleX * deltaX;
    *outy = pixy + scaleY * deltaY;
}

/* ---- shoot_ray, kernel0.cl:206-217 ----------------------------------------- */

static clm_f3 shoot_ray(float sx, float sy, const bdpt_camera *cam)
{
    clm_f3 look = v3(cam->look_at), up = v3(cam->up);
    /* z == 1.0f, so lookAt*z and up*z are exact */
    clm_f3 vCompY = clm_mad3(sy, up, look);                              /* :211 */
    clm_f3 vCompX = clm_mad3(sx, clm_cross3(up, look), look);            /* :212 */
    return clm_normalize3(clm_add3(vCompX, vCompY));                     /* :215 */
}

/* ---- rayPlaneIntersect, kernel0.cl:232-271 --------------------------------- */

static Inter rayPlaneIntersect(clm_f3 vPos, clm_f3 vDir, const bdpt_plane *pl)
{
    Inter x;
    x.pos = clm_make3(0, 0, 0);
    x.hit = 0;
    clm_f3 normal = v3(pl->normal), pt = v3(pl->point);

    float angle = clm_dot3(vDir, normal);                                /* :249 */
    if (angle >= 0) return x;                                            /* :252 */

    float temp = clm_dot3(clm_sub3(pt, vPos), normal);                   /* :256 */
    float desc = clm_dot3(clm_scale3(temp, normal), vDir);               /* :257 */
    if (desc > 0) x.hit = 1;                                             /* :258 */

    float k = clm_div(temp, angle);                                      /* :263 */
    x.pos = clm_mad3(k, vDir, vPos);                                     /* :264 */
    return x;
}

/* ---- rayTriangleIntersect, kernel0.cl:273-312 ------------------------------ */

static Inter rayTriangleIntersect(clm_f3 vPos, clm_f3 vDir, const bdpt_triangle *t)
{
    Inter ip;
    ip.hit = 0;
    ip.pos = clm_make3(0, 0, 0);
    clm_f3 a = v3(t->p0), b = v3(t->p1), c = v3(t->p2);

    clm_f3 normal = clm_normalize3(clm_cross3(clm_sub3(b, a), clm_sub3(c, a)));   /* :285 */
    float val = clm_dot3(vDir, normal);                                            /* :286 */
    if (val > 0) return ip;                                                        /* :288 */

    float lineDist = clm_div(clm_dot3(clm_sub3(a, vPos), normal), val);            /* :291 */
    if (lineDist < 0) return ip;                                                   /* :292 */

    clm_f3 point = clm_mad3(lineDist, vDir, vPos);                                 /* :295 */
    float dotA = clm_dot3(clm_cross3(clm_sub3(point, a), clm_sub3(c, a)), normal); /* :298 */
    float dotB = clm_dot3(clm_cross3(clm_sub3(point, c), clm_sub3(b, c)), normal); /* :299 */
    float dotC = clm_dot3(clm_cross3(clm_sub3(point, b), clm_sub3(a, b)), normal); /* :300 */
    if (dotA < 0 || dotB < 0 || dotC < 0) return ip;                               /* :304 */

    ip.hit = 1;
    ip.pos = point;
    return ip;
}

/* ---- raySphereIntersect, kernel0.cl:314-361 -------------------------------- */

static Inter raySphereIntersect(clm_f3 vPos, clm_f3 vDir, const bdpt_sphere *sph)
{
    Inter ip;
    ip.hit = 0;
    ip.pos = clm_make3(0, 0, 0);
    clm_f3 oc = clm_sub3(vPos, v3(sph->center));

    float dot1 = clm_dot3(vDir, oc);                                     /* :331 */
    if (dot1 > 0) return ip;                                             /* :333 */

    float dist = clm_dot3(oc, oc);                                       /* :336 */
    float curDisc = dot1 * dot1 - dist + (sph->radius * sph->radius);    /* :337 */
    if (curDisc < 0) return ip;                                          /* :341 */

    float discSqrt = clm_sqrt(curDisc);                                  /* :344 half_sqrt */
    float distA = -dot1 - discSqrt;                                      /* :347 */
    if (distA < 0) return ip;                                            /* :350 */

    ip.hit = 1;
    ip.pos = clm_mad3(distA, vDir, vPos);                                /* :354 */
    return ip;
}

/* ---- ray_intersect, kernel0.cl:365-478 ------------------------------------- */

static void copy4(float *d, const float *s) { d[0] = s[0]; d[1] = s[1]; d[2] = s[2]; d[3] = s[3]; }

static Full ray_intersect(Lane *L, const Scene *sc, clm_f3 vPos, clm_f3 vDir,
                          int originPrim, int originIndex)
{
    float minDist = 10000.0f;                                            /* :374 */
    Inter currInter;
    currInter.hit = 0;
    currInter.pos = clm_make3(0, 0, 0);
    int primType = -1, index = -1;                                       /* :384-385 */
    uint32_t surf = 0;

    for (int i = 0; i < sc->n_spheres; i++) {                            /* :388 */
        if (originPrim == 0 && originIndex == i) continue;               /* :390 */
        Inter inter = raySphereIntersect(vPos, vDir, &sc->spheres[i]);
        if (inter.hit != 0) {
            float dist = clm_distance3(inter.pos, vPos);                 /* :398 */
            if (dist < minDist) {                                        /* :400 */
                minDist = dist;
                currInter = inter;
                primType = 0;
                index = i;
                surf = sc->spheres[i].surface;
            }
        }
    }
    for (int i = 0; i < sc->n_planes; i++) {                             /* :413 */
        if (originPrim == 1 && originIndex == i) continue;
        Inter inter = rayPlaneIntersect(vPos, vDir, &sc->planes[i]);
        if (inter.hit != 0) {
            float dist = clm_distance3(inter.pos, vPos);                 /* :422 */
            if (dist < minDist) {
                minDist = dist;
                primType = 1;
                currInter = inter;
                index = i;
                surf = sc->planes[i].surface;
            }
        }
    }
    for (int i = 0; i < sc->n_triangles; i++) {                          /* :434 */
        if (originPrim == 2 && originIndex == i) continue;
        Inter inter = rayTriangleIntersect(vPos, vDir, &sc->triangles[i]);
        if (inter.hit != 0) {
            float dist = clm_distance3(inter.pos, vPos);                 /* :441 */
            if (dist < minDist) {
                minDist = dist;
                primType = 2;
                currInter = inter;
                index = i;
                surf = sc->triangles[i].surface;
            }
        }
    }

    Full full;
    memset(&full, 0, sizeof full);
    full.type = primType;
    full.index = index;
    full.k = currInter;
    full.n = clm_make3(0, 0, 0);
    if (primType >= 0) {
        copy4(full.diffuse, sc->surfaces[surf].color);                   /* :457 */
        copy4(full.emissive, sc->surfaces[surf].emissive);               /* :458 */
    }

    if (primType == 0) {                                                 /* :461 */
        full.n = clm_normalize3(clm_sub3(v3(sc->spheres[index].center), full.k.pos));
    } else if (primType == 1) {                                          /* :463 */
        full.n = v3(sc->planes[index].normal);
    } else if (primType == 2) {                                          /* :465 */
        const bdpt_triangle *t = &sc->triangles[index];
        clm_f3 a = v3(t->p0), b = v3(t->p1), c = v3(t->p2);
        full.n = clm_normalize3(clm_cross3(clm_sub3(a, b), clm_sub3(b, c)));      /* :467 */
    }
    if (clm_dot3(vDir, full.n) > 0) full.n = clm_neg3(full.n);           /* :471 */

    /* bookkeeping for the parity outputs (not in the reference) */
    L->rays++;
    uint32_t id = primType < 0 ? 0xFFFFFFFFu : (((uint32_t)primType << 28) | (uint32_t)index);
    L->hash = (L->hash ^ id) * 16777619u + 0x9E3779B9u;
    return full;
}

/* ---- sampleSphereSurface, kernel0.cl:547-569 ------------------------------- */

static void sample_sphere_point(Lane *L, const bdpt_sphere *sph, clm_f3 *pos, clm_f3 *nrm)
{
    float p0 = get_rng_float(L);
    float p1 = get_rng_float(L);

    p0 = (float)((double)p0 * CLM_PI_F64);                               /* :553 */
    /* :554  2.0f*M_1_PI*p0 and the subtraction in fp64, acos(double), narrowed */
    p0 = (float)clm_acos_f64(1.0 - (((double)2.0f * CLM_1_PI_F64) * (double)p0));
    p1 = (float)((double)(p1 * 2.0f) * CLM_PI_F64);                      /* :556 */

    float s0, c0, s1, c1;
    clm_sincos(p0, &s0, &c0);
    clm_sincos(p1, &s1, &c1);
    clm_f3 vDir = clm_make3(s0 * s1, c0, s0 * c1);                       /* :558 */
    *pos = clm_mad3(sph->radius, vDir, v3(sph->center));                 /* :559 */
    *nrm = vDir;                                                         /* :561 */
}

/* ---- bi_directional_path_trace, kernel0.cl:571-864: ONE sample ------------- */

typedef struct {
    int early;          /* 1: the reference returned at :649 or :655 */
    float col[4];       /* value of this sample (or the raw value written on an early return) */
    int eye_type, eye_index;
} Sample;

static Sample trace_sample(Lane *L, const orc_job *J, const Scene *sc, int x, int y)
{
    const bdpt_params *P = J->params;
    const bdpt_camera *cam = J->camera;
    const int W = J->width, H = J->height;
    Sample S;
    memset(&S, 0, sizeof S);

    rng_next(L);                                                          /* :592 */

    float xf = clm_div((float)x, (float)W * 1.0f);                        /* :600 */
    float yf = clm_div((float)y, (float)H * 1.0f);                        /* :601 */
    /* :603-604 fp64: (xf*2.0 - 1.0f) * (WIDTH*1.0f / HEIGHT) */
    float aspect = clm_div((float)W * 1.0f, (float)H);
    float xt = (float)(((double)xf * 2.0 - (double)1.0f) * (double)aspect);
    float yt = (float)((double)yf * 2.0 - (double)1.0f);

    float sx, sy;
    findRandomPoint(L, P->blur_factor, W, H, xt, yt, &sx, &sy);           /* :613 */
    clm_f3 ray = shoot_ray(sx, sy, cam);                                  /* :615 */

    clm_f3 camPos = v3(cam->pos);
    clm_f3 eyePos = clm_mad3(0.00001f, ray, camPos);                      /* :632 */
    Full firstPatch = ray_intersect(L, sc, eyePos, ray, 4, 0);            /* :634 */
    clm_f3 vTestPos = firstPatch.k.pos;                                   /* :637 */
    S.eye_type = firstPatch.type;
    S.eye_index = firstPatch.index;

    if (firstPatch.k.hit) {                                               /* :642 */
        if (firstPatch.emissive[3] != 0.0f) {                             /* :645 */
            copy4(S.col, firstPatch.diffuse);                             /* :648 */
            S.early = 1;
            return S;
        }
    } else {
        S.col[0] = S.col[1] = S.col[2] = S.col[3] = 0.0f;                 /* :654 */
        S.early = 1;
        return S;
    }

    const int lIndex = P->light_index;                                    /* :663 */
    const bdpt_sphere *lsph = &sc->spheres[lIndex];
    Full patch;                                                           /* :669, :675 */
    memset(&patch, 0, sizeof patch);
    sample_sphere_point(L, lsph, &patch.k.pos, &patch.n);
    patch.k.hit = 1;
    const float *lightEmissive = sc->surfaces[lsph->surface].emissive;    /* :563 */
    if (P->light_origin == BDPT_LIGHT_ORIGIN_SELF) { patch.type = 0; patch.index = lIndex; }
    else { patch.type = -1; patch.index = -1; }

    const int maxL = P->max_light_bounces;                                /* :683 */
    const float pCon = P->p_continue;                                     /* :687 */
    float fullContrib[4] = {0, 0, 0, 0};                                  /* :689 */
    float accContrib[4] = {1, 1, 1, 1};                                   /* :692 */
    float preFactor = clm_div(1.0f - pCon, 1.0f - clm_powi(pCon, maxL));  /* :694 */
    int samplesTaken = 0;
    clm_f3 incomingDir = clm_make3(0.0f, 0.0f, 0.0f);                     /* :659 */

    for (int lti = 0; lti < maxL; lti++) {                                /* :696 */
        clm_f3 testDir = clm_normalize3(clm_sub3(vTestPos, patch.k.pos)); /* :699 */
        Full testPatch = ray_intersect(L, sc, patch.k.pos, testDir, patch.type, patch.index);  /* :705 */
        float contrib[4] = {0, 0, 0, 0};

        if (firstPatch.type == testPatch.type && firstPatch.index == testPatch.index) {       /* :712 */
            if (clm_length3(clm_sub3(vTestPos, testPatch.k.pos)) < 0.0001f &&
                clm_dot3(firstPatch.n, testPatch.n) > 0) {                /* :714 */
                clm_f3 ptVec = clm_sub3(testPatch.k.pos, patch.k.pos);    /* :716 */
                float r2 = ptVec.x * ptVec.x + ptVec.y * ptVec.y + ptVec.z * ptVec.z;
                float samplingFactor;
                if (P->connection_falloff == BDPT_FALLOFF_INV_SQ_PLUS_C)
                    samplingFactor = clm_div(1.0f, P->falloff_c + r2);
                else
                    samplingFactor = clm_div(1.0f, r2);                   /* :719 */
                float impFactor = preFactor;                              /* :730 */
                float directionalFactor = clm_dot3(firstPatch.n, testDir) * clm_dot3(patch.n, testDir);  /* :732 */
                directionalFactor *= clm_sign(directionalFactor);         /* :733 */
                float invImp = clm_div(1.0f, impFactor);
                for (int k = 0; k < 4; k++)                               /* :741 */
                    contrib[k] = accContrib[k] * samplingFactor * directionalFactor * invImp;
            }
        }
        preFactor *= pCon;                                                /* :745 */

        if (!P->indirect_only || lti != 0)                                /* :752-755 */
            for (int k = 0; k < 4; k++) fullContrib[k] += contrib[k];
        samplesTaken++;                                                   /* :757 */

        clm_f3 launchRef = patch.n;                                       /* :760 */
        clm_f3 launchPos = clm_mad3(0.00001f, patch.n, patch.k.pos);      /* :761 */
        clm_f3 launchDir;
        if (P->sampler == BDPT_SAMPLER_HEMISPHERE) launchDir = random_sample_hemisphere(L, launchRef);
        else if (P->sampler == BDPT_SAMPLER_REFLECTIVE) launchDir = brdf_reflective(L, launchRef, incomingDir);
        else if (P->sampler == BDPT_SAMPLER_LAMBERT_IMP)                  /* the call at :945, aimed at the light's centre */
            launchDir = brdf_lambertian_imp(L, launchRef, clm_normalize3(clm_sub3(v3(lsph->center), launchPos)),
                                            P->imp_light_weight, P->imp_half_angle);
        else launchDir = random_sample_cone(L, launchRef, 3.1415f * 0.50f);   /* :767, :486 */

        float fate = get_rng_float(L);                                    /* :787 */
        if (fate > pCon) break;                                           /* :788 */

        Full target = ray_intersect(L, sc, launchPos, launchDir, patch.type, patch.index);     /* :793 */
        if (target.k.hit) {                                               /* :796 */
            if (target.emissive[3] != 0.0f) break;                        /* :797-803 */
        } else {
            break;                                                        /* :812 */
        }
        /* :822, :824 force samplingFactor = directionalFactor = 1.0f */
        for (int k = 0; k < 4; k++)                                       /* :827 */
            accContrib[k] *= target.diffuse[k] * 1.0f * 1.0f * 1.0f;
        incomingDir = launchDir;                                          /* :831 */
        patch = target;                                                   /* :832 */
    }

    float invTaken = clm_div(1.0f, (float)samplesTaken);                  /* :846 */
    for (int k = 0; k < 4; k++) {
        float c = lightEmissive[k] * fullContrib[k] * 1.0f * invTaken;
        if (!P->indirect_only) c = c * firstPatch.diffuse[k];             /* :844 vs :846 */
        S.col[k] = c;
    }
    return S;
}

/* ---- path_trace, kernel0.cl:872-1001 (commented out in the reference): ONE sample of its loop :908-996 ---- */

static Sample path_trace_sample(Lane *L, const orc_job *J, const Scene *sc, float xt, float yt)
{
    const bdpt_params *P = J->params;
    const bdpt_camera *cam = J->camera;
    Sample S;
    memset(&S, 0, sizeof S);
    S.eye_type = S.eye_index = -1;

    float sx, sy;
    findRandomPoint(L, P->blur_factor, J->width, J->height, xt, yt, &sx, &sy);   /* :910 */
    clm_f3 ray = shoot_ray(sx, sy, cam);                                  /* :912 */

    Full patch;                                                           /* eyePatch, :919-923, :928 */
    memset(&patch, 0, sizeof patch);
    patch.n = ray;
    patch.k.pos = v3(cam->pos);
    patch.type = -1;
    patch.index = -1;

    float col[4] = {1.0f, 1.0f, 1.0f, 1.0f};                              /* :925 */
    clm_f3 incomingDir = clm_make3(0.0f, 0.0f, 0.0f);                     /* :930 */
    for (int i = 0; i < 5; i++) {                                         /* :932 */
        clm_f3 launchRef = patch.n;                                       /* :935 */
        clm_f3 launchPos = clm_mad3(0.00001f, patch.n, patch.k.pos);      /* :936 */
        clm_f3 launchDir = brdf_reflective(L, patch.n, incomingDir);      /* :943 */
        float wt = (float)((double)0.5f * CLM_1_PI_F64);                  /* :947, launchDir.w of :505 */
        if (i == 0) {                                                     /* :951-954 */
            wt = 1.0f;
            launchDir = launchRef;
        }
        if (i == 4) {                                                     /* :956-959 */
            for (int k = 0; k < 4; k++) col[k] *= 0.3f;
            break;
        }
        Full patch2 = ray_intersect(L, sc, launchPos, launchDir, patch.type, patch.index);   /* :964 */
        if (i == 0) { S.eye_type = patch2.type; S.eye_index = patch2.index; }
        incomingDir = launchDir;                                          /* :966 */
        if (patch2.k.hit) {                                               /* :968 */
            if (patch2.emissive[3] != 0.0f) {                             /* :969-974 */
                for (int k = 0; k < 4; k++)
                    col[k] *= clm_div(patch2.emissive[k], wt) * ((float)CLM_1_PI_F64 * 0.5f);
                break;
            }
            for (int k = 0; k < 4; k++)                                   /* :976 */
                col[k] *= clm_div(patch2.diffuse[k], wt) * (float)CLM_1_PI_F64 * 0.5f;
        } else {                                                          /* :978-982 */
            col[0] *= 0.3f; col[1] *= 0.3f; col[2] *= 0.4f; col[3] *= 0.3f;
            break;
        }
        patch = patch2;                                                   /* :984 */
    }
    copy4(S.col, col);                                                    /* :989 */
    return S;
}

/* One launch of path_trace with sampleRate = spp for one pixel (:884-1000). */
static void path_trace_pixel(orc_job *J, const Scene *sc, int x, int y, uint64_t cnt[3])
{
    const int W = J->width, H = J->height;
    const size_t pix = (size_t)y * W + x;
    const size_t sidx = ((size_t)x * H + y) * 4 + 3;                      /* :890 */
    Lane L;
    L.draws = J->aov_rng_draws ? J->aov_rng_draws[pix] : 0;
    L.rays = J->aov_rays ? J->aov_rays[pix] : 0;
    L.hash = J->aov_hit_hash ? J->aov_hit_hash[pix] : 0;
    L.w = J->seeds[sidx];
    float *acc = &J->accum[pix * 4];                                      /* prevValue, :887 */
    rng_next(&L);                                                         /* :892 */
    float xf = clm_div((float)x, (float)W * 1.0f);                        /* :899 */
    float yf = clm_div((float)y, (float)H * 1.0f);                        /* :900 */
    float aspect = clm_div((float)W * 1.0f, (float)H);
    float xt = (float)(((double)xf * 2.0 - (double)1.0f) * (double)aspect);   /* :902 */
    float yt = (float)((double)yf * 2.0 - (double)1.0f);                  /* :903 */
    Sample S;
    memset(&S, 0, sizeof S);
    S.eye_type = S.eye_index = -1;
    for (int k = 0; k < J->spp; k++) {                                    /* :908 */
        const int n = J->num_samples + k;
        uint32_t rays0 = L.rays;
        S = path_trace_sample(&L, J, sc, xt, yt);
        cnt[0] += L.rays - rays0;
        cnt[1]++;
        if (n == 0) {                                                     /* :991-992 */
            for (int c = 0; c < 4; c++) acc[c] = S.col[c];
        } else {                                                          /* :990 */
            float s1 = clm_div((float)n, (float)(n + 1));
            float s2 = clm_div(1.0f, (float)(n + 1));
            for (int c = 0; c < 4; c++) acc[c] = clm_fma(S.col[c], s2, acc[c] * s1);
        }
    }
    J->seeds[sidx] = L.w;                                                 /* :998 */
    if (J->aov_eye_hit) { J->aov_eye_hit[pix * 2] = S.eye_type; J->aov_eye_hit[pix * 2 + 1] = S.eye_index; }
    if (J->aov_radiance) for (int k = 0; k < 4; k++) J->aov_radiance[pix * 4 + k] = S.col[k];
    if (J->aov_hit_hash) J->aov_hit_hash[pix] = L.hash;
    if (J->aov_rng_draws) J->aov_rng_draws[pix] = L.draws;
    if (J->aov_rays) J->aov_rays[pix] = L.rays;
}

/* LCG multiplier for one sample's stride in STRIDED mode: 48271^32 mod 2^32 */
static uint32_t pow_u32(uint32_t base, uint32_t e)
{
    uint32_t r = 1u;
    while (e) {
        if (e & 1u) r *= base;
        base *= base;
        e >>= 1;
    }
    return r;
}

/* Class of a STRIDED sample: the number of light-subpath vertices Russian roulette allows it
 * (the draws at kernel0.cl:787 are draws 9, 13, 17, ... of the sample -- 10, 15, 20, ... with the
 * reflective sampler's extra draw per vertex -- and depend on nothing else). */
static int roulette_class(const bdpt_params *P, uint32_t start_state)
{
    const int per_vertex = (P->sampler == BDPT_SAMPLER_REFLECTIVE || P->sampler == BDPT_SAMPLER_LAMBERT_IMP) ? 5 : 4;   /* draws from one roulette draw to the next */
    uint32_t w = start_state;
    for (int j = 0; j < 5 + per_vertex; j++) w *= 48271u;
    for (int k = 0; k + 1 < P->max_light_bounces; k++) {
        float fate = clm_div((float)w, 4294967296.0f);
        if (fate > P->p_continue) return k + 1;
        for (int j = 0; j < per_vertex; j++) w *= 48271u;
    }
    return P->max_light_bounces;
}

static void render_pixel(orc_job *J, const Scene *sc, int x, int y, uint64_t cnt[3])
{
    const bdpt_params *P = J->params;
    const int W = J->width, H = J->height;
    const size_t pix = (size_t)y * W + x;
    const size_t sidx = ((size_t)x * H + y) * 4 + 3;                      /* :590 */
    Lane L;
    L.draws = J->aov_rng_draws ? J->aov_rng_draws[pix] : 0;
    L.rays = J->aov_rays ? J->aov_rays[pix] : 0;
    L.hash = J->aov_hit_hash ? J->aov_hit_hash[pix] : 0;
    float *acc = &J->accum[pix * 4];
    Sample S;
    memset(&S, 0, sizeof S);

    /* A STRIDED + SUM call adds its samples class by class (longest roulette class first, sample
     * index ascending within a class) unless sample_order says otherwise; every other combination
     * takes them in index order. */
    int reorder = P->rng_mode == BDPT_RNG_STRIDED && P->accum_mode == BDPT_ACCUM_SUM &&
                  P->sample_order != BDPT_ORDER_INDEX;
    const uint32_t stride = pow_u32(48271u, 32u);
    if (reorder && P->sample_order != BDPT_ORDER_ROULETTE_CLASS) {
        /* ADAPTIVE (and AUTO): by classes only if the first sample index of the call enters the light loop */
        Lane probe = L;
        probe.w = J->seeds[sidx] * pow_u32(stride, (uint32_t)(J->sample_offset + J->num_samples));
        reorder = !trace_sample(&probe, J, sc, x, y).early;
    }
    int taken = 0;
    for (int cls = reorder ? P->max_light_bounces : 1; cls >= 1; cls--)
    for (int si = 0; si < J->spp; si++) {
        const uint32_t start = J->seeds[sidx] * pow_u32(stride, (uint32_t)(J->sample_offset + J->num_samples + si));
        if (reorder && roulette_class(P, start) != cls) continue;
        const int s = taken++;
        int n = J->num_samples + s;                                       /* :588 */
        uint32_t rays0 = L.rays;
        if (P->rng_mode == BDPT_RNG_STRIDED)
            L.w = start;
        else
            L.w = J->seeds[sidx];

        S = trace_sample(&L, J, sc, x, y);
        cnt[0] += L.rays - rays0;
        cnt[S.early ? 2 : 1]++;

        if (S.early && P->rng_mode == BDPT_RNG_STREAM) {
            /* :648/:654 + return: raw value written, history dropped, no seed write-back */
            for (int k = 0; k < 4; k++) acc[k] = S.col[k];
            continue;
        }
        if (n == 0) {                                                     /* :855 */
            for (int k = 0; k < 4; k++) acc[k] = S.col[k];
        } else if (P->accum_mode == BDPT_ACCUM_SUM) {
            for (int k = 0; k < 4; k++) acc[k] = acc[k] + S.col[k];
        } else {                                                          /* :854 */
            float s1 = clm_div((float)n, (float)(n + 1));
            float s2 = clm_div(1.0f, (float)(n + 1));
            for (int k = 0; k < 4; k++) acc[k] = clm_fma(S.col[k], s2, acc[k] * s1);
        }
        if (P->rng_mode == BDPT_RNG_STREAM) J->seeds[sidx] = L.w;         /* :861 */
    }

    if (J->aov_eye_hit) { J->aov_eye_hit[pix * 2] = S.eye_type; J->aov_eye_hit[pix * 2 + 1] = S.eye_index; }
    if (J->aov_radiance) for (int k = 0; k < 4; k++) J->aov_radiance[pix * 4 + k] = S.col[k];
    if (J->aov_hit_hash) J->aov_hit_hash[pix] = L.hash;
    if (J->aov_rng_draws) J->aov_rng_draws[pix] = L.draws;
    if (J->aov_rays) J->aov_rays[pix] = L.rays;
}

int orc_render(orc_job *J)
{
    if (!J || !J->camera || !J->params || !J->seeds || !J->accum) return -1;
    if (J->x0 < 0 || J->y0 < 0 || J->x0 + J->w > J->width || J->y0 + J->h > J->height) return -1;
    if (J->params->light_index < 0 || J->params->light_index >= J->n_spheres) return -1;
    Scene sc;
    sc.spheres = J->spheres;     sc.n_spheres = J->n_spheres;
    sc.planes = J->planes;       sc.n_planes = J->n_planes;
    sc.triangles = J->triangles; sc.n_triangles = J->n_triangles;
    sc.surfaces = J->surfaces;
    uint64_t c0 = 0, c1 = 0, c2 = 0;
    const int npix = J->w * J->h;
#ifdef _OPENMP
    int nt = J->n_threads > 1 ? J->n_threads : 1;
#pragma omp parallel for schedule(dynamic, 64) num_threads(nt) reduction(+ : c0, c1, c2)
#endif
    for (int i = 0; i < npix; i++) {
        uint64_t cnt[3] = {0, 0, 0};
        if (J->params->integrator == BDPT_INTEGRATOR_PATH_TRACE) path_trace_pixel(J, &sc, J->x0 + i % J->w, J->y0 + i / J->w, cnt);
        else render_pixel(J, &sc, J->x0 + i % J->w, J->y0 + i / J->w, cnt);
        c0 += cnt[0]; c1 += cnt[1]; c2 += cnt[2];
    }
    J->counters[0] = c0; J->counters[1] = c1; J->counters[2] = c2;
    return 0;
}

/* ---- single-function entry points for known-answer tests ------------------- */

static void st3(float *o, clm_f3 v) { o[0] = v.x; o[1] = v.y; o[2] = v.z; }

int orc_ray_intersect(const bdpt_sphere *spheres, int n_spheres,
                      const bdpt_plane *planes, int n_planes,
                      const bdpt_triangle *triangles, int n_triangles,
                      const bdpt_surface *surfaces,
                      const float origin[3], const float dir[3],
                      int origin_type, int origin_index,
                      int out_id[2], float out_pos[3], float out_normal[3])
{
    Scene sc;
    sc.spheres = spheres;     sc.n_spheres = n_spheres;
    sc.planes = planes;       sc.n_planes = n_planes;
    sc.triangles = triangles; sc.n_triangles = n_triangles;
    sc.surfaces = surfaces;
    Lane L;
    memset(&L, 0, sizeof L);
    Full f = ray_intersect(&L, &sc, v3(origin), v3(dir), origin_type, origin_index);
    out_id[0] = f.type;
    out_id[1] = f.index;
    st3(out_pos, f.k.pos);
    st3(out_normal, f.n);
    return f.k.hit;
}

int orc_ray_sphere(const bdpt_sphere *s, const float o[3], const float d[3], float pos[3])
{
    Inter i = raySphereIntersect(v3(o), v3(d), s);
    st3(pos, i.pos);
    return i.hit;
}

int orc_ray_plane(const bdpt_plane *p, const float o[3], const float d[3], float pos[3])
{
    Inter i = rayPlaneIntersect(v3(o), v3(d), p);
    st3(pos, i.pos);
    return i.hit;
}

int orc_ray_triangle(const bdpt_triangle *t, const float o[3], const float d[3], float pos[3])
{
    Inter i = rayTriangleIntersect(v3(o), v3(d), t);
    st3(pos, i.pos);
    return i.hit;
}

float orc_rng_float(uint32_t *w)
{
    Lane L;
    memset(&L, 0, sizeof L);
    L.w = *w;
    float f = get_rng_float(&L);
    *w = L.w;
    return f;
}

void orc_sample_cone(uint32_t *w, const float n[3], float half_angle, float out[3])
{
    Lane L;
    memset(&L, 0, sizeof L);
    L.w = *w;
    st3(out, random_sample_cone(&L, v3(n), half_angle));
    *w = L.w;
}

void orc_sample_sphere(uint32_t *w, const bdpt_sphere *s, float pos[3], float nrm[3])
{
    Lane L;
    memset(&L, 0, sizeof L);
    L.w = *w;
    clm_f3 p, n;
    sample_sphere_point(&L, s, &p, &n);
    st3(pos, p);
    st3(nrm, n);
    *w = L.w;
}

void orc_shoot_ray(const bdpt_camera *cam, float sx, float sy, float out[3])
{
    st3(out, shoot_ray(sx, sy, cam));
}

void orc_sincos(float x, float *s, float *c) { clm_sincos(x, s, c); }
float orc_acosf(float x) { return clm_acosf(x); }
double orc_acos_f64(double x) { return clm_acos_f64(x); }

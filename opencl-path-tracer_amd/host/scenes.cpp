/*
 * scenes.cpp -- primitive constructors, the reference's hard-coded scene and
 * the generated scenes of the benchmark configurations, and the seed fill.
 */
#include <cstring>

#include "bdpt_host.h"

namespace bdpt_host {

/* ---- src/geometry/primitives.cpp:3-68 ------------------------------------- */

bdpt_sphere make_sphere(float cx, float cy, float cz, float radius, uint32_t surf)
{
    bdpt_sphere s;
    std::memset(&s, 0, sizeof s);
    s.center[0] = cx; s.center[1] = cy; s.center[2] = cz; s.center[3] = 0.0f;
    s.radius = radius;
    s.surface = surf;
    return s;
}

bdpt_triangle make_triangle(vec3 p0, vec3 p1, vec3 p2, uint32_t surf)
{
    bdpt_triangle t;
    std::memset(&t, 0, sizeof t);
    t.p0[0] = p0.x; t.p0[1] = p0.y; t.p0[2] = p0.z;
    t.p1[0] = p1.x; t.p1[1] = p1.y; t.p1[2] = p1.z;
    t.p2[0] = p2.x; t.p2[1] = p2.y; t.p2[2] = p2.z;
    t.surface = surf;
    return t;
}

bdpt_plane make_plane(vec3 normal, vec3 point, uint32_t surf)
{
    bdpt_plane p;
    std::memset(&p, 0, sizeof p);
    p.normal[0] = normal.x; p.normal[1] = normal.y; p.normal[2] = normal.z;
    p.point[0] = point.x; p.point[1] = point.y; p.point[2] = point.z;
    p.surface = surf;
    return p;
}

bdpt_surface make_surface(float r, float g, float b, float a, float er, float eg, float eb, float ea)
{
    bdpt_surface s;
    s.color[0] = r; s.color[1] = g; s.color[2] = b; s.color[3] = a;
    s.emissive[0] = er; s.emissive[1] = eg; s.emissive[2] = eb; s.emissive[3] = ea;
    return s;
}

/* ---- src/renderer/main.cpp:270-422 ----------------------------------------- */

static void cornell_spheres_and_surfaces(Scene &sc)
{
    sc.spheres.resize(3);
    sc.spheres[1] = make_sphere(0.0f, -2.0f, 0.0f, 1.0f, 0);             /* main.cpp:281-283 */
    sc.spheres[0] = make_sphere(0.0f, +1.50f, 0.0f, 0.2f, 2);            /* :288-290, the light */
    sc.spheres[2] = make_sphere(-2.0f, -2.0f, -2.0f, 1.0f, 0);           /* :296-298 */

    const float lPower = 5.0f;                                            /* :417 */
    sc.surfaces.resize(5);
    sc.surfaces[0] = make_surface(1.0f, 1.0f, 0.0f, 1.0f);               /* :412 yellow */
    sc.surfaces[1] = make_surface(1.0f, 1.0f, 1.0f, 1.0f);               /* :414 white */
    sc.surfaces[2] = make_surface(1.0f, 1.0f, 1.0f, 1.0f, lPower, lPower, lPower, lPower);   /* :416-418 */
    sc.surfaces[3] = make_surface(1.0f, 0.0f, 0.0f, 1.0f);               /* :420 red */
    sc.surfaces[4] = make_surface(0.0f, 1.0f, 0.0f, 1.0f);               /* :422 green */
}

Scene cornell_scene()
{
    Scene sc;
    cornell_spheres_and_surfaces(sc);
    const float B = 3.0f;                                                 /* boxSize, main.cpp:334 */
    auto T = [&](vec3 p0, vec3 p1, vec3 p2, uint32_t surf) { sc.triangles.push_back(make_triangle(p0, p1, p2, surf)); };
    /* main.cpp:335-394; where the reference assigns p2 before p1 the arguments below are already in p0,p1,p2 order */
    T({-B, -B, -B}, {-B, B, -B}, {-B, -B, B}, 1);    /* 0 */
    T({-B, B, B}, {-B, -B, B}, {-B, B, -B}, 1);      /* 1 */
    T({B, -B, -B}, {B, -B, B}, {B, B, -B}, 1);       /* 2 */
    T({B, B, B}, {B, B, -B}, {B, -B, B}, 1);         /* 3 */
    T({-B, -B, -B}, {-B, -B, B}, {B, -B, -B}, 1);    /* 4 */
    T({B, -B, B}, {B, -B, -B}, {-B, -B, B}, 1);      /* 5 */
    T({-B, B, -B}, {B, B, -B}, {-B, B, B}, 1);       /* 6 */
    T({B, B, B}, {-B, B, B}, {B, B, -B}, 1);         /* 7 */
    T({-B, -B, -B}, {B, -B, -B}, {-B, B, -B}, 3);    /* 8 */
    T({B, B, -B}, {-B, B, -B}, {B, -B, -B}, 3);      /* 9 */
    T({-B, -B, B}, {-B, B, B}, {B, -B, B}, 4);       /* 10 */
    T({B, B, B}, {B, -B, B}, {-B, B, B}, 4);         /* 11 */
    sc.view_dir = {1.0f, 0.1f, -0.1f};                                    /* main.cpp:519 */
    sc.view_radius = 8.0f;                                                /* main.cpp:517 */
    return sc;
}

Scene cornell_planes_scene()
{
    Scene sc;
    cornell_spheres_and_surfaces(sc);
    const float P = 3.0f;                                                 /* planeSize, main.cpp:304 */
    sc.planes.push_back(make_plane({0, 1, 0}, {0, -P, 0}, 1));            /* main.cpp:306-308 */
    sc.planes.push_back(make_plane({0, -1, 0}, {0, P, 0}, 1));            /* :310-312 */
    sc.planes.push_back(make_plane({1, 0, 0}, {-P, 0, 0}, 1));            /* :314-316 */
    sc.planes.push_back(make_plane({-1, 0, 0}, {P, 0, 0}, 1));            /* :318-320 */
    sc.planes.push_back(make_plane({0, 0, 1}, {0, 0, -P}, 3));            /* :322-324, coloured like the triangle box */
    sc.planes.push_back(make_plane({0, 0, -1}, {0, 0, P}, 4));            /* :326-328 */
    return sc;
}

/* Reconstruction of the sphere arrangement whose screenshots the reference's
 * README shows and whose coordinates survive only as comments
 * (main.cpp:284-302): a small light at the origin, unit spheres around it. */
Scene outdoor_scene()
{
    Scene sc;
    cornell_spheres_and_surfaces(sc);
    sc.spheres.clear();
    sc.spheres.push_back(make_sphere(0.0f, 0.0f, 0.0f, 0.2f, 2));
    sc.spheres.push_back(make_sphere(+1.0f, 0.0f, +1.0f, 1.0f, 0));
    sc.spheres.push_back(make_sphere(-1.0f, 0.0f, +1.0f, 1.0f, 0));
    sc.spheres.push_back(make_sphere(+1.0f, 0.0f, -1.0f, 1.0f, 0));
    sc.spheres.push_back(make_sphere(-1.0f, 0.0f, -1.0f, 1.0f, 0));
    sc.spheres.push_back(make_sphere(0.0f, 1.3f, 0.0f, 1.0f, 0));
    sc.view_dir = {1.0f, 0.6f, -0.4f};
    sc.view_radius = 8.0f;
    return sc;
}

static float lcg_unit(uint32_t &state)
{
    state = state * 1664525u + 1013904223u;
    return (float)(state >> 8) * (1.0f / 16777216.0f);
}

Scene soup_scene(int n_soup, uint32_t seed)
{
    Scene sc = cornell_scene();
    static const uint32_t surf_of[4] = {0, 1, 3, 4};
    uint32_t st = seed;
    sc.triangles.reserve(sc.triangles.size() + (size_t)(n_soup > 0 ? n_soup : 0));
    for (int i = 0; i < n_soup; i++) {
        float v[9];
        for (int k = 0; k < 9; k++) v[k] = lcg_unit(st);
        vec3 c = {v[0] * 5.0f - 2.5f, v[1] * 5.0f - 2.5f, v[2] * 5.0f - 2.5f};
        vec3 e1 = {v[3] * 0.1f - 0.05f, v[4] * 0.1f - 0.05f, v[5] * 0.1f - 0.05f};
        vec3 e2 = {v[6] * 0.1f - 0.05f, v[7] * 0.1f - 0.05f, v[8] * 0.1f - 0.05f};
        sc.triangles.push_back(make_triangle(c, {c.x + e1.x, c.y + e1.y, c.z + e1.z},
                                             {c.x + e2.x, c.y + e2.y, c.z + e2.z}, surf_of[i & 3]));
    }
    return sc;
}

/* ---- seeds: main.cpp:456-460 fills W*H*4 words with rand() ----------------- */

static uint32_t mix32(uint32_t x)
{
    x ^= x >> 16; x *= 0x7feb352du;
    x ^= x >> 15; x *= 0x846ca68bu;
    x ^= x >> 16;
    return x;
}

void make_seeds(int width, int height, uint32_t global_seed, uint32_t *out)
{
    const size_t n = (size_t)width * height * 4;
    const uint32_t salt = mix32(global_seed ^ 0x9E3779B9u);
    for (size_t i = 0; i < n; i++) {
        uint32_t v = mix32((uint32_t)i + salt);
        if ((i & 3) == 3) v |= 1u;
        out[i] = v;
    }
}

}  // namespace bdpt_host

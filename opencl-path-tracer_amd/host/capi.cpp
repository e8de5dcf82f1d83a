/*
 * capi.cpp -- C-ABI over the host-side scene, camera and seed helpers, for
 * bindings that cannot call C++ (the Python tests and the benchmark driver).
 */
#include <cstdio>
#include <string>

#include "bdpt_host.h"

using namespace bdpt_host;

struct bdpt_host_scene {
    Scene sc;
};

extern "C" {

bdpt_host_scene *bdpt_host_scene_create(int kind, int n_soup, uint32_t seed)
{
    bdpt_host_scene *s = new bdpt_host_scene();
    switch (kind) {
    case 0: s->sc = cornell_scene(); break;
    case 1: s->sc = cornell_planes_scene(); break;
    case 2: s->sc = outdoor_scene(); break;
    case 3: s->sc = soup_scene(n_soup, seed); break;
    default: delete s; return nullptr;
    }
    return s;
}

void bdpt_host_scene_destroy(bdpt_host_scene *s) { delete s; }

int bdpt_host_scene_counts(const bdpt_host_scene *s, int counts[4])
{
    if (!s || !counts) return -1;
    counts[0] = (int)s->sc.spheres.size();
    counts[1] = (int)s->sc.planes.size();
    counts[2] = (int)s->sc.triangles.size();
    counts[3] = (int)s->sc.surfaces.size();
    return 0;
}

const bdpt_sphere *bdpt_host_scene_spheres(const bdpt_host_scene *s) { return s ? s->sc.spheres.data() : nullptr; }
const bdpt_plane *bdpt_host_scene_planes(const bdpt_host_scene *s) { return s ? s->sc.planes.data() : nullptr; }
const bdpt_triangle *bdpt_host_scene_triangles(const bdpt_host_scene *s) { return s ? s->sc.triangles.data() : nullptr; }
const bdpt_surface *bdpt_host_scene_surfaces(const bdpt_host_scene *s) { return s ? s->sc.surfaces.data() : nullptr; }

void bdpt_host_camera_reset(float radius, const float orientation[3], const float dir[3], bdpt_camera *out)
{
    ModelCamera cam;
    cam.setRadius(radius);
    cam.setOrientation({orientation[0], orientation[1], orientation[2]});
    cam.reset({dir[0], dir[1], dir[2]});
    *out = cam.getCLCamera();
}

void bdpt_host_scene_default_camera(const bdpt_host_scene *s, bdpt_camera *out)
{
    const float orientation[3] = {0.0f, -1.0f, 0.0f};                     /* main.cpp:518 */
    const float dir[3] = {s->sc.view_dir.x, s->sc.view_dir.y, s->sc.view_dir.z};
    bdpt_host_camera_reset(s->sc.view_radius, orientation, dir, out);
}

void bdpt_host_camera_orbit(float radius, const float orientation[3], const float dir[3], float speed_x, float speed_y,
                            double dx, double dy, int steps, bdpt_camera *out, int *changed)
{
    ModelCamera cam;
    cam.setSpeedX(speed_x);
    cam.setSpeedY(speed_y);
    cam.setRadius(radius);
    cam.setOrientation({orientation[0], orientation[1], orientation[2]});
    cam.reset({dir[0], dir[1], dir[2]});
    out[0] = cam.getCLCamera();
    cam.handleCursor(true, false, 0.0, 0.0);                              /* press: becomes active, no motion yet */
    for (int i = 0; i < steps; i++) {
        cam.handleCursor(true, false, dx, dy);
        out[i + 1] = cam.getCLCamera();
        if (changed) changed[i] = cam.isChanged();
    }
}

void bdpt_host_camera_zoom(float radius, const float orientation[3], const float dir[3], float scroll_speed, double dy,
                           int steps, bdpt_camera *out)
{
    ModelCamera cam;
    cam.setScrollSpeed(scroll_speed);
    cam.setRadius(radius);
    cam.setOrientation({orientation[0], orientation[1], orientation[2]});
    cam.reset({dir[0], dir[1], dir[2]});
    out[0] = cam.getCLCamera();
    cam.handleCursor(false, true, 0.0, 0.0);                              /* press: scrolling becomes active */
    for (int i = 0; i < steps; i++) {
        cam.handleCursor(false, true, 0.0, dy);
        out[i + 1] = cam.getCLCamera();
    }
}

void bdpt_host_make_seeds(int width, int height, uint32_t global_seed, uint32_t *out)
{
    make_seeds(width, height, global_seed, out);
}

int bdpt_host_write_pfm(const char *path, const float *rgba, int width, int height)
{
    return write_pfm(path, rgba, width, height) ? 0 : -1;
}

static void put_err(char *err, int err_len, const std::string &what)
{
    if (err && err_len > 0) std::snprintf(err, (size_t)err_len, "%s", what.c_str());
}

bdpt_host_scene *bdpt_host_scene_from_arrays(const bdpt_sphere *spheres, int n_spheres, const bdpt_plane *planes, int n_planes,
                                            const bdpt_triangle *triangles, int n_triangles, const bdpt_surface *surfaces,
                                            int n_surfaces, const float view[4])
{
    if (n_spheres < 0 || n_planes < 0 || n_triangles < 0 || n_surfaces < 0) return nullptr;
    if ((n_spheres && !spheres) || (n_planes && !planes) || (n_triangles && !triangles) || (n_surfaces && !surfaces)) return nullptr;
    bdpt_host_scene *s = new bdpt_host_scene();
    s->sc.spheres.assign(spheres, spheres + n_spheres);
    s->sc.planes.assign(planes, planes + n_planes);
    s->sc.triangles.assign(triangles, triangles + n_triangles);
    s->sc.surfaces.assign(surfaces, surfaces + n_surfaces);
    if (view) { s->sc.view_dir = {view[0], view[1], view[2]}; s->sc.view_radius = view[3]; }
    return s;
}

bdpt_host_scene *bdpt_host_scene_load(const char *path, char *err, int err_len)
{
    bdpt_host_scene *s = new bdpt_host_scene();
    std::string what;
    if (!path || !load_scene(path, s->sc, &what)) {
        put_err(err, err_len, path ? what : "null path");
        delete s;
        return nullptr;
    }
    return s;
}

int bdpt_host_scene_save(const bdpt_host_scene *s, const char *path, int binary)
{
    if (!s || !path) return -1;
    return (binary ? save_scene_binary(path, s->sc) : save_scene_text(path, s->sc)) ? 0 : -1;
}

void bdpt_host_scene_view(const bdpt_host_scene *s, float dir_radius[4])
{
    dir_radius[0] = s->sc.view_dir.x; dir_radius[1] = s->sc.view_dir.y; dir_radius[2] = s->sc.view_dir.z;
    dir_radius[3] = s->sc.view_radius;
}

int bdpt_host_scene_add_obj(bdpt_host_scene *s, const char *path, uint32_t surface, float scale, const float translate[3],
                            char *err, int err_len)
{
    if (!s || !path) { put_err(err, err_len, "null argument"); return -1; }
    if (surface >= s->sc.surfaces.size()) { put_err(err, err_len, "surface does not exist"); return -1; }
    ObjMesh mesh;
    std::string what;
    if (!load_obj(path, mesh, &what)) { put_err(err, err_len, what); return -1; }
    const vec3 tr = translate ? vec3{translate[0], translate[1], translate[2]} : vec3{0.f, 0.f, 0.f};
    return (int)append_mesh(s->sc, mesh, surface, scale, tr);
}

int bdpt_host_write_image(const char *path, const float *rgba, int width, int height, float scale, float gamma)
{
    return (path && rgba && write_image(path, rgba, width, height, scale, gamma)) ? 0 : -1;
}

}  /* extern "C" */

/*
 * scene_io.cpp -- scene input from files instead of the literals of
 * /root/reference/src/renderer/main.cpp:270-422.
 *
 *   load_obj            Wavefront OBJ -> triangle list.  Follows the reference's
 *                       ModelLoader::load (src/geometry/loader.cpp:8-85): `v`,
 *                       `vt`, `vn` and `f` records, 1-based indices, faces kept in
 *                       file order with the winding of the file (the tracer's
 *                       triangle normal is cross(b-a, c-a), kernel0.cl:285, so the
 *                       winding decides which side is lit).  Where the reference
 *                       needs the caller to say whether faces are `i j k` or
 *                       `i/t/n` (its bVtxOnly flag) and reads out of bounds on
 *                       anything else, this parser accepts `i`, `i/t`, `i//n` and
 *                       `i/t/n` per corner, negative (relative) indices, polygons
 *                       (fan-triangulated), and reports bad input as an error.
 *   load_scene_text     line-oriented scene description (see the grammar below).
 *   save/load_scene_binary   the four upload arrays verbatim behind a header.
 */
#include <cerrno>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <sstream>

#include "bdpt_host.h"

namespace bdpt_host {

namespace {

bool fail(std::string *err, const std::string &what)
{
    if (err) *err = what;
    return false;
}

std::string at_line(const std::string &path, int line, const std::string &what)
{
    std::ostringstream o;
    o << path << ":" << line << ": " << what;
    return o.str();
}

/* splits on blanks; '#' starts a comment */
std::vector<std::string> tokens(const std::string &line)
{
    std::vector<std::string> out;
    std::istringstream in(line.substr(0, line.find('#')));
    std::string t;
    while (in >> t) out.push_back(t);
    return out;
}

bool to_float(const std::string &s, float &v)
{
    char *end = nullptr;
    errno = 0;
    v = std::strtof(s.c_str(), &end);
    return end != s.c_str() && *end == '\0' && errno == 0 && std::isfinite(v);
}

bool to_int(const std::string &s, long &v)
{
    char *end = nullptr;
    errno = 0;
    v = std::strtol(s.c_str(), &end, 10);
    return end != s.c_str() && *end == '\0' && errno == 0;
}

bool floats(const std::vector<std::string> &t, size_t first, size_t n, float *out)
{
    if (t.size() < first + n) return false;
    for (size_t i = 0; i < n; i++)
        if (!to_float(t[first + i], out[i])) return false;
    return true;
}

/* position index of one face corner: "7", "7/2", "7//3", "7/2/3", "-1" */
bool corner_index(const std::string &tok, size_t n_vertices, size_t &idx)
{
    long v;
    if (!to_int(tok.substr(0, tok.find('/')), v) || v == 0) return false;
    if (v < 0) v += (long)n_vertices + 1;                                 /* relative to the vertices read so far */
    if (v < 1 || (size_t)v > n_vertices) return false;
    idx = (size_t)v - 1;                                                  /* loader.cpp:68: vVertices[index - 1] */
    return true;
}

}  // namespace

bool load_obj(const std::string &path, ObjMesh &mesh, std::string *err)
{
    std::ifstream in(path);
    if (!in) return fail(err, path + ": cannot open");                   /* loader.cpp:19-21 returns NULL */
    mesh = ObjMesh();
    std::string line;
    int ln = 0;
    while (std::getline(in, line)) {
        ln++;
        const std::vector<std::string> t = tokens(line);
        if (t.empty()) continue;
        if (t[0] == "v") {                                                /* loader.cpp:33-38 */
            float p[3];
            if (!floats(t, 1, 3, p)) return fail(err, at_line(path, ln, "v needs three numbers"));
            mesh.positions.push_back({p[0], p[1], p[2]});
        } else if (t[0] == "vt") {                                        /* loader.cpp:39-44 */
            mesh.n_uvs++;
        } else if (t[0] == "vn") {                                        /* loader.cpp:45-50 */
            mesh.n_normals++;
        } else if (t[0] == "f") {                                         /* loader.cpp:51-79 */
            if (t.size() < 4) return fail(err, at_line(path, ln, "f needs at least three corners"));
            std::vector<size_t> c(t.size() - 1);
            for (size_t i = 1; i < t.size(); i++)
                if (!corner_index(t[i], mesh.positions.size(), c[i - 1]))
                    return fail(err, at_line(path, ln, "bad vertex reference '" + t[i] + "'"));
            for (size_t i = 1; i + 1 < c.size(); i++) {                   /* a triangle is its own fan */
                mesh.indices.push_back((uint32_t)c[0]);
                mesh.indices.push_back((uint32_t)c[i]);
                mesh.indices.push_back((uint32_t)c[i + 1]);
            }
        }                                                                 /* everything else (o, g, s, usemtl, ...) is skipped, as in the reference */
    }
    return true;
}

size_t append_mesh(Scene &scene, const ObjMesh &mesh, uint32_t surface, float scale, vec3 translate)
{
    auto place = [&](uint32_t i) {
        const vec3 &p = mesh.positions[i];
        return vec3{p.x * scale + translate.x, p.y * scale + translate.y, p.z * scale + translate.z};
    };
    const size_t n = mesh.indices.size() / 3;
    scene.triangles.reserve(scene.triangles.size() + n);
    for (size_t f = 0; f < n; f++)
        scene.triangles.push_back(make_triangle(place(mesh.indices[3 * f]), place(mesh.indices[3 * f + 1]),
                                                place(mesh.indices[3 * f + 2]), surface));
    return n;
}

/*
 * Text scene description, one record per line, '#' comments:
 *
 *   surface  r g b a  er eg eb ea          Surface (primitives.cpp:56-68); ea != 0 marks a light
 *   sphere   cx cy cz radius surface       the light must be sphere 0 unless params say otherwise
 *   plane    nx ny nz  px py pz  surface
 *   triangle ax ay az  bx by bz  cx cy cz  surface
 *   obj      file surface [scale s] [translate x y z]     path relative to the scene file
 *   view     dx dy dz radius               ModelCamera::reset(direction) at this radius (main.cpp:513-519)
 *
 * Surfaces are numbered in order of appearance; a primitive may name a surface
 * that is declared later, the check is made at the end.
 */
bool load_scene_text(const std::string &path, Scene &scene, std::string *err)
{
    std::ifstream in(path);
    if (!in) return fail(err, path + ": cannot open");
    scene = Scene();
    const std::string dir = path.find('/') == std::string::npos ? std::string() : path.substr(0, path.rfind('/') + 1);
    std::string line;
    int ln = 0;
    auto surf_of = [&](const std::vector<std::string> &t, size_t i, uint32_t &s) {
        long v;
        if (i >= t.size() || !to_int(t[i], v) || v < 0) return false;
        s = (uint32_t)v;
        return true;
    };
    while (std::getline(in, line)) {
        ln++;
        const std::vector<std::string> t = tokens(line);
        if (t.empty()) continue;
        float v[9];
        uint32_t s = 0;
        if (t[0] == "surface") {
            if (t.size() != 9 || !floats(t, 1, 8, v)) return fail(err, at_line(path, ln, "surface needs 8 numbers"));
            scene.surfaces.push_back(make_surface(v[0], v[1], v[2], v[3], v[4], v[5], v[6], v[7]));
        } else if (t[0] == "sphere") {
            if (t.size() != 6 || !floats(t, 1, 4, v) || !surf_of(t, 5, s) || !(v[3] > 0.0f))
                return fail(err, at_line(path, ln, "sphere needs cx cy cz radius(>0) surface"));
            scene.spheres.push_back(make_sphere(v[0], v[1], v[2], v[3], s));
        } else if (t[0] == "plane") {
            if (t.size() != 8 || !floats(t, 1, 6, v) || !surf_of(t, 7, s))
                return fail(err, at_line(path, ln, "plane needs nx ny nz px py pz surface"));
            scene.planes.push_back(make_plane({v[0], v[1], v[2]}, {v[3], v[4], v[5]}, s));
        } else if (t[0] == "triangle") {
            if (t.size() != 11 || !floats(t, 1, 9, v) || !surf_of(t, 10, s))
                return fail(err, at_line(path, ln, "triangle needs 9 numbers and a surface"));
            scene.triangles.push_back(make_triangle({v[0], v[1], v[2]}, {v[3], v[4], v[5]}, {v[6], v[7], v[8]}, s));
        } else if (t[0] == "obj") {
            if (t.size() < 3 || !surf_of(t, 2, s)) return fail(err, at_line(path, ln, "obj needs a file and a surface"));
            float scale = 1.0f;
            vec3 tr{0.0f, 0.0f, 0.0f};
            for (size_t i = 3; i < t.size();) {
                if (t[i] == "scale" && floats(t, i + 1, 1, v)) { scale = v[0]; i += 2; }
                else if (t[i] == "translate" && floats(t, i + 1, 3, v)) { tr = {v[0], v[1], v[2]}; i += 4; }
                else return fail(err, at_line(path, ln, "obj options are: scale s, translate x y z"));
            }
            ObjMesh mesh;
            std::string sub;
            const std::string file = (!t[1].empty() && t[1][0] == '/') ? t[1] : dir + t[1];
            if (!load_obj(file, mesh, &sub)) return fail(err, at_line(path, ln, sub));
            append_mesh(scene, mesh, s, scale, tr);
        } else if (t[0] == "view") {
            if (t.size() != 5 || !floats(t, 1, 4, v) || !(v[3] > 0.0f))
                return fail(err, at_line(path, ln, "view needs dx dy dz radius(>0)"));
            scene.view_dir = {v[0], v[1], v[2]};
            scene.view_radius = v[3];
        } else {
            return fail(err, at_line(path, ln, "unknown record '" + t[0] + "'"));
        }
    }
    return validate_scene(scene, err ? err : nullptr, path);
}

bool validate_scene(const Scene &scene, std::string *err, const std::string &what)
{
    const size_t ns = scene.surfaces.size();
    if (scene.spheres.empty()) return fail(err, what + ": no sphere (sphere 0 is the light, kernel0.cl:663)");
    for (const bdpt_sphere &p : scene.spheres)
        if (p.surface >= ns) return fail(err, what + ": a sphere names a surface that does not exist");
    for (const bdpt_plane &p : scene.planes)
        if (p.surface >= ns) return fail(err, what + ": a plane names a surface that does not exist");
    for (const bdpt_triangle &p : scene.triangles)
        if (p.surface >= ns) return fail(err, what + ": a triangle names a surface that does not exist");
    return true;
}

/* ---- binary form: header + the four upload arrays as bdpt_upload_scene takes them ---- */

namespace {
struct BinHeader {
    char magic[8];                  /* "BDPTSCN1" */
    uint32_t n_spheres, n_planes, n_triangles, n_surfaces;
    float view_dir[3], view_radius;
};
const char kMagic[8] = {'B', 'D', 'P', 'T', 'S', 'C', 'N', '1'};

template <class T> bool put(FILE *f, const std::vector<T> &v)
{
    return v.empty() || std::fwrite(v.data(), sizeof(T), v.size(), f) == v.size();
}
template <class T> bool get(FILE *f, std::vector<T> &v, size_t n)
{
    v.resize(n);
    return n == 0 || std::fread(v.data(), sizeof(T), n, f) == n;
}
}  // namespace

bool save_scene_binary(const std::string &path, const Scene &scene)
{
    FILE *f = std::fopen(path.c_str(), "wb");
    if (!f) return false;
    BinHeader h;
    std::memcpy(h.magic, kMagic, 8);
    h.n_spheres = (uint32_t)scene.spheres.size();
    h.n_planes = (uint32_t)scene.planes.size();
    h.n_triangles = (uint32_t)scene.triangles.size();
    h.n_surfaces = (uint32_t)scene.surfaces.size();
    h.view_dir[0] = scene.view_dir.x; h.view_dir[1] = scene.view_dir.y; h.view_dir[2] = scene.view_dir.z;
    h.view_radius = scene.view_radius;
    bool ok = std::fwrite(&h, sizeof h, 1, f) == 1 && put(f, scene.spheres) && put(f, scene.planes) &&
              put(f, scene.triangles) && put(f, scene.surfaces);
    return (std::fclose(f) == 0) && ok;
}

bool load_scene_binary(const std::string &path, Scene &scene, std::string *err)
{
    FILE *f = std::fopen(path.c_str(), "rb");
    if (!f) return fail(err, path + ": cannot open");
    BinHeader h;
    bool ok = std::fread(&h, sizeof h, 1, f) == 1 && !std::memcmp(h.magic, kMagic, 8);
    if (ok) {
        std::fseek(f, 0, SEEK_END);
        const long size = std::ftell(f);
        const unsigned long long want = sizeof h + 32ull * h.n_spheres + 48ull * h.n_planes + 64ull * h.n_triangles +
                                        32ull * h.n_surfaces;
        ok = size >= 0 && (unsigned long long)size == want;             /* also bounds the allocations below */
        std::fseek(f, (long)sizeof h, SEEK_SET);
    }
    if (!ok) { std::fclose(f); return fail(err, path + ": not a scene file (bad header or size)"); }
    scene = Scene();
    ok = get(f, scene.spheres, h.n_spheres) && get(f, scene.planes, h.n_planes) && get(f, scene.triangles, h.n_triangles) &&
         get(f, scene.surfaces, h.n_surfaces);
    std::fclose(f);
    if (!ok) return fail(err, path + ": short read");
    scene.view_dir = {h.view_dir[0], h.view_dir[1], h.view_dir[2]};
    scene.view_radius = h.view_radius;
    return validate_scene(scene, err, path);
}

bool load_scene(const std::string &path, Scene &scene, std::string *err)
{
    char magic[8] = {0};
    FILE *f = std::fopen(path.c_str(), "rb");
    if (!f) return fail(err, path + ": cannot open");
    const size_t n = std::fread(magic, 1, 8, f);
    std::fclose(f);
    if (n == 8 && !std::memcmp(magic, kMagic, 8)) return load_scene_binary(path, scene, err);
    return load_scene_text(path, scene, err);
}

bool save_scene_text(const std::string &path, const Scene &scene)
{
    FILE *f = std::fopen(path.c_str(), "w");
    if (!f) return false;
    /* %.9g round-trips every float */
    for (const bdpt_surface &s : scene.surfaces)
        std::fprintf(f, "surface %.9g %.9g %.9g %.9g %.9g %.9g %.9g %.9g\n", s.color[0], s.color[1], s.color[2], s.color[3],
                     s.emissive[0], s.emissive[1], s.emissive[2], s.emissive[3]);
    for (const bdpt_sphere &s : scene.spheres)
        std::fprintf(f, "sphere %.9g %.9g %.9g %.9g %u\n", s.center[0], s.center[1], s.center[2], s.radius, s.surface);
    for (const bdpt_plane &p : scene.planes)
        std::fprintf(f, "plane %.9g %.9g %.9g %.9g %.9g %.9g %u\n", p.normal[0], p.normal[1], p.normal[2], p.point[0],
                     p.point[1], p.point[2], p.surface);
    for (const bdpt_triangle &t : scene.triangles)
        std::fprintf(f, "triangle %.9g %.9g %.9g %.9g %.9g %.9g %.9g %.9g %.9g %u\n", t.p0[0], t.p0[1], t.p0[2], t.p1[0], t.p1[1],
                     t.p1[2], t.p2[0], t.p2[1], t.p2[2], t.surface);
    std::fprintf(f, "view %.9g %.9g %.9g %.9g\n", scene.view_dir.x, scene.view_dir.y, scene.view_dir.z, scene.view_radius);
    return std::fclose(f) == 0;
}

}  // namespace bdpt_host

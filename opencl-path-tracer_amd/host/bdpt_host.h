/*
 * bdpt_host.h -- the renderer surface above the C-ABI: primitives with the
 * reference's constructors, the orbit camera with its change flag, the scene
 * builders and the per-frame loop.  Headless: GLFW mouse input is replaced by
 * scripted input, the GL present blit by an image read-back.
 *
 * Mirrors, in /root/reference/src:
 *   geometry/primitives.cpp:3-68   -> make_sphere / make_triangle / make_plane / make_surface
 *   renderer/camera.cpp:12-141     -> ModelCamera
 *   renderer/main.cpp:270-422      -> cornell_scene (the hard-coded scene)
 *   renderer/main.cpp:100-172      -> Renderer::frame (mainLoop)
 *   renderer/main.cpp:456-460      -> make_seeds (deterministic instead of rand())
 *   geometry/loader.cpp:8-85       -> load_obj (+ append_mesh: OBJ faces as tracer triangles)
 *   renderer/main.cpp:140-145      -> write_image (PFM / PPM / PNG instead of the present blit)
 */
#ifndef BDPT_HOST_H
#define BDPT_HOST_H

#include <cstdint>
#include <string>
#include <vector>

#include "../../include/bdpt.h"

namespace bdpt_host {

struct vec3 {
    float x, y, z;
};

/* primitives.cpp:3-68 -- same defaults as the reference's constructors */
bdpt_sphere make_sphere(float cx = 0.f, float cy = 0.f, float cz = 0.f, float radius = 1.f, uint32_t surf = 0);
bdpt_triangle make_triangle(vec3 p0, vec3 p1, vec3 p2, uint32_t surf);
bdpt_plane make_plane(vec3 normal, vec3 point, uint32_t surf);
bdpt_surface make_surface(float r = 0.f, float g = 0.f, float b = 0.f, float a = 0.f, float er = 0.f, float eg = 0.f,
                          float eb = 0.f, float ea = 0.f);

struct Scene {
    std::vector<bdpt_sphere> spheres;
    std::vector<bdpt_plane> planes;
    std::vector<bdpt_triangle> triangles;
    std::vector<bdpt_surface> surfaces;
    /* default view of the scene: ModelCamera::reset(direction) at this radius */
    vec3 view_dir{1.0f, 0.1f, -0.1f};
    float view_radius = 8.0f;
};

/* main.cpp:270-422: 3 spheres, 12 triangles (a +-3 box), 5 surfaces */
Scene cornell_scene();
/* the same box with its six walls as infinite planes (the block the reference
 * keeps commented out at main.cpp:306-328), for the plane intersector */
Scene cornell_planes_scene();
/* light sphere at the origin inside a ring of five unit spheres, no walls */
Scene outdoor_scene();
/* cornell_scene plus n_soup small random triangles (fixed generator) */
Scene soup_scene(int n_soup, uint32_t seed = 0xC0FFEEu);

/* width*height uint4 seeds in the reference's layout (index x*height+y), from a
 * fixed integer hash of (global_seed, index); .w is forced odd so that the
 * multiplicative generator of kernel0.cl:106-109 keeps its full period. */
void make_seeds(int width, int height, uint32_t global_seed, uint32_t *out);

/* camera.cpp:7-141 */
class ModelCamera {
public:
    ModelCamera();
    void setSpeedX(float s) { fSpeedX = s; }
    void setSpeedY(float s) { fSpeedY = s; }
    float getSpeedX() const { return fSpeedX; }
    float getSpeedY() const { return fSpeedY; }
    void setRadius(float r) { fRadius = r; }
    void setScrollSpeed(float s) { fScrollSpeed = s; }
    void setOrientation(vec3 v) { vOrientation = v; }
    void reset(vec3 vDir);                       /* camera.cpp:44-53 */
    void setPose(const bdpt_camera &c);          /* restores eye, look direction and up from a CLCamera */
    vec3 getEye() const { return vPos; }
    vec3 getLookAt() const { return vLookAt; }
    int isChanged() const { return iState == CAM_ACTIVE; }   /* camera.cpp:62-65: held, not moved */
    /* camera.cpp:67-133 with the GLFW polls replaced by arguments: state of the
     * left / middle buttons and the cursor displacement since the last call. */
    void handleCursor(bool left_down, bool middle_down, double dx, double dy);
    bdpt_camera getCLCamera() const;             /* camera.cpp:135-141 */

private:
    static const int CAM_ACTIVE = 1, CAM_INACTIVE = 0, SCROLL_ACTIVE = 1, SCROLL_INACTIVE = 0;
    vec3 vLookAt{0, 0, -1}, vUp{0, 1, 0}, vPos{0, 0, 0}, vOrientation{0, 1, 0};
    float fSpeedX = 0, fSpeedY = 0, fScrollSpeed = 0, fRadius = 1;
    int iState = CAM_INACTIVE, scrollState = SCROLL_INACTIVE;
};

/* one entry of a scripted input track: what the mouse does during one frame */
struct InputFrame {
    bool left_down = false, middle_down = false;
    double dx = 0, dy = 0;
};

/* main.cpp:100-172 (mainLoop) and :513-552 (set-up) without the window */
class Renderer {
public:
    Renderer(int device, int width, int height);
    /* Several GPUs of the node behind one renderer (bdpt_group_*): every frame's samples are dealt to the
     * devices and the image is the combined mean.  The reference has one device; with a list of one this is
     * still a group (counter-addressed generator, sum accumulation), not the single-device renderer above. */
    Renderer(const std::vector<int> &devices, int width, int height, int combine = BDPT_COMBINE_AUTO);
    ~Renderer();
    Renderer(const Renderer &) = delete;
    Renderer &operator=(const Renderer &) = delete;

    void setScene(const Scene &scene);           /* uploads; resets the camera to the scene's view */
    void setSeeds(uint32_t global_seed);
    void setParams(const bdpt_params &p);
    ModelCamera &camera() { return cam; }
    /* One frame: launch `spp` samples (the reference's SAMPLES is 1), then feed
     * the camera one InputFrame; while the camera reports a change, zero the
     * sample counter and upload the new camera (main.cpp:149-156). */
    void frame(const InputFrame &in, int spp = 1);
    int numSamples() const;
    std::vector<float> readImage();              /* width*height*4 floats */
    /* Progressive state to a file and back: image size, sample counter, camera pose, accumulator and
     * generator state.  A render resumed from a checkpoint continues bit for bit as if it had never
     * stopped (same scene and parameters assumed).  The reference cannot do this: its accumulator lives
     * in a GL texture and its seeds are re-drawn at every start (main.cpp:456-460). */
    bool saveCheckpoint(const std::string &path);
    bool loadCheckpoint(const std::string &path, std::string *err = nullptr);
    float lastLaunchMs();
    /* runs per tile of a render call (bdpt_set_launch_chunks): 0 chosen per launch, 1 unsliced; every device of a group */
    void setLaunchChunks(int chunks);
    /* device memory a context may take to keep a launch's sample order (bdpt_set_order_scratch_limit); every device of a group */
    void setOrderScratchLimit(size_t bytes);
    bdpt_ctx *ctx() { return c; }
    bdpt_group *group() { return g; }
    int numDevices() const;
    int width() const { return w; }
    int height() const { return h; }
    bool verbose = true;                         /* "numSamples: N" every 10 samples (main.cpp:168-169) */

private:
    void check(int rc, const char *what);        /* main.cpp:41-46: print and exit */
    void initCamera();
    bdpt_ctx *c = nullptr;                       /* single device ... */
    bdpt_group *g = nullptr;                     /* ... or a group of them */
    int w, h;
    ModelCamera cam;
};

/* ---- scene input from files (scene_io.cpp) ---- */

/* what ModelLoader::load keeps of an OBJ file that matters to the tracer */
struct ObjMesh {
    std::vector<vec3> positions;
    std::vector<uint32_t> indices;      /* 3 per triangle, 0-based into positions, file order and winding */
    size_t n_uvs = 0, n_normals = 0;    /* `vt` / `vn` records seen (the tracer has no use for their values) */
};
bool load_obj(const std::string &path, ObjMesh &mesh, std::string *err = nullptr);
/* adds the mesh's faces as triangles of `surface`, each vertex scaled then translated; returns the number added */
size_t append_mesh(Scene &scene, const ObjMesh &mesh, uint32_t surface, float scale = 1.0f, vec3 translate = {0.f, 0.f, 0.f});
bool validate_scene(const Scene &scene, std::string *err, const std::string &what);
bool load_scene_text(const std::string &path, Scene &scene, std::string *err = nullptr);
bool save_scene_text(const std::string &path, const Scene &scene);
bool load_scene_binary(const std::string &path, Scene &scene, std::string *err = nullptr);
bool save_scene_binary(const std::string &path, const Scene &scene);
bool load_scene(const std::string &path, Scene &scene, std::string *err = nullptr);   /* binary or text, by its first bytes */

/* ---- image output (renderer.cpp, image_io.cpp) ---- */

/* little-endian PFM (RGB, bottom row first) of an RGBA float image */
bool write_pfm(const std::string &path, const float *rgba, int width, int height);
/* 8-bit RGB: clamp(v * scale, 0, 1) ^ (1 / gamma), rounded to 0..255, top row first */
std::vector<uint8_t> to_rgb8(const float *rgba, int width, int height, float scale = 1.0f, float gamma = 1.0f);
bool write_ppm(const std::string &path, const float *rgba, int width, int height, float scale = 1.0f, float gamma = 1.0f);
bool write_png(const std::string &path, const float *rgba, int width, int height, float scale = 1.0f, float gamma = 1.0f);
/* by extension: .pfm, .ppm, .png */
bool write_image(const std::string &path, const float *rgba, int width, int height, float scale = 1.0f, float gamma = 1.0f);

}  // namespace bdpt_host

/* ---- C-ABI for bindings (Python tests and benchmark) ---------------------- */
extern "C" {
typedef struct bdpt_host_scene bdpt_host_scene;
/* kind: 0 cornell, 1 cornell with plane walls, 2 outdoor, 3 soup (n_soup, seed) */
bdpt_host_scene *bdpt_host_scene_create(int kind, int n_soup, uint32_t seed);
void bdpt_host_scene_destroy(bdpt_host_scene *s);
int bdpt_host_scene_counts(const bdpt_host_scene *s, int counts[4]);   /* spheres, planes, triangles, surfaces */
const bdpt_sphere *bdpt_host_scene_spheres(const bdpt_host_scene *s);
const bdpt_plane *bdpt_host_scene_planes(const bdpt_host_scene *s);
const bdpt_triangle *bdpt_host_scene_triangles(const bdpt_host_scene *s);
const bdpt_surface *bdpt_host_scene_surfaces(const bdpt_host_scene *s);
/* ModelCamera: radius, orientation and reset direction in, CLCamera out */
void bdpt_host_scene_default_camera(const bdpt_host_scene *s, bdpt_camera *out);
void bdpt_host_camera_reset(float radius, const float orientation[3], const float dir[3], bdpt_camera *out);
/* reset(dir), press the left button, then `steps` cursor moves of (dx, dy);
 * out receives steps+1 cameras (the first is the reset pose); changed[i] is
 * isChanged() after step i. */
void bdpt_host_camera_orbit(float radius, const float orientation[3], const float dir[3], float speed_x, float speed_y,
                            double dx, double dy, int steps, bdpt_camera *out, int *changed);
/* reset(dir), press the middle button, then `steps` cursor moves of dy with the button held (camera.cpp:77-89:
 * the radius shrinks by dy * scroll_speed per move and the pose is rebuilt from the eye direction); out receives
 * steps+1 cameras */
void bdpt_host_camera_zoom(float radius, const float orientation[3], const float dir[3], float scroll_speed, double dy,
                           int steps, bdpt_camera *out);
void bdpt_host_make_seeds(int width, int height, uint32_t global_seed, uint32_t *out);
int bdpt_host_write_pfm(const char *path, const float *rgba, int width, int height);
/* a scene from caller-held arrays (copied); view = direction xyz + radius of the default camera, or NULL */
bdpt_host_scene *bdpt_host_scene_from_arrays(const bdpt_sphere *spheres, int n_spheres, const bdpt_plane *planes, int n_planes,
                                            const bdpt_triangle *triangles, int n_triangles, const bdpt_surface *surfaces,
                                            int n_surfaces, const float view[4]);
/* scene files: NULL on failure, with the reason in err (if err_len > 0) */
bdpt_host_scene *bdpt_host_scene_load(const char *path, char *err, int err_len);
int bdpt_host_scene_save(const bdpt_host_scene *s, const char *path, int binary);
void bdpt_host_scene_view(const bdpt_host_scene *s, float dir_radius[4]);
/* appends the faces of an OBJ file as triangles; returns how many, or -1 with the reason in err */
int bdpt_host_scene_add_obj(bdpt_host_scene *s, const char *path, uint32_t surface, float scale, const float translate[3],
                            char *err, int err_len);
/* by extension (.pfm / .ppm / .png); 0 on success */
int bdpt_host_write_image(const char *path, const float *rgba, int width, int height, float scale, float gamma);
}

#endif

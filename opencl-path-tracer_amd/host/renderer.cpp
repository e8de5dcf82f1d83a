/*
 * renderer.cpp -- the per-frame loop of the reference (mainLoop,
 * src/renderer/main.cpp:100-172) and its one-time set-up (:513-546) over the
 * C-ABI instead of OpenCL, without a window.
 */
#include <cstdio>
#include <cstdlib>
#include <cstring>

#include "bdpt_host.h"

namespace bdpt_host {

/* main.cpp:41-46 checkErr: print "ERROR: <what> (<code>)" and exit */
void Renderer::check(int rc, const char *what)
{
    if (rc == BDPT_OK) return;
    std::fprintf(stderr, "ERROR: %s (%d): %s\n", what, rc, g ? bdpt_group_last_error(g) : bdpt_last_error(c));
    std::exit(EXIT_FAILURE);
}

Renderer::Renderer(int device, int width, int height) : w(width), h(height)
{
    check(bdpt_create(device, width, height, &c), "bdpt_create");
    initCamera();
}

Renderer::Renderer(const std::vector<int> &devices, int width, int height, int combine) : w(width), h(height)
{
    const int rc = bdpt_group_create(devices.data(), (int)devices.size(), width, height, combine, &g);
    if (rc != BDPT_OK) {
        std::fprintf(stderr, "ERROR: bdpt_group_create (%d): %s\n", rc, bdpt_group_last_error(nullptr));
        std::exit(EXIT_FAILURE);
    }
    initCamera();
}

int Renderer::numDevices() const { return g ? bdpt_group_size(g) : 1; }

void Renderer::initCamera()
{
    cam.setSpeedX(0.03f);                                                 /* main.cpp:514 */
    cam.setSpeedY(0.03f);                                                 /* :515 */
    cam.setRadius(8.0f);                                                  /* :517 */
    cam.setOrientation({0.0f, -1.0f, 0.0f});                              /* :518 */
    cam.reset({1.0f, 0.1f, -0.1f});                                       /* :519 */
}

Renderer::~Renderer()
{
    bdpt_destroy(c);
    bdpt_group_destroy(g);
}

void Renderer::setScene(const Scene &scene)
{
    if (g)
        check(bdpt_group_upload_scene(g, scene.spheres.data(), (int)scene.spheres.size(), scene.planes.data(),
                                      (int)scene.planes.size(), scene.triangles.data(), (int)scene.triangles.size(),
                                      scene.surfaces.data(), (int)scene.surfaces.size()),
              "bdpt_group_upload_scene");
    else
        check(bdpt_upload_scene(c, scene.spheres.data(), (int)scene.spheres.size(), scene.planes.data(),
                                (int)scene.planes.size(), scene.triangles.data(), (int)scene.triangles.size(),
                                scene.surfaces.data(), (int)scene.surfaces.size()),
              "bdpt_upload_scene");                                       /* main.cpp:534-539 */
    cam.setRadius(scene.view_radius);
    cam.reset(scene.view_dir);
    bdpt_camera cl = cam.getCLCamera();
    if (g) check(bdpt_group_set_camera(g, &cl), "bdpt_group_set_camera");
    else check(bdpt_set_camera(c, &cl), "bdpt_set_camera");               /* main.cpp:524-533 */
}

void Renderer::setSeeds(uint32_t global_seed)
{
    std::vector<uint32_t> seeds((size_t)w * h * 4);
    make_seeds(w, h, global_seed, seeds.data());
    if (g) check(bdpt_group_upload_seeds(g, seeds.data()), "bdpt_group_upload_seeds");
    else check(bdpt_upload_seeds(c, seeds.data()), "bdpt_upload_seeds");  /* main.cpp:538 */
}

void Renderer::setParams(const bdpt_params &p)
{
    if (g) check(bdpt_group_set_params(g, &p), "bdpt_group_set_params");
    else check(bdpt_set_params(c, &p), "bdpt_set_params");
}

void Renderer::frame(const InputFrame &in, int spp)
{
    if (g) check(bdpt_group_render(g, spp), "bdpt_group_render");
    else check(bdpt_render(c, spp), "bdpt_render");                       /* main.cpp:108-131 */
    cam.handleCursor(in.left_down, in.middle_down, in.dx, in.dy);         /* :149 */
    if (cam.isChanged()) {                                                /* :150-156 */
        bdpt_camera cl = cam.getCLCamera();
        if (g) check(bdpt_group_set_camera(g, &cl), "bdpt_group_set_camera");
        else check(bdpt_set_camera(c, &cl), "bdpt_set_camera");           /* zeroes the sample counter */
    }
    if (verbose && numSamples() % 10 == 0)                                /* :168-169 */
        std::printf("numSamples: %d\n", numSamples());
}

int Renderer::numSamples() const { return g ? bdpt_group_num_samples(g) : bdpt_num_samples(c); }

std::vector<float> Renderer::readImage()
{
    std::vector<float> img((size_t)w * h * 4);
    if (g) check(bdpt_group_read_image(g, img.data()), "bdpt_group_read_image");
    else check(bdpt_read_accum(c, img.data()), "bdpt_read_accum");
    return img;
}

namespace {
struct CheckpointHeader {
    char magic[8];                  /* "BDPTCKP1" */
    int32_t width, height, num_samples, reserved;
    bdpt_camera camera;
};
const char kCkptMagic[8] = {'B', 'D', 'P', 'T', 'C', 'K', 'P', '1'};
}  // namespace

bool Renderer::saveCheckpoint(const std::string &path)
{
    if (g) return false;                         /* a group's state is spread over its members */
    CheckpointHeader hd;
    std::memset(&hd, 0, sizeof hd);
    std::memcpy(hd.magic, kCkptMagic, 8);
    hd.width = w;
    hd.height = h;
    hd.num_samples = bdpt_num_samples(c);
    hd.camera = cam.getCLCamera();
    std::vector<float> img((size_t)w * h * 4);
    std::vector<uint32_t> seeds((size_t)w * h * 4);
    check(bdpt_read_accum(c, img.data()), "bdpt_read_accum");
    check(bdpt_read_seeds(c, seeds.data()), "bdpt_read_seeds");
    FILE *f = std::fopen(path.c_str(), "wb");
    if (!f) return false;
    const bool ok = std::fwrite(&hd, sizeof hd, 1, f) == 1 && std::fwrite(img.data(), sizeof(float), img.size(), f) == img.size() &&
                    std::fwrite(seeds.data(), sizeof(uint32_t), seeds.size(), f) == seeds.size();
    return (std::fclose(f) == 0) && ok;
}

bool Renderer::loadCheckpoint(const std::string &path, std::string *err)
{
    auto bad = [&](const std::string &what) { if (err) *err = path + ": " + what; return false; };
    if (g) return bad("checkpoints are for single-device renderers");
    FILE *f = std::fopen(path.c_str(), "rb");
    if (!f) return bad("cannot open");
    CheckpointHeader hd;
    std::vector<float> img((size_t)w * h * 4);
    std::vector<uint32_t> seeds((size_t)w * h * 4);
    bool ok = std::fread(&hd, sizeof hd, 1, f) == 1 && !std::memcmp(hd.magic, kCkptMagic, 8);
    if (!ok) { std::fclose(f); return bad("not a checkpoint file"); }
    if (hd.width != w || hd.height != h || hd.num_samples < 0) { std::fclose(f); return bad("image size differs from this renderer's"); }
    ok = std::fread(img.data(), sizeof(float), img.size(), f) == img.size() &&
         std::fread(seeds.data(), sizeof(uint32_t), seeds.size(), f) == seeds.size();
    std::fclose(f);
    if (!ok) return bad("short read");
    cam.setPose(hd.camera);
    check(bdpt_set_camera(c, &hd.camera), "bdpt_set_camera");             /* zeroes the counter; restored below */
    check(bdpt_upload_seeds(c, seeds.data()), "bdpt_upload_seeds");
    check(bdpt_write_accum(c, img.data()), "bdpt_write_accum");
    check(bdpt_set_num_samples(c, hd.num_samples), "bdpt_set_num_samples");
    return true;
}

float Renderer::lastLaunchMs()
{
    float ms = 0.0f;                             /* a group: the slowest member's last launch */
    for (int i = 0; i < numDevices(); i++) {
        float one = 0.0f;
        check(bdpt_last_launch_ms(g ? bdpt_group_member(g, i) : c, &one), "bdpt_last_launch_ms");
        ms = one > ms ? one : ms;
    }
    return ms;
}

void Renderer::setLaunchChunks(int chunks)
{
    for (int i = 0; i < numDevices(); i++)
        check(bdpt_set_launch_chunks(g ? bdpt_group_member(g, i) : c, chunks), "bdpt_set_launch_chunks");
}

void Renderer::setOrderScratchLimit(size_t bytes)
{
    for (int i = 0; i < numDevices(); i++)
        check(bdpt_set_order_scratch_limit(g ? bdpt_group_member(g, i) : c, bytes), "bdpt_set_order_scratch_limit");
}

bool write_pfm(const std::string &path, const float *rgba, int width, int height)
{
    FILE *f = std::fopen(path.c_str(), "wb");
    if (!f) return false;
    std::fprintf(f, "PF\n%d %d\n-1.0\n", width, height);
    std::vector<float> row((size_t)width * 3);
    for (int y = height - 1; y >= 0; y--) {
        for (int x = 0; x < width; x++)
            for (int k = 0; k < 3; k++) row[(size_t)x * 3 + k] = rgba[((size_t)y * width + x) * 4 + k];
        if (std::fwrite(row.data(), sizeof(float), row.size(), f) != row.size()) { std::fclose(f); return false; }
    }
    return std::fclose(f) == 0;
}

}  // namespace bdpt_host

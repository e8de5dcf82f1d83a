/*
 * main.cpp -- headless counterpart of the reference's application
 * (src/renderer/main.cpp:176-564): build the scene, create the context, upload,
 * then run the frame loop for a fixed number of frames with a scripted camera
 * instead of until the window closes, and write the image instead of
 * presenting it.
 *
 *   bdpt_host [--scene cornell|planes|outdoor|soup] [--soup N] [--scene-file F]
 *             [--obj model.obj [--obj-surface I] [--obj-scale S] [--obj-translate X,Y,Z]]
 *             [--width W] [--height H] [--frames F] [--spp S] [--orbit-frames K]
 *             [--seed G] [--device D | --devices D0,D1,... [--combine auto|rccl|copy]]
 *             [--out image.pfm|.ppm|.png [--gamma G]] [--quiet] [--launch-chunks C] [--order-scratch-mib M]
 *             [--resume state.ckpt] [--checkpoint state.ckpt]
 */
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <memory>
#include <string>
#include <vector>

#include "bdpt_host.h"

using namespace bdpt_host;

int main(int argc, char **argv)
{
    std::string scene_name = "cornell", out_path, scene_file, obj_path, resume_path, checkpoint_path;
    int obj_surface = 0;
    float obj_scale = 1.0f, gamma = 1.0f;
    vec3 obj_translate{0.f, 0.f, 0.f};
    int width = 1067, height = 600;                                       /* main.cpp:35-36 */
    int frames = 64, spp = 1, orbit_frames = 0, n_soup = 100000, device = 0;
    uint32_t seed = 0x5EEDu;
    bool quiet = false;
    int launch_chunks = 0;
    long order_scratch_mib = -1;                    /* --order-scratch-mib: -1 = the library's default */
    std::vector<int> devices;                       /* --devices: several GPUs, each frame's samples dealt to them */
    int combine = BDPT_COMBINE_AUTO;
    for (int i = 1; i < argc; i++) {
        auto next = [&](const char *flag) -> const char * {
            if (i + 1 >= argc) { std::fprintf(stderr, "ERROR: %s needs a value\n", flag); std::exit(EXIT_FAILURE); }
            return argv[++i];
        };
        if (!std::strcmp(argv[i], "--scene")) scene_name = next("--scene");
        else if (!std::strcmp(argv[i], "--soup")) n_soup = std::atoi(next("--soup"));
        else if (!std::strcmp(argv[i], "--width")) width = std::atoi(next("--width"));
        else if (!std::strcmp(argv[i], "--height")) height = std::atoi(next("--height"));
        else if (!std::strcmp(argv[i], "--frames")) frames = std::atoi(next("--frames"));
        else if (!std::strcmp(argv[i], "--spp")) spp = std::atoi(next("--spp"));
        else if (!std::strcmp(argv[i], "--orbit-frames")) orbit_frames = std::atoi(next("--orbit-frames"));
        else if (!std::strcmp(argv[i], "--seed")) seed = (uint32_t)std::strtoul(next("--seed"), nullptr, 0);
        else if (!std::strcmp(argv[i], "--device")) device = std::atoi(next("--device"));
        else if (!std::strcmp(argv[i], "--devices")) {
            const char *p = next("--devices");
            while (*p) {
                char *end = nullptr;
                devices.push_back((int)std::strtol(p, &end, 10));
                if (end == p) { std::fprintf(stderr, "ERROR: --devices needs D0,D1,...\n"); return EXIT_FAILURE; }
                p = *end == ',' ? end + 1 : end;
            }
        }
        else if (!std::strcmp(argv[i], "--combine")) {
            const std::string v = next("--combine");
            if (v == "auto") combine = BDPT_COMBINE_AUTO;
            else if (v == "rccl") combine = BDPT_COMBINE_RCCL;
            else if (v == "copy") combine = BDPT_COMBINE_COPY;
            else { std::fprintf(stderr, "ERROR: --combine needs auto, rccl or copy\n"); return EXIT_FAILURE; }
        }
        else if (!std::strcmp(argv[i], "--out")) out_path = next("--out");
        else if (!std::strcmp(argv[i], "--resume")) resume_path = next("--resume");
        else if (!std::strcmp(argv[i], "--checkpoint")) checkpoint_path = next("--checkpoint");
        else if (!std::strcmp(argv[i], "--gamma")) gamma = (float)std::atof(next("--gamma"));
        else if (!std::strcmp(argv[i], "--scene-file")) scene_file = next("--scene-file");
        else if (!std::strcmp(argv[i], "--obj")) obj_path = next("--obj");
        else if (!std::strcmp(argv[i], "--obj-surface")) obj_surface = std::atoi(next("--obj-surface"));
        else if (!std::strcmp(argv[i], "--obj-scale")) obj_scale = (float)std::atof(next("--obj-scale"));
        else if (!std::strcmp(argv[i], "--obj-translate")) {
            if (std::sscanf(next("--obj-translate"), "%f,%f,%f", &obj_translate.x, &obj_translate.y, &obj_translate.z) != 3) {
                std::fprintf(stderr, "ERROR: --obj-translate needs X,Y,Z\n");
                return EXIT_FAILURE;
            }
        }
        else if (!std::strcmp(argv[i], "--launch-chunks")) launch_chunks = std::atoi(next("--launch-chunks"));
        else if (!std::strcmp(argv[i], "--order-scratch-mib")) order_scratch_mib = std::atol(next("--order-scratch-mib"));
        else if (!std::strcmp(argv[i], "--quiet")) quiet = true;
        else { std::fprintf(stderr, "ERROR: unknown argument %s\n", argv[i]); return EXIT_FAILURE; }
    }

    Scene scene;
    std::string why;
    if (!scene_file.empty()) {
        if (!load_scene(scene_file, scene, &why)) { std::fprintf(stderr, "ERROR: %s\n", why.c_str()); return EXIT_FAILURE; }
    }
    else if (scene_name == "cornell") scene = cornell_scene();
    else if (scene_name == "planes") scene = cornell_planes_scene();
    else if (scene_name == "outdoor") scene = outdoor_scene();
    else if (scene_name == "soup") scene = soup_scene(n_soup);
    else { std::fprintf(stderr, "ERROR: unknown scene %s\n", scene_name.c_str()); return EXIT_FAILURE; }

    if (!obj_path.empty()) {
        ObjMesh mesh;
        if (!load_obj(obj_path, mesh, &why)) { std::fprintf(stderr, "ERROR: %s\n", why.c_str()); return EXIT_FAILURE; }
        if (obj_surface < 0 || (size_t)obj_surface >= scene.surfaces.size()) {
            std::fprintf(stderr, "ERROR: --obj-surface %d does not exist\n", obj_surface);
            return EXIT_FAILURE;
        }
        const size_t n = append_mesh(scene, mesh, (uint32_t)obj_surface, obj_scale, obj_translate);
        std::printf("%s: %zu vertices, %zu triangles\n", obj_path.c_str(), mesh.positions.size(), n);   /* loader.cpp:83 */
    }

    std::unique_ptr<Renderer> holder(devices.empty() ? new Renderer(device, width, height)
                                                     : new Renderer(devices, width, height, combine));
    Renderer &r = *holder;
    if (!devices.empty())
        std::printf("%d devices, partial images combined by %s\n", r.numDevices(),
                    bdpt_group_combine(r.group()) == BDPT_COMBINE_RCCL ? "an RCCL reduce" : "device-to-device copies");
    r.verbose = !quiet;
    if (launch_chunks != 0) r.setLaunchChunks(launch_chunks);
    if (order_scratch_mib >= 0) r.setOrderScratchLimit((size_t)order_scratch_mib << 20);
    r.setScene(scene);
    r.setSeeds(seed);
    if (!resume_path.empty()) {
        if (!r.loadCheckpoint(resume_path, &why)) { std::fprintf(stderr, "ERROR: %s\n", why.c_str()); return EXIT_FAILURE; }
        std::printf("resumed %s at %d samples\n", resume_path.c_str(), r.numSamples());
    }
    std::printf("%s: %zu spheres, %zu planes, %zu triangles, %dx%d, %d frames x %d spp\n", bdpt_version(),
                scene.spheres.size(), scene.planes.size(), scene.triangles.size(), width, height, frames, spp);

    const auto t0 = std::chrono::steady_clock::now();
    for (int f = 0; f < frames; f++) {
        InputFrame in;
        /* scripted mouse: hold the left button and drag for the first orbit_frames frames */
        if (f < orbit_frames) { in.left_down = true; in.dx = 0.5 / 0.03; }
        r.frame(in, spp);
    }
    const double sec = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    const double paths = (double)width * height * (double)spp * frames;
    std::printf("%d frames in %.3f s: %.2f Mpaths/s, %d samples accumulated\n", frames, sec, paths / sec / 1e6,
                r.numSamples());

    if (!checkpoint_path.empty()) {
        if (!r.saveCheckpoint(checkpoint_path)) {
            std::fprintf(stderr, "ERROR: cannot write %s\n", checkpoint_path.c_str());
            return EXIT_FAILURE;
        }
        std::printf("checkpoint %s at %d samples\n", checkpoint_path.c_str(), r.numSamples());
    }
    if (!out_path.empty()) {
        std::vector<float> img = r.readImage();
        if (!write_image(out_path, img.data(), width, height, 1.0f, gamma)) {
            std::fprintf(stderr, "ERROR: cannot write %s\n", out_path.c_str());
            return EXIT_FAILURE;
        }
        std::printf("wrote %s\n", out_path.c_str());
    }
    return EXIT_SUCCESS;
}

/*
 * cl_baseline.c -- times the reference's own, unmodified OpenCL kernel on a CPU OpenCL device (PoCL or the like) of
 * this host, as the CPU baseline beside the GPU numbers.  Headless restatement of the reference host's call sequence
 * (/root/reference/src/renderer/main.cpp): context on CL_DEVICE_TYPE_CPU (:242, :255), eight buffers (:424-446), two
 * W x H images (:448-452; plain CL_RGBA / CL_FLOAT images swapped after every frame in place of the GL blit
 * :133-138), run-time build of the kernel file with "-cl-opt-disable" (:468-477) and, for comparison, with the
 * compiler's defaults, the ten kernel arguments (:489-509), blocking uploads (:533-539), and per frame the
 * ImageDescriptor write, one NDRange of (WIDTH, HEIGHT) work-items with a NULL local size and a wait (:108-124),
 * numSamples += 1 (:131).
 *
 * The kernel file is NOT part of this repository: it is read at run time from --kernel, $BDPT_REFERENCE_KERNEL or
 * /root/reference/src/kernel/kernel0.cl, and its two size macros (kernel0.cl:5-6) are rewritten in memory to the
 * requested frame size.  Scene, camera and seeds come from a file the caller writes (bench.py, the tests): the
 * reference's host structs byte for byte, which is what its main() memcpy's into the buffers.
 *
 * Without a CPU OpenCL device, or without the kernel file, prints  "pocl baseline unavailable: <reason>"  and exits 0.
 *
 *   cl_baseline --input scene.bin [--kernel kernel0.cl] [--frames 64] [--warmup 2] [--out image.bin] [--device-type cpu|gpu]
 *
 * --device-type gpu runs the same sequence on a GPU OpenCL device instead (the vendor's OpenCL runtime): an "untuned
 * GPU" data point for whoever has the kernel file beside such a device, and the way tests/test_gpu_system.py drives
 * every call of this file on real hardware with a stand-in kernel of the same signature.
 *
 * Input file (little endian): int32 magic 0x42445054, width, height, n_spheres, n_planes, n_triangles, n_surfaces,
 * reserved; then spheres (32 B each), planes (48 B), triangles (64 B), surfaces (32 B), camera (48 B), seeds
 * (width * height * 4 uint32, index x * height + y as kernel0.cl:590 reads them).
 * Output: one JSON line per build option on stdout; --out: the accumulated image after the last frame of the
 * "-cl-opt-disable" run, width * height * 4 float32, row-major.
 *
 * Build: gcc -O2 -std=c99 tools/cl_baseline.c -o tools/cl_baseline -lOpenCL
 */
#define CL_TARGET_OPENCL_VERSION 120
#define _POSIX_C_SOURCE 200809L
#include <CL/cl.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>
#include <unistd.h>

#define MAGIC 0x42445054

static double now_s(void)
{
    struct timespec t;
    clock_gettime(CLOCK_MONOTONIC, &t);
    return (double)t.tv_sec + 1e-9 * (double)t.tv_nsec;
}

static int unavailable(const char *why)
{
    printf("pocl baseline unavailable: %s\n", why);
    return 0;
}

/* checkErr of the reference (main.cpp:41-46): print and give up */
#define CL_TRY(call, what)                                                   \
    do {                                                                     \
        cl_int e_ = (call);                                                  \
        if (e_ != CL_SUCCESS) {                                              \
            fprintf(stderr, "ERROR: %s (%d)\n", what, (int)e_);              \
            return 1;                                                        \
        }                                                                    \
    } while (0)

static char *read_file(const char *path, size_t *len)
{
    FILE *f = fopen(path, "rb");
    if (!f) return NULL;
    fseek(f, 0, SEEK_END);
    long n = ftell(f);
    fseek(f, 0, SEEK_SET);
    char *buf = (char *)malloc((size_t)n + 1);
    if (buf && fread(buf, 1, (size_t)n, f) != (size_t)n) { free(buf); buf = NULL; }
    fclose(f);
    if (buf) { buf[n] = 0; *len = (size_t)n; }
    return buf;
}

/* "#define NAME <number>" at the start of a line -> "#define NAME <value>"; returns a new string or NULL if absent */
static char *redefine(const char *src, const char *name, int value)
{
    char key[64];
    snprintf(key, sizeof key, "#define %s ", name);
    const char *p = src;
    while ((p = strstr(p, key)) != NULL) {
        if (p == src || p[-1] == '\n') break;
        p += strlen(key);
    }
    if (!p) return NULL;
    const char *eol = strchr(p, '\n');
    if (!eol) eol = p + strlen(p);
    size_t head = (size_t)(p - src), n = strlen(src);
    char *out = (char *)malloc(n + 64);
    if (!out) return NULL;
    memcpy(out, src, head);
    int w = snprintf(out + head, 64, "#define %s %d", name, value);
    strcpy(out + head + w, eol);
    return out;
}

typedef struct {
    int32_t width, height, n_spheres, n_planes, n_triangles, n_surfaces;
    unsigned char *spheres, *planes, *triangles, *surfaces, *camera;
    uint32_t *seeds;
    unsigned char *blob;
} Input;

static int load_input(const char *path, Input *in)
{
    size_t len = 0;
    in->blob = (unsigned char *)read_file(path, &len);
    if (!in->blob || len < 32) return -1;
    const int32_t *h = (const int32_t *)in->blob;
    if (h[0] != MAGIC) return -1;
    in->width = h[1]; in->height = h[2]; in->n_spheres = h[3]; in->n_planes = h[4]; in->n_triangles = h[5]; in->n_surfaces = h[6];
    if (in->width <= 0 || in->height <= 0 || in->n_spheres < 0 || in->n_planes < 0 || in->n_triangles < 0 || in->n_surfaces <= 0) return -1;
    size_t need = 32 + (size_t)in->n_spheres * 32 + (size_t)in->n_planes * 48 + (size_t)in->n_triangles * 64 + (size_t)in->n_surfaces * 32 + 48 +
                  (size_t)in->width * in->height * 16;
    if (len != need) return -1;
    unsigned char *p = in->blob + 32;
    in->spheres = p; p += (size_t)in->n_spheres * 32;
    in->planes = p; p += (size_t)in->n_planes * 48;
    in->triangles = p; p += (size_t)in->n_triangles * 64;
    in->surfaces = p; p += (size_t)in->n_surfaces * 32;
    in->camera = p; p += 48;
    in->seeds = (uint32_t *)p;
    return 0;
}

static cl_mem make_buffer(cl_context ctx, cl_mem_flags flags, size_t bytes, cl_int *err)
{
    return clCreateBuffer(ctx, flags, bytes ? bytes : 16, NULL, err);   /* the reference's Plane buffer has zero elements (main.cpp:271, :427) */
}

/* one timed run of `frames` launches with the program built with `options`; image_out may be NULL */
static int run(cl_context ctx, cl_device_id dev, const char *source, const char *options, const char *label, const Input *in, int frames,
               int warmup, float *image_out, const char *dev_name, unsigned units, int patched_semicolon)
{
    cl_int err = CL_SUCCESS;
    const size_t W = (size_t)in->width, H = (size_t)in->height;
    cl_program prog = clCreateProgramWithSource(ctx, 1, &source, NULL, &err);
    CL_TRY(err, "clCreateProgramWithSource");
    err = clBuildProgram(prog, 1, &dev, options, NULL, NULL);
    if (err != CL_SUCCESS) {
        size_t n = 0;
        clGetProgramBuildInfo(prog, dev, CL_PROGRAM_BUILD_LOG, 0, NULL, &n);
        char *log = (char *)malloc(n + 1);
        if (log) { clGetProgramBuildInfo(prog, dev, CL_PROGRAM_BUILD_LOG, n, log, NULL); log[n] = 0; fprintf(stderr, "Build log:%s\n", log); free(log); }
        clReleaseProgram(prog);
        return 2;                                                         /* the caller may retry with the one-character patch */
    }
    cl_kernel k = clCreateKernel(prog, "bi_directional_path_trace", &err);
    CL_TRY(err, "clCreateKernel");
    cl_command_queue q = clCreateCommandQueue(ctx, dev, 0, &err);          /* in order, no profiling (main.cpp:522) */
    CL_TRY(err, "clCreateCommandQueue");

    cl_mem b_sph = make_buffer(ctx, CL_MEM_READ_ONLY, (size_t)in->n_spheres * 32, &err);   CL_TRY(err, "clCreateBuffer spheres");
    cl_mem b_pln = make_buffer(ctx, CL_MEM_READ_ONLY, (size_t)in->n_planes * 48, &err);    CL_TRY(err, "clCreateBuffer planes");
    cl_mem b_tri = make_buffer(ctx, CL_MEM_READ_ONLY, (size_t)in->n_triangles * 64, &err); CL_TRY(err, "clCreateBuffer triangles");
    cl_mem b_cam = make_buffer(ctx, CL_MEM_READ_ONLY, 48, &err);                           CL_TRY(err, "clCreateBuffer camera");
    cl_mem b_geo = make_buffer(ctx, CL_MEM_READ_ONLY, 12, &err);                           CL_TRY(err, "clCreateBuffer geometry");
    cl_mem b_img = make_buffer(ctx, CL_MEM_READ_ONLY, 16, &err);                           CL_TRY(err, "clCreateBuffer image descriptor");
    cl_mem b_seed = make_buffer(ctx, CL_MEM_READ_WRITE, W * H * 16, &err);                 CL_TRY(err, "clCreateBuffer seeds");
    cl_mem b_surf = make_buffer(ctx, CL_MEM_READ_WRITE, (size_t)in->n_surfaces * 32, &err); CL_TRY(err, "clCreateBuffer surfaces");
    cl_image_format fmt = {CL_RGBA, CL_FLOAT};
    cl_image_desc desc;
    memset(&desc, 0, sizeof desc);
    desc.image_type = CL_MEM_OBJECT_IMAGE2D;
    desc.image_width = W;
    desc.image_height = H;
    /* read-write for the host's sake: each image is the kernel's output in one frame and its input in the next */
    cl_mem img[2];
    img[0] = clCreateImage(ctx, CL_MEM_READ_WRITE, &fmt, &desc, NULL, &err); CL_TRY(err, "clCreateImage");
    img[1] = clCreateImage(ctx, CL_MEM_READ_WRITE, &fmt, &desc, NULL, &err); CL_TRY(err, "clCreateImage");

    const int32_t geometry[3] = {in->n_spheres, in->n_planes, in->n_triangles};           /* primitives.h GeometryDescriptor */
    int32_t img_desc[4] = {0, 0, 1, 0};                                                  /* renderer.h:3-9: numSamples, bChanged, sampleRate, reserved1 */
    CL_TRY(clEnqueueWriteBuffer(q, b_cam, CL_TRUE, 0, 48, in->camera, 0, NULL, NULL), "write camera");
    if (in->n_spheres) CL_TRY(clEnqueueWriteBuffer(q, b_sph, CL_TRUE, 0, (size_t)in->n_spheres * 32, in->spheres, 0, NULL, NULL), "write spheres");
    if (in->n_planes) CL_TRY(clEnqueueWriteBuffer(q, b_pln, CL_TRUE, 0, (size_t)in->n_planes * 48, in->planes, 0, NULL, NULL), "write planes");
    if (in->n_triangles) CL_TRY(clEnqueueWriteBuffer(q, b_tri, CL_TRUE, 0, (size_t)in->n_triangles * 64, in->triangles, 0, NULL, NULL), "write triangles");
    CL_TRY(clEnqueueWriteBuffer(q, b_geo, CL_TRUE, 0, 12, geometry, 0, NULL, NULL), "write geometry");
    CL_TRY(clEnqueueWriteBuffer(q, b_surf, CL_TRUE, 0, (size_t)in->n_surfaces * 32, in->surfaces, 0, NULL, NULL), "write surfaces");

    CL_TRY(clSetKernelArg(k, 0, sizeof(cl_mem), &b_cam), "setArg 0");
    CL_TRY(clSetKernelArg(k, 3, sizeof(cl_mem), &b_sph), "setArg 3");
    CL_TRY(clSetKernelArg(k, 4, sizeof(cl_mem), &b_pln), "setArg 4");
    CL_TRY(clSetKernelArg(k, 5, sizeof(cl_mem), &b_tri), "setArg 5");
    CL_TRY(clSetKernelArg(k, 6, sizeof(cl_mem), &b_geo), "setArg 6");
    CL_TRY(clSetKernelArg(k, 7, sizeof(cl_mem), &b_img), "setArg 7");
    CL_TRY(clSetKernelArg(k, 8, sizeof(cl_mem), &b_seed), "setArg 8");
    CL_TRY(clSetKernelArg(k, 9, sizeof(cl_mem), &b_surf), "setArg 9");

    const size_t global[2] = {W, H};
    double seconds = 0.0;
    int out = 0;
    for (int phase = 0; phase < 2; phase++) {                             /* warm-up launches, then the run proper from a fresh state */
        const int n = phase == 0 ? warmup : frames;
        CL_TRY(clEnqueueWriteBuffer(q, b_seed, CL_TRUE, 0, W * H * 16, in->seeds, 0, NULL, NULL), "write seeds");
        img_desc[0] = 0;
        const double t0 = now_s();
        for (int f = 0; f < n; f++) {
            CL_TRY(clEnqueueWriteBuffer(q, b_img, CL_TRUE, 0, 16, img_desc, 0, NULL, NULL), "write image descriptor");   /* main.cpp:108 */
            CL_TRY(clSetKernelArg(k, 1, sizeof(cl_mem), &img[out]), "setArg 1");
            CL_TRY(clSetKernelArg(k, 2, sizeof(cl_mem), &img[out ^ 1]), "setArg 2");
            CL_TRY(clEnqueueNDRangeKernel(q, k, 2, NULL, global, NULL, 0, NULL, NULL), "clEnqueueNDRangeKernel");     /* main.cpp:116 */
            CL_TRY(clFinish(q), "clFinish");                                                                           /* main.cpp:121-124 */
            img_desc[0] += 1;                                                                                          /* main.cpp:131 */
            out ^= 1;                                                                                                  /* main.cpp:133-138 */
        }
        if (phase == 1) seconds = now_s() - t0;
    }
    if (image_out && frames > 0) {
        const size_t origin[3] = {0, 0, 0}, region[3] = {W, H, 1};
        CL_TRY(clEnqueueReadImage(q, img[out ^ 1], CL_TRUE, origin, region, 0, 0, image_out, 0, NULL, NULL), "clEnqueueReadImage");
    }
    printf("{\"what\": \"kernel file on an OpenCL device\", \"device\": \"%s\", \"compute_units\": %u, \"nproc\": %ld, \"build_options\": \"%s\", "
           "\"width\": %d, \"height\": %d, \"frames\": %d, \"seconds\": %.6f, \"mpaths_per_s\": %.6f, \"patched_missing_semicolon\": %s}\n",
           dev_name, units, sysconf(_SC_NPROCESSORS_ONLN), label, in->width, in->height, frames, seconds,
           seconds > 0.0 ? (double)W * (double)H * frames / seconds / 1e6 : 0.0, patched_semicolon ? "true" : "false");
    fflush(stdout);
    clReleaseMemObject(b_sph); clReleaseMemObject(b_pln); clReleaseMemObject(b_tri); clReleaseMemObject(b_cam); clReleaseMemObject(b_geo);
    clReleaseMemObject(b_img); clReleaseMemObject(b_seed); clReleaseMemObject(b_surf); clReleaseMemObject(img[0]); clReleaseMemObject(img[1]);
    clReleaseCommandQueue(q); clReleaseKernel(k); clReleaseProgram(prog);
    return 0;
}

int main(int argc, char **argv)
{
    const char *kernel_path = getenv("BDPT_REFERENCE_KERNEL"), *input = NULL, *out_path = NULL;
    if (!kernel_path || !*kernel_path) kernel_path = "/root/reference/src/kernel/kernel0.cl";
    int frames = 64, warmup = 2;
    cl_device_type dev_type = CL_DEVICE_TYPE_CPU;
    for (int i = 1; i < argc; i++) {
        if (!strcmp(argv[i], "--device-type") && i + 1 < argc) {
            const char *t = argv[++i];
            if (!strcmp(t, "gpu")) dev_type = CL_DEVICE_TYPE_GPU;
            else if (strcmp(t, "cpu")) { fprintf(stderr, "cl_baseline: --device-type cpu or gpu\n"); return 1; }
            continue;
        }
        if (!strcmp(argv[i], "--kernel") && i + 1 < argc) kernel_path = argv[++i];
        else if (!strcmp(argv[i], "--input") && i + 1 < argc) input = argv[++i];
        else if (!strcmp(argv[i], "--frames") && i + 1 < argc) frames = atoi(argv[++i]);
        else if (!strcmp(argv[i], "--warmup") && i + 1 < argc) warmup = atoi(argv[++i]);
        else if (!strcmp(argv[i], "--out") && i + 1 < argc) out_path = argv[++i];
        else { fprintf(stderr, "usage: cl_baseline --input scene.bin [--kernel kernel0.cl] [--frames N] [--warmup N] [--out image.bin] [--device-type cpu|gpu]\n"); return 1; }
    }
    if (!input || frames < 1 || warmup < 0) { fprintf(stderr, "cl_baseline: --input is required, --frames >= 1, --warmup >= 0\n"); return 1; }
    Input in;
    memset(&in, 0, sizeof in);
    if (load_input(input, &in) != 0) { fprintf(stderr, "cl_baseline: %s is not a scene file of this tool\n", input); return 1; }

    /* a CPU device on any platform (main.cpp:227-255 takes the first platform's) */
    cl_uint n_plat = 0;
    if (clGetPlatformIDs(0, NULL, &n_plat) != CL_SUCCESS || n_plat == 0) return unavailable("no OpenCL platform on this host");
    cl_platform_id plats[16];
    if (n_plat > 16) n_plat = 16;
    clGetPlatformIDs(n_plat, plats, NULL);
    cl_device_id dev = NULL;
    cl_platform_id plat = NULL;
    for (cl_uint p = 0; p < n_plat && !dev; p++) {
        cl_uint n = 0;
        if (clGetDeviceIDs(plats[p], dev_type, 1, &dev, &n) == CL_SUCCESS && n > 0) plat = plats[p];
        else dev = NULL;
    }
    if (!dev) {
        char why[128];
        snprintf(why, sizeof why, "%u OpenCL platform(s) on this host, none with a %s device (CL_DEVICE_TYPE_%s)", (unsigned)n_plat,
                 dev_type == CL_DEVICE_TYPE_CPU ? "CPU" : "GPU", dev_type == CL_DEVICE_TYPE_CPU ? "CPU" : "GPU");
        return unavailable(why);
    }
    size_t src_len = 0;
    char *src = read_file(kernel_path, &src_len);
    if (!src) {
        char why[512];
        snprintf(why, sizeof why, "an OpenCL device exists but the reference's kernel file %s cannot be read (it is not part of this repository; pass --kernel or set BDPT_REFERENCE_KERNEL)", kernel_path);
        return unavailable(why);
    }
    char *w = redefine(src, "WIDTH", in.width);
    char *wh = w ? redefine(w, "HEIGHT", in.height) : NULL;
    if (!wh) { fprintf(stderr, "cl_baseline: %s has no '#define WIDTH' / '#define HEIGHT' lines to set\n", kernel_path); return 1; }

    char dev_name[256] = "?";
    cl_uint units = 0;
    clGetDeviceInfo(dev, CL_DEVICE_NAME, sizeof dev_name, dev_name, NULL);
    clGetDeviceInfo(dev, CL_DEVICE_MAX_COMPUTE_UNITS, sizeof units, &units, NULL);
    cl_bool images = CL_FALSE;
    clGetDeviceInfo(dev, CL_DEVICE_IMAGE_SUPPORT, sizeof images, &images, NULL);
    if (!images) return unavailable("the OpenCL device has no image support (the kernel reads and writes image2d_t)");
    cl_int err = CL_SUCCESS;
    cl_context_properties props[3] = {CL_CONTEXT_PLATFORM, (cl_context_properties)plat, 0};
    cl_context ctx = clCreateContext(props, 1, &dev, NULL, NULL, &err);
    CL_TRY(err, "clCreateContext");

    float *image = out_path ? (float *)malloc((size_t)in.width * in.height * 16) : NULL;
    const char *source = wh;
    char *fixed = NULL;
    int patched = 0;
    /* the reference's build (main.cpp:477), then the compiler's defaults */
    int rc = run(ctx, dev, source, "-cl-opt-disable", "-cl-opt-disable", &in, frames, warmup, image, dev_name, (unsigned)units, 0);
    if (rc == 2) {
        /* kernel0.cl:69 lacks a ';' after the last member of ImageDescriptor; some compilers refuse it */
        const char *at = strstr(wh, "int sampleRate\n");
        if (at) {
            fixed = (char *)malloc(strlen(wh) + 2);
            size_t head = (size_t)(at - wh) + strlen("int sampleRate");
            memcpy(fixed, wh, head);
            fixed[head] = ';';
            strcpy(fixed + head + 1, wh + head);
            source = fixed;
            patched = 1;
            rc = run(ctx, dev, source, "-cl-opt-disable", "-cl-opt-disable", &in, frames, warmup, image, dev_name, (unsigned)units, 1);
        }
    }
    if (rc != 0) { fprintf(stderr, "cl_baseline: the kernel did not build or run on %s\n", dev_name); return 1; }
    if (image && out_path) {
        FILE *f = fopen(out_path, "wb");
        if (!f || fwrite(image, 16, (size_t)in.width * in.height, f) != (size_t)in.width * in.height) { fprintf(stderr, "cl_baseline: cannot write %s\n", out_path); return 1; }
        fclose(f);
    }
    rc = run(ctx, dev, source, "", "default", &in, frames, warmup, NULL, dev_name, (unsigned)units, patched);
    clReleaseContext(ctx);
    free(src); free(w); free(wh); free(fixed); free(image); free(in.blob);
    return rc == 0 ? 0 : 1;
}

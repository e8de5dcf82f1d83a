/* A stand-in with the signature and the access pattern of the kernel that tools/cl_baseline drives -- ten arguments,
 * a WIDTH x HEIGHT NDRange, a read-only and a write-only image, a per-pixel generator word at index x*HEIGHT + y --
 * and arithmetic simple enough to predict on the host.  It exists to exercise every OpenCL call of the harness on a
 * real device; it traces nothing. */
#define WIDTH 1067
#define HEIGHT 600

__constant sampler_t probe_sampler = CLK_NORMALIZED_COORDS_FALSE | CLK_ADDRESS_CLAMP_TO_EDGE | CLK_FILTER_NEAREST;

typedef struct { float4 pos, look, up; } ProbeCamera;
typedef struct { float4 c; float r; uint surf; } ProbeSphere;
typedef struct { float4 a, b, c; uint surf; } ProbeTriangle;
typedef struct { float4 n, p; uint surf; } ProbePlane;
typedef struct { int n_spheres, n_planes, n_triangles; } ProbeGeometry;
typedef struct { int n_samples, changed, rate; } ProbeImage;
typedef struct { float4 colour, emissive; } ProbeSurface;

__kernel void bi_directional_path_trace(__global ProbeCamera *cam, __write_only image2d_t img_out, __read_only image2d_t img_in,
                                        __global ProbeSphere *spheres, __global ProbePlane *planes, __global ProbeTriangle *triangles,
                                        __global ProbeGeometry *geom, __global ProbeImage *desc, __global uint4 *seeds,
                                        __global ProbeSurface *surfaces)
{
    const int x = get_global_id(0), y = get_global_id(1);
    const float4 prev = read_imagef(img_in, probe_sampler, (int2)(x, y));
    const int n = desc[0].n_samples;
    uint4 state = seeds[x * HEIGHT + y];
    state.w *= 48271u;
    /* one number from every argument, exactly representable sums */
    const float from_scene = spheres[geom[0].n_spheres - 1].r * 4.0f + triangles[geom[0].n_triangles - 1].a.x + surfaces[2].emissive.x +
                             (float)geom[0].n_planes;
    const float4 value = (float4)((float)(x + 2 * y), (float)(state.w >> 24), from_scene, cam[0].pos.x);
    const float4 mean = n == 0 ? value : prev * ((float)n / (float)(n + 1)) + value / (float)(n + 1);
    seeds[x * HEIGHT + y] = state;
    write_imagef(img_out, (int2)(x, y), mean);
}

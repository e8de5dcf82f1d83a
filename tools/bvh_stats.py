"""How well the lanes of a wavefront keep together in the hierarchy walk of the many-triangle kernel (BDPT_KERNEL_BVH),
from a build with -DBDPT_BVH_STATS:  tools/ab_build.sh stats -DBDPT_BVH_STATS;  BDPT_LIB_DIR=build/ab/stats python tools/bvh_stats.py [triangles] [spp]"""
import sys, os, ctypes
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "opencl-path-tracer_amd"))
import bdpt
W, H = 1920, 1080
n, spp = (int(sys.argv[1]) if len(sys.argv) > 1 else 100000), (int(sys.argv[2]) if len(sys.argv) > 2 else 2)
sc = bdpt.Scene.soup(n)
with bdpt.Context(W, H) as ctx:
    ctx.set_params(bdpt.default_params(kernel_variant=bdpt.KERNEL_BVH, rng_mode=bdpt.RNG_STRIDED, accum_mode=bdpt.ACCUM_SUM))
    ctx.upload_scene(sc); ctx.upload_seeds(bdpt.make_seeds(W, H, 0x5EED)); ctx.set_camera(sc.camera)
    ctx.render(spp)
    ms = ctx.last_launch_ms()
    out = (ctypes.c_ulonglong * 8)()
    bdpt.lib.bdpt_bvh_stats_read(out)
    A, B, C, D, F, G = [int(x) for x in out[:6]]
    print("%d triangles, %d spp: %.1f ms (instrumented build)" % (n, spp, ms))
    print("searches: %d of lanes in %d of wavefronts (%.1f lanes per search)" % (F, G, F / G))
    print("node steps: %.1f per lane, %.1f trips per wavefront: %.3f of 64 lanes busy" % (A / F, B / G, A / (64.0 * B)))
    print("leaves: %.1f per lane, %.1f trips per wavefront: %.3f of 64 lanes busy" % (C / F, D / G, C / (64.0 * D)))

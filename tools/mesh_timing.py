"""Launch times of the many-triangle kernels on a connected surface (tests/bvh_scenes.py: a grid of quads over the floor of the
Cornell box, heights random) at 1080p, 4 samples: hierarchy against exhaustive search, images compared.
usage: mesh_timing.py [cells per side: 2 x cells^2 triangles]"""
import sys, os, time
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "opencl-path-tracer_amd")); sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "tests"))
import numpy as np, bdpt, bvh_scenes
W, H = 1920, 1080
cells = int(sys.argv[1]) if len(sys.argv) > 1 else 300
t0 = time.time(); sc = bvh_scenes.mesh(bdpt, np.random.default_rng(1), cells); print("mesh of", len(sc.triangles), "triangles built in %.1f s" % (time.time() - t0), flush=True)
res = {}
for v in [6, 4] if len(sc.triangles) < 300000 else [6]:
    with bdpt.Context(W, H) as ctx:
        ctx.set_params(bdpt.default_params(kernel_variant=v, rng_mode=1, accum_mode=1))
        t0 = time.time(); ctx.upload_scene(sc); up = time.time() - t0
        ctx.upload_seeds(bdpt.make_seeds(W, H, 0x5EED)); ctx.set_camera(sc.camera)
        ctx.render(4); res[v] = ctx.read_accum()
        print("variant", v, ctx.hierarchy_info(), "upload %.2f s" % up, "4 spp: %.1f ms" % ctx.last_launch_ms(), flush=True)
if 4 in res: print("images", "same" if np.array_equal(res[6].view(np.uint32), res[4].view(np.uint32)) else "DIFFER")

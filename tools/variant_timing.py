"""Launch time of every kernel variant on Cornell + N soup triangles (1080p).  usage: variant_timing.py N [spp]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "opencl-path-tracer_amd"))
import bdpt
N = int(sys.argv[1]); SPP = int(sys.argv[2]) if len(sys.argv) > 2 else 16
W, H = 1920, 1080
scene = bdpt.Scene.soup(N)
for variant, name in ((1, "direct"), (3, "lds"), (4, "tiled")):
    ctx = bdpt.Context(W, H)
    try:
        ctx.set_params(bdpt.default_params(rng_mode=bdpt.RNG_STRIDED, accum_mode=bdpt.ACCUM_SUM, kernel_variant=variant))
        ctx.upload_scene(scene); ctx.upload_seeds(bdpt.make_seeds(W, H, 0x5EED)); ctx.set_camera(scene.camera)
        ctx.render(SPP)
        print("soup %d, %d spp, %-6s %.1f ms" % (N, SPP, name, ctx.last_launch_ms()), flush=True)
    except bdpt.BdptError as e:
        print("soup %d %s: %s" % (N, name, e))
    ctx.close()

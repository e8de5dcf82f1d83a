"""Times one launch of the 100k-triangle soup scene at 1080p (tiled kernel).  BDPT_LIB_DIR selects the build."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "opencl-path-tracer_amd"))
import bdpt
W, H, SPP = 1920, 1080, int(sys.argv[1]) if len(sys.argv) > 1 else 2
scene = bdpt.Scene.soup(100000)
ctx = bdpt.Context(W, H)
ctx.set_params(bdpt.default_params(rng_mode=bdpt.RNG_STRIDED, accum_mode=bdpt.ACCUM_SUM))
ctx.upload_scene(scene); ctx.upload_seeds(bdpt.make_seeds(W, H, 0x5EED)); ctx.set_camera(scene.camera)
ctx.render(SPP)
print(os.environ.get("BDPT_LIB_DIR", "current"), "%d spp: %.1f ms" % (SPP, ctx.last_launch_ms()), flush=True)
ctx.close()

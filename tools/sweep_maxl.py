"""Launch time of the Cornell benchmark frame as a function of max_light_bounces (per-iteration cost of the light loop)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "opencl-path-tracer_amd"))
import bdpt
W, H, SPP = 1920, 1080, 256
scene = bdpt.Scene.cornell()
for mode in ("stream", "strided"):
    for maxl in (1, 2, 3, 4, 5):
        ctx = bdpt.Context(W, H)
        kw = dict(max_light_bounces=maxl)
        if mode == "strided":
            kw.update(rng_mode=bdpt.RNG_STRIDED, accum_mode=bdpt.ACCUM_SUM)
        ctx.set_params(bdpt.default_params(**kw))
        ctx.upload_scene(scene); ctx.upload_seeds(bdpt.make_seeds(W, H)); ctx.set_camera(scene.camera)
        ctx.render(8); ctx.num_samples = 0; ctx.render(SPP)
        print(mode, "max_l", maxl, "%.2f ms" % ctx.last_launch_ms(), flush=True)
        ctx.close()

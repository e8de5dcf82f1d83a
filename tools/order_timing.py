"""Launch time of the Cornell benchmark frame for the three sample orders: STREAM, STRIDED by index, STRIDED by roulette class."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "opencl-path-tracer_amd"))
import bdpt
W, H = 1920, 1080
scene = bdpt.Scene.cornell()
for spp in (256, 1024):
    for name, kw in (("stream/mean", {}),
                     ("strided/mean (index order)", dict(rng_mode=bdpt.RNG_STRIDED)),
                     ("strided/sum (class order)", dict(rng_mode=bdpt.RNG_STRIDED, accum_mode=bdpt.ACCUM_SUM, sample_order=bdpt.ORDER_ROULETTE_CLASS)),
                     ("strided/sum (adaptive order)", dict(rng_mode=bdpt.RNG_STRIDED, accum_mode=bdpt.ACCUM_SUM))):
        ctx = bdpt.Context(W, H)
        ctx.set_params(bdpt.default_params(**kw))
        ctx.upload_scene(scene); ctx.upload_seeds(bdpt.make_seeds(W, H, 0x5EED)); ctx.set_camera(scene.camera)
        ctx.render(8); ctx.num_samples = 0; ctx.render(spp)
        print(spp, name, "%.2f ms" % ctx.last_launch_ms(), flush=True)
        ctx.close()

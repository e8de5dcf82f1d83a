"""Small driver for profiler runs: one warm-up launch and N timed launches of the
bench workload (Cornell, 1920x1080 unless overridden) through the C-ABI, no torch.

    python3 tools/prof_run.py [--spp 128] [--launches 2] [--scene cornell|soup:N|outdoor|planes]
                              [--width W --height H] [--mode strided|stream] [--kernel K] [--chunks C] [--paths 0..3]
"""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "opencl-path-tracer_amd"))
import bdpt  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--spp", type=int, default=128)
ap.add_argument("--launches", type=int, default=2)
ap.add_argument("--scene", default="cornell")
ap.add_argument("--width", type=int, default=1920)
ap.add_argument("--height", type=int, default=1080)
ap.add_argument("--mode", default="strided")
ap.add_argument("--kernel", type=int, default=0)
ap.add_argument("--chunks", type=int, default=0, help="bdpt_set_launch_chunks (0 = automatic, 1 = unsliced)")
ap.add_argument("--paths", type=int, default=0, help="bdpt_set_path_scheduling (0 = the library chooses, 1 = one lane per pixel, 2 = paths pooled by phase, 3 = pixels sorted by the class of their next sample)")
ap.add_argument("--order-scratch", type=int, default=-1, help="bdpt_set_order_scratch_limit in MiB (-1 = the default)")
a = ap.parse_args()

if a.scene.startswith("soup:"):
    scene = bdpt.Scene.soup(int(a.scene.split(":")[1]))
else:
    scene = {"cornell": bdpt.Scene.cornell, "outdoor": bdpt.Scene.outdoor, "planes": bdpt.Scene.cornell_planes}[a.scene]()
ctx = bdpt.Context(a.width, a.height)
if a.mode == "strided":
    ctx.set_params(bdpt.default_params(rng_mode=bdpt.RNG_STRIDED, accum_mode=bdpt.ACCUM_SUM, kernel_variant=a.kernel))
else:
    ctx.set_params(bdpt.default_params(kernel_variant=a.kernel))
ctx.upload_scene(scene)
ctx.upload_seeds(bdpt.make_seeds(a.width, a.height, 0x5EED))
ctx.set_camera(scene.camera)
ctx.set_launch_chunks(a.chunks)
ctx.set_path_scheduling(a.paths)
if a.order_scratch >= 0:
    ctx.set_order_scratch_limit(a.order_scratch << 20)
ctx.render(min(a.spp, 8))
ms = []
for _ in range(a.launches):
    ctx.num_samples = 0
    ctx.render(a.spp)
    ms.append(ctx.last_launch_ms())
paths = a.width * a.height * a.spp
print("scene=%s %dx%d spp=%d kernel=%d mode=%s paths=%d: %s ms -> %.1f Mpaths/s" % (a.scene, a.width, a.height, a.spp, a.kernel, a.mode,
      ctx.last_launch_path_scheduling(), ["%.3f" % m for m in ms], paths / (min(ms) / 1e3) / 1e6))
ctx.close()

"""Where the two-stage kernel's wavefronts spend their time.  Needs the instrumented build:

    tools/ab_build.sh clocks -DBDPT_PHASE_CLOCKS
    BDPT_LIB_DIR=build/ab/clocks python3 tools/phase_clocks.py [--spp 64] [--mode strided|stream]

Prints, for one launch of the Cornell box at 1920x1080, the share of wavefront clock ticks spent in each part of the
closest-hit search and of the sample around it.  A wavefront's ticks include the time
it waits for the SIMD's other wavefronts, so shares are shares of residency, which is what a SIMD's issue slots are
divided by.
"""
import argparse
import ctypes
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "opencl-path-tracer_amd"))
import bdpt  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--spp", type=int, default=64)
ap.add_argument("--mode", default="strided")
ap.add_argument("--width", type=int, default=1920)
ap.add_argument("--height", type=int, default=1080)
a = ap.parse_args()

lib = ctypes.CDLL(os.path.join(os.environ.get("BDPT_LIB_DIR", ""), "libbdpt_amd.so"))
if not hasattr(lib, "bdpt_phase_clocks"):
    sys.exit("this libbdpt_amd.so was not built with -DBDPT_PHASE_CLOCKS")
scene = bdpt.Scene.cornell()
ctx = bdpt.Context(a.width, a.height)
kw = dict(kernel_variant=5)
if a.mode == "strided":
    kw.update(rng_mode=bdpt.RNG_STRIDED, accum_mode=bdpt.ACCUM_SUM)
ctx.set_params(bdpt.default_params(**kw))
ctx.upload_scene(scene)
ctx.upload_seeds(bdpt.make_seeds(a.width, a.height, 0x5EED))
ctx.set_camera(scene.camera)
ctx.set_launch_chunks(1)
ctx.render(4)
out = (ctypes.c_ulonglong * 12)()
lib.bdpt_phase_clocks(out, 1)
ctx.num_samples = 0
ctx.render(a.spp)
lib.bdpt_phase_clocks(out, 1)
total = float(out[0])
names = ["whole pixel", "search: spheres and planes", "search: stage 1", "search: stage 2", "search: winner's record, normal",
         "camera ray, eye vertex", "point on the light", "connection: direction, term", "roulette, bounce direction",
         "bounce bookkeeping, sample's colour", "next sample (order), running sum", "before the first sample"]
print("launch %.2f ms (instrumented)" % ctx.last_launch_ms())
for i in range(1, 12):
    print("%-36s %5.1f %%" % (names[i], 100.0 * out[i] / total))
print("%-36s %5.1f %%" % ("the rest", 100.0 * (total - sum(out[1:12])) / total))
ctx.close()

"""Balance of an image-tile split of the 1080p Cornell frame over N ranks, measured on one GPU: each share is
rendered alone and its kernel time taken (HIP events), for contiguous row bands and for tiles dealt round-robin.
The frame is not uniform -- eye rays left and right of the box end their path after one ray
(/root/reference/src/kernel/kernel0.cl:645-656) -- so the question is which split gives every rank the same mix.

    python tools/tile_balance.py [--world 8 --spp 16] > profiles/rNN_tile_balance.json
"""
import argparse
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "opencl-path-tracer_amd"))

import numpy as np  # noqa: E402

import bdpt  # noqa: E402
import dist_render as dr  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--width", type=int, default=1920)
ap.add_argument("--height", type=int, default=1080)
ap.add_argument("--world", type=int, default=8)
ap.add_argument("--spp", type=int, default=16)
ap.add_argument("--reps", type=int, default=3)
a = ap.parse_args()
W, H = a.width, a.height
scene = bdpt.Scene.cornell()


def timed(setup):
    """Median kernel time of `setup(ctx)`-configured launches of a.spp samples from a fresh counter."""
    with bdpt.Context(W, H) as ctx:
        ctx.upload_scene(scene)
        ctx.upload_seeds(bdpt.make_seeds(W, H, 0x5EED))
        ctx.set_camera(scene.camera)
        if not setup(ctx):
            return 0.0
        ms = []
        for _ in range(a.reps + 1):
            ctx.num_samples = 0
            ctx.render(a.spp)
            ms.append(ctx.last_launch_ms())
        return float(np.median(ms[1:]))


def band(rank):
    def setup(ctx):
        y0, h = dr.row_band(H, a.world, rank)
        if h > 0:
            ctx.set_region(0, y0, W, h)
        return h > 0
    return setup


def tiles(rank, tile):
    def setup(ctx):
        ctx.set_tile_interleave(tile[0], tile[1], a.world, rank)
        return True
    return setup


out = {"frame": "%dx%d Cornell box, %d spp per launch, STREAM + MEAN" % (W, H, a.spp), "ranks": a.world,
       "whole_frame_ms": round(timed(lambda ctx: True), 3), "splits": {}}
cases = {"row bands": [band(r) for r in range(a.world)]}
for tile in ((64, 32), (128, 64), (32, 16), (16, 8)):
    cases["tiles %dx%d round-robin" % tile] = [tiles(r, tile) for r in range(a.world)]
for name, setups in cases.items():
    ms = [timed(s) for s in setups]
    out["splits"][name] = {"share_ms": [round(m, 3) for m in ms], "max_ms": round(max(ms), 3), "mean_ms": round(float(np.mean(ms)), 3),
                           "max_over_mean": round(max(ms) / float(np.mean(ms)), 3),
                           "speedup_vs_whole_frame": round(out["whole_frame_ms"] / max(ms), 2)}
print(json.dumps(out, indent=1))

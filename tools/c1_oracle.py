"""Configuration C1 (plumbing, no GPU): Cornell box, 512x512, 64 launches of one sample each with the sample counter
0..63 (the reference's frame loop), rendered by the CPU oracle on the host's cores.  Writes the image and prints one
JSON line with the wall time.

    python tools/c1_oracle.py [--out c1.pfm] [--threads N] [--size 512] [--frames 64]
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "opencl-path-tracer_amd"))
sys.path.insert(0, os.path.join(ROOT, "oracle"))

ap = argparse.ArgumentParser()
ap.add_argument("--out", default="")
ap.add_argument("--threads", type=int, default=min(os.cpu_count() or 1, int(os.environ.get("OMP_NUM_THREADS", "16"))))
ap.add_argument("--size", type=int, default=512)
ap.add_argument("--frames", type=int, default=64)
a = ap.parse_args()

import bdpt  # noqa: E402  (scene builders and seeds only: nothing here touches a GPU)
import oracle_bind as oracle  # noqa: E402

W = H = a.size
scene = bdpt.Scene.cornell()
st = oracle.State(W, H, bdpt.make_seeds(W, H, 0x5EED))
p = bdpt.default_params()
t0 = time.perf_counter()
for _ in range(a.frames):                      # one sample per launch, numSamples 0..frames-1 (main.cpp:108-131)
    oracle.render(st, scene, scene.camera, p, 1, n_threads=a.threads)
sec = time.perf_counter() - t0
if a.out:
    bdpt.write_image(a.out, st.accum)
print(json.dumps({"config": "C1: Cornell box %dx%d, %d frames x 1 spp, CPU oracle" % (W, H, a.frames), "threads": a.threads,
                  "seconds": round(sec, 3), "mpaths_per_s": round(W * H * a.frames / sec / 1e6, 3),
                  "mean_rgb": [round(float(v), 6) for v in st.accum[..., :3].mean(axis=(0, 1))]}))

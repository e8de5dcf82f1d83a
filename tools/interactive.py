"""Interactive camera-move mode: 1080p progressive frames, 1 sample per pixel per frame, accumulation reset
while the camera moves, image tiles dealt round-robin to the ranks and gathered on rank 0 every frame.

    python tools/interactive.py [--frames-moving 60 --frames-still 60]
    python -m torch.distributed.run --nproc-per-node N tools/interactive.py ... [--backend gloo]

Script: the reference's ModelCamera (camera.cpp:67-133) with the left button held and the cursor dragged
so that the camera orbits 0.5 degrees per frame about its up vector, then released.  The frame loop is the
reference's (main.cpp:100-156): render, then handle input, and while the button is held zero the counter and
upload the new camera -- so the first frame after the release renders sample 0 of the final pose.  That frame
is checked against a fresh one-sample render from the same generator state.  Latency = launch -> gathered
image on rank 0.  Prints one JSON line on rank 0.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "opencl-path-tracer_amd"))

ap = argparse.ArgumentParser()
ap.add_argument("--width", type=int, default=1920)
ap.add_argument("--height", type=int, default=1080)
ap.add_argument("--frames-moving", type=int, default=60)
ap.add_argument("--frames-still", type=int, default=60)
ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"], help="gloo: rehearse N ranks on one GPU")
a = ap.parse_args()

rank = int(os.environ.get("RANK", "0"))
world = int(os.environ.get("WORLD_SIZE", "1"))
local_rank = int(os.environ.get("LOCAL_RANK", "0"))

import numpy as np  # noqa: E402
import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

if a.backend == "gloo":
    local_rank %= torch.cuda.device_count()
torch.cuda.set_device(local_rank)
if world > 1:
    if a.backend == "nccl":
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
    else:
        dist.init_process_group("gloo")

import bdpt  # noqa: E402
import dist_render as dr  # noqa: E402

W, H = a.width, a.height
scene = bdpt.Scene.cornell()
cams, changed = bdpt.camera_orbit(8.0, (0, -1, 0), (1.0, 0.1, -0.1), 0.03, 0.03, 0.5 / 0.03, 0.0, a.frames_moving)
ctx = bdpt.Context(W, H, device=local_rank)
ctx.upload_scene(scene)
ctx.upload_seeds(bdpt.make_seeds(W, H, 0x5EED))
ctx.set_camera(cams[0])
image = torch.zeros((H, W, 4), dtype=torch.float32, device="cuda")
ctx.bind_accum(image.data_ptr())
ctx.follow_stream(torch.cuda.current_stream().cuda_stream)

lat = []
first_still = seeds_before_still = None
for f in range(a.frames_moving + a.frames_still):
    moving = f < a.frames_moving
    if f == a.frames_moving:
        seeds_before_still = ctx.read_seeds()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    dr.render_tile_split(ctx, H, W, 1, world, rank, wait=False)
    out = dr.gather_tile_split(image, H, world, dist if world > 1 else None)
    torch.cuda.synchronize()
    lat.append((time.perf_counter() - t0) * 1e3)
    if f == a.frames_moving:
        first_still = image.cpu().numpy().copy()     # this rank's tiles are what is checked below
    if moving:
        ctx.set_camera(cams[f + 1])          # held button: counter zeroed and camera uploaded after the frame (main.cpp:150-156)

# the first frame after the release must be sample 0 of the final pose: a fresh context, the generator state of that
# moment, one sample -- bit for bit on the pixels this rank owns
ok = True
if first_still is not None:
    with bdpt.Context(W, H, device=local_rank) as fresh:
        fresh.upload_scene(scene)
        fresh.upload_seeds(seeds_before_still)
        fresh.set_camera(cams[a.frames_moving])
        dr.render_tile_split(fresh, H, W, 1, world, rank)
        want = fresh.read_accum()
    mine = dr.tile_owner_map(H, W, world) == rank
    ok = bool(np.array_equal(first_still[mine].view(np.uint32), want[mine].view(np.uint32)))
if world > 1:
    flag = torch.tensor([1 if ok else 0], dtype=torch.int32, device="cuda" if a.backend == "nccl" else "cpu")
    dist.all_reduce(flag, op=dist.ReduceOp.MIN)
    ok = bool(flag.item())
if rank == 0:
    ms = np.array(lat)
    mv, st = ms[: a.frames_moving], ms[a.frames_moving:]
    print(json.dumps({"metric": "interactive frame latency, %dx%d, 1 spp/frame" % (W, H), "n_gpus": world,
                      "split": "tiles of %dx%d pixels dealt round-robin" % dr.TILE if world > 1 else "none",
                      "moving_ms_p50": round(float(np.percentile(mv, 50)), 3), "moving_ms_p99": round(float(np.percentile(mv, 99)), 3),
                      "still_ms_p50": round(float(np.percentile(st, 50)), 3), "still_ms_p99": round(float(np.percentile(st, 99)), 3),
                      "fps_p50": round(1e3 / float(np.percentile(ms, 50)), 1), "samples_accumulated": ctx.num_samples,
                      "first_still_frame_equals_fresh_1spp_render": ok}))
ctx.close()
if world > 1:
    dist.barrier()
    dist.destroy_process_group()
if not ok:
    sys.exit("the first frame after the camera stopped is not a fresh one-sample render")

"""Time-to-target RMSE (the second half of the headline metric): wall time from a camera reset until
RMSE(accumulated image, reference) < target, sampled every `--chunk` samples per pixel.

    python tools/time_to_rmse.py [--width 1920 --height 1080 --ref-spp 16384 --chunk 64 --target 0.02]
    python -m torch.distributed.run --nproc-per-node N tools/time_to_rmse.py ...   (sample split + RCCL reduce per chunk)

The reference image is a --ref-spp render with a different seed set (STREAM mode, the reference's own
sequence).  RMSE is taken on linear RGB clamped to [0,1], over the whole frame and, separately, over the
pixels that are not frozen in the reference image: in STREAM mode a pixel one of whose samples returned at the
eye ray keeps that raw value for good (/root/reference/src/kernel/kernel0.cl:645-656).  The curve also lists
RMSE * sqrt(spp): it is not constant, because the estimator's connection term 1/r^2 (kernel0.cl:719) is unbounded
and single samples outweigh thousands; --falloff-c C renders both images with the bounded 1/(C + r^2) instead.
Prints one JSON line on rank 0.
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
ap.add_argument("--ref-spp", type=int, default=16384)
ap.add_argument("--chunk", type=int, default=64, help="samples per pixel (whole job) between RMSE checks")
ap.add_argument("--target", type=float, default=0.02)
ap.add_argument("--max-spp", type=int, default=8192)
ap.add_argument("--falloff-c", type=float, default=None, help="use connection_falloff 1/(c + r^2) instead of 1/r^2")
a = ap.parse_args()
extra = {} if a.falloff_c is None else dict(connection_falloff=1, falloff_c=a.falloff_c)

rank = int(os.environ.get("RANK", "0"))
world = int(os.environ.get("WORLD_SIZE", "1"))
local_rank = int(os.environ.get("LOCAL_RANK", "0"))

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

torch.cuda.set_device(local_rank)
if world > 1:
    dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))

import bdpt  # noqa: E402
import dist_render as dr  # noqa: E402

W, H = a.width, a.height
scene = bdpt.Scene.cornell()
dev = torch.device("cuda", local_rank)


def make_ctx(seed, **params):
    ctx = bdpt.Context(W, H, device=local_rank)
    ctx.set_params(bdpt.default_params(**params))
    ctx.upload_scene(scene)
    ctx.upload_seeds(bdpt.make_seeds(W, H, seed))
    ctx.set_camera(scene.camera)
    buf = torch.zeros((H, W, 4), dtype=torch.float32, device=dev)
    ctx.bind_accum(buf.data_ptr())
    ctx.follow_stream(torch.cuda.current_stream().cuda_stream)
    return ctx, buf


ref = live = None
if rank == 0:
    import numpy as np
    rctx, rbuf = make_ctx(0xBEEF, **extra)
    rctx.render(a.ref_spp)
    ref = rbuf[..., :3].clamp(0, 1).clone()
    before = rctx.read_seeds()                       # frozen pixels: one more sample does not move their generator word
    rctx.render(1)
    xs, ys = np.meshgrid(np.arange(W), np.arange(H))
    idx = (xs * H + ys) * 4 + 3
    live = torch.from_numpy(before[idx] != rctx.read_seeds()[idx]).to(dev)
    rctx.close()

ctx, part = make_ctx(0x5EED, rng_mode=bdpt.RNG_STRIDED, accum_mode=bdpt.ACCUM_SUM, **extra)
ctx.render(1)                      # warm-up: code object load and first-touch are not part of the measurement
ctx.set_camera(scene.camera)       # the reset the clock starts at
total = torch.zeros_like(part)
if rank == 0:                      # the same for the device code of the RMSE evaluation
    total += part
    warm = ((total[..., :3] / 1).clamp(0, 1) - ref) ** 2
    torch.sqrt(torch.mean(warm)).item()
    torch.sqrt(torch.mean(warm[live])).item()
    total.zero_()
if world > 1:
    dist.reduce(part, dst=0, op=dist.ReduceOp.SUM)
    dist.barrier()
torch.cuda.synchronize()
t0 = time.perf_counter()
done, hit, curve = 0, None, []
while done < a.max_spp:
    first, count = dr.sample_range(a.chunk, world, rank)
    ctx.set_sample_offset(done + first)
    ctx.num_samples = 0
    if count:
        ctx.render_async(count)
    else:
        part.zero_()
    if world > 1:
        dist.reduce(part, dst=0, op=dist.ReduceOp.SUM)
    done += a.chunk
    stop = torch.zeros(1, device=dev)
    if rank == 0:
        total += part
        sq = ((total[..., :3] / done).clamp(0, 1) - ref) ** 2
        err = torch.sqrt(torch.mean(sq)).item()
        t = time.perf_counter() - t0
        err_live = torch.sqrt(torch.mean(sq[live])).item()
        curve.append((done, round(t * 1e3, 2), round(err, 5), round(err_live, 5), round(err_live * done ** 0.5, 3)))
        if err < a.target:
            hit = (done, t)
            stop += 1
    if world > 1:
        dist.broadcast(stop, src=0)
    if stop.item() > 0:
        break
if rank == 0:
    print(json.dumps({"metric": "time to RMSE < %g at %dx%d Cornell box" % (a.target, W, H), "n_gpus": world,
                      "seconds": None if hit is None else round(hit[1], 4), "spp_at_target": None if hit is None else hit[0],
                      "reference_spp": a.ref_spp, "chunk_spp": a.chunk, "connection_falloff": "1/r^2" if a.falloff_c is None else "1/(%g + r^2)" % a.falloff_c,
                      "frozen_fraction_of_reference": round(1.0 - float(live.float().mean().item()), 4),
                      "curve_spp_ms_rmse_rmseLive_rmseLiveSqrtN": curve[:: max(1, len(curve) // 12)] + curve[-1:]}))
ctx.close()
if world > 1:
    dist.barrier()
    dist.destroy_process_group()

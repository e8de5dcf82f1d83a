"""BDPT throughput benchmark: Cornell box, 1920x1080, 1024 samples per pixel per step.

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

One step = one pass of the hot path over one batch: every rank renders the full
frame for its own `spp` samples of every pixel (STRIDED generator mode, rank r
owns sample indices [r*spp, (r+1)*spp)) into a sum accumulator, then the partial
sums are combined on rank 0 by one RCCL reduce over xGMI.  Per-GPU work is fixed
as N grows (weak scaling); value = all ranks' paths / max-over-ranks time.
One path = one pixel-sample = one execution of the reference's
bi_directional_path_trace (/root/reference/src/kernel/kernel0.cl:571-864).

Rank 0 prints one JSON line.  `roofline` prices the render kernel against the
FP32 vector peak with the algorithmic flop count (rays counted by the device in
a separate, untimed launch of the same workload); `cpu_baseline` times the CPU
oracle on this host's cores on a bounded crop of the same workload.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(ROOT, "opencl-path-tracer_amd"))

PEAK_FP32_VALU_TFLOPS = 157.3   # MI355X vector FP32 peak (256 CU x 4 SIMD x 64 flop/clk x 2.4 GHz)
F_SPHERE, F_PLANE, F_TRIANGLE, F_RAY_FIXED = 35, 37, 113, 30   # algorithmic flops per primitive test / per ray
F_PATH_FULL, F_PATH_EARLY = 500, 100                           # per path outside the closest-hit queries


def hbm_traffic_bytes(workload, width, height, spp):
    """HBM-side bytes per launch of the render kernel from the committed rocprofv3 PMC passes (FETCH_SIZE doubled as the
    gfx950 correction for wide coalesced reads, plus WRITE_SIZE), or None when no pass was recorded for this workload
    and image size."""
    path = os.path.join(ROOT, "profiles", "hbm_traffic.json")
    try:
        with open(path) as f:
            rec = json.load(f).get("%s/%dx%d" % (workload, width, height))
    except (OSError, ValueError):
        return None
    return None if rec is None else rec["fixed"] + rec["per_spp"] * spp


def cpu_opencl_status():
    """Whether the reference's OpenCL kernel could be timed on this host: it needs a CPU OpenCL device (PoCL or the
    like).  Returns a short reason string; this build has no harness for that route because no image it was developed
    on has such a device."""
    import ctypes
    try:
        cl = ctypes.CDLL("libOpenCL.so.1")
    except OSError:
        return "unavailable: no OpenCL loader on this host"
    n = ctypes.c_uint(0)
    if cl.clGetPlatformIDs(0, None, ctypes.byref(n)) != 0 or n.value == 0:
        return "unavailable: no OpenCL platform"
    plats = (ctypes.c_void_p * n.value)()
    cl.clGetPlatformIDs(n.value, plats, None)
    CL_DEVICE_TYPE_CPU = 1 << 1
    cpus = 0
    for p in plats:
        m = ctypes.c_uint(0)
        if cl.clGetDeviceIDs(ctypes.c_void_p(p), ctypes.c_ulong(CL_DEVICE_TYPE_CPU), 0, None, ctypes.byref(m)) == 0:
            cpus += m.value
    if cpus == 0:
        return "unavailable: %d OpenCL platform(s), none with a CPU device" % n.value
    return "not run: a CPU OpenCL device exists but this build has no harness for the reference kernel"


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--width", type=int, default=1920)
    ap.add_argument("--height", type=int, default=1080)
    ap.add_argument("--spp", type=int, default=None,
                    help="samples per pixel per rank per step (default: 1024 for c2, 128 for c3, 16 for c4)")
    ap.add_argument("--kernel", type=int, default=0, help="bdpt_params.kernel_variant (0 auto)")
    ap.add_argument("--workload", default="c2", choices=["c2", "c3", "c4"],
                    help="c2: Cornell box (the headline configuration); c3: outdoor sphere arrangement, 3840x2160 unless "
                         "--width/--height are given; c4: Cornell box + 100k-triangle soup")
    ap.add_argument("--soup", type=int, default=100000)
    ap.add_argument("--order", default="auto", choices=["auto", "index", "class", "adaptive"],
                    help="bdpt_params.sample_order of the STRIDED + SUM launches (auto = adaptive)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"],
                    help="torch.distributed backend for N > 1; gloo only to rehearse the N > 1 code path with several ranks on one GPU")
    ap.add_argument("--stream-line", action="store_true",
                    help="also time one launch of the same frame in STREAM + MEAN mode (the reference's generator and blend)")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            sys.exit("bench.py --gpus %d must be launched with torch.distributed.run --nproc-per-node %d" % (args.gpus, args.gpus))
        args.gpus = world

    import torch   # before the binding, so that both share one HIP runtime in this process
    import torch.distributed as dist

    if not torch.cuda.is_available():
        sys.exit("bench.py needs a GPU: the path tracer has no CPU path")
    if args.backend == "gloo":
        local_rank %= torch.cuda.device_count()
    torch.cuda.set_device(local_rank)
    if world > 1:
        if args.backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group("gloo")

    import numpy as np

    import bdpt

    if args.workload == "c3" and (args.width, args.height) == (1920, 1080):
        args.width, args.height = 3840, 2160
    if args.spp is None:
        args.spp = {"c2": 1024, "c3": 128, "c4": 16}[args.workload]
    W, H, SPP = args.width, args.height, args.spp
    if args.workload == "c2":
        scene, scene_name = bdpt.Scene.cornell(), "Cornell box (3 spheres + 12 triangles)"
    elif args.workload == "c3":
        scene, scene_name = bdpt.Scene.outdoor(), "outdoor sphere arrangement (6 spheres, no walls)"
    else:
        scene, scene_name = bdpt.Scene.soup(args.soup), "Cornell box + %d-triangle soup" % args.soup
    ctx = bdpt.Context(W, H, device=local_rank)
    order = {"auto": bdpt.ORDER_AUTO, "index": bdpt.ORDER_INDEX, "class": bdpt.ORDER_ROULETTE_CLASS,
             "adaptive": bdpt.ORDER_ADAPTIVE}[args.order]
    ctx.set_params(bdpt.default_params(rng_mode=bdpt.RNG_STRIDED, accum_mode=bdpt.ACCUM_SUM, kernel_variant=args.kernel,
                                       sample_order=order))
    ctx.upload_scene(scene)
    ctx.upload_seeds(bdpt.make_seeds(W, H, 0x5EED))
    ctx.set_camera(scene.camera)
    ctx.set_sample_offset(rank * SPP)
    accum = torch.zeros((H, W, 4), dtype=torch.float32, device="cuda")
    ctx.bind_accum(accum.data_ptr())
    ctx.set_stream(torch.cuda.current_stream().cuda_stream)

    launch_ms = []

    def step(record):
        ctx.num_samples = 0                 # a fresh image each step; the first sample overwrites the accumulator
        ctx.render_async(SPP)
        if world > 1 and args.backend == "nccl":
            dist.reduce(accum, dst=0, op=dist.ReduceOp.SUM)
        elif world > 1:                     # rehearsal: gloo reduces host tensors
            host = accum.cpu()
            dist.reduce(host, dst=0, op=dist.ReduceOp.SUM)
            if rank == 0:
                accum.copy_(host)
        if record:
            launch_ms.append(ctx.last_launch_ms())

    def fence():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    # set-up that is not a step: for N > 1, build the communicator (it is built on first use)
    if world > 1:
        probe = torch.zeros(1, device="cuda" if args.backend == "nccl" else "cpu")
        dist.all_reduce(probe)
    for _ in range(args.warmup):
        step(False)
    fence()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step(True)
    fence()
    elapsed = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device="cuda" if args.backend == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    paths_per_step = world * W * H * SPP
    value = paths_per_step * args.steps / elapsed / 1e6
    out = {
        "metric": "BDPT Mpaths/s (whole node) at 1080p Cornell box",
        "value": round(value, 2),
        "unit": "Mpaths/s",
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": round(elapsed / args.steps * 1e3, 3),
        "higher_is_better": True,
        "scaling": "weak",
        "vs_baseline": None,
        "dtype": "f32",
        "data": "synthetic",
        "config": {
            "workload": "%s, %dx%d, %d spp per GPU per step, exhaustive closest hit" % (scene_name, W, H, SPP),
            "rng_mode": "strided", "accumulate": "sum", "sample_order": {"auto": "adaptive (roulette classes for pixels whose first eye ray reaches a surface)", "index": "index",
                             "class": "roulette class", "adaptive": "adaptive"}[args.order],
            "split": "samples per pixel" if world > 1 else "none",
            "combine": "RCCL reduce(sum) of %dx%dx4 f32 to rank 0" % (W, H) if world > 1 else "none",
        },
    }

    if rank == 0:
        # algorithmic work of one launch: rays and path classes counted by device atomics, untimed
        ctx.enable_counters(True)
        ctx.num_samples = 0
        ctx.render(SPP)
        rays, full, early = ctx.read_counters()
        ctx.enable_counters(False)
        n_s, n_p, n_t = len(scene.spheres), len(scene.planes), len(scene.triangles)
        f_ray = F_SPHERE * n_s + F_PLANE * n_p + F_TRIANGLE * n_t + F_RAY_FIXED
        flops = rays * f_ray + full * F_PATH_FULL + early * F_PATH_EARLY
        avg_s = sum(launch_ms) / len(launch_ms) / 1e3
        achieved = flops / avg_s / 1e12
        note = ("the flop count prices every ray of the reference formulation (counted by a launch that traces them all); "
                "the timed launch leaves out rays and sampler evaluations whose results nothing reads "
                "(the bounce ray of the last subpath vertex, bounce directions that roulette discards)")
        if achieved > PEAK_FP32_VALU_TFLOPS:
            note = ("algorithmic flops are the reference formulation's (113 per ray-triangle pair); the tiled kernel decides "
                    "most pairs with a 25-flop bounding-sphere test, so it does less arithmetic than that")
        out["roofline"] = {
            "bound": "valu_fp32", "achieved": round(achieved, 3), "peak": PEAK_FP32_VALU_TFLOPS, "unit": "TFLOP/s",
            "frac": round(achieved / PEAK_FP32_VALU_TFLOPS, 4), "traffic": hbm_traffic_bytes(args.workload, W, H, SPP),
            "kernel_ms": round(avg_s * 1e3, 3), "flops_per_launch": flops, "rays_per_launch": rays,
            "paths_full": full, "paths_early": early, "flops_per_ray": f_ray,
        }
        if note:
            out["roofline"]["note"] = note
        # the same launch against the HBM roofline, to show that it is not the one that binds:
        # 16 B accumulator read + 16 B write + 4 B seed word per pixel per launch, whatever spp is
        hbm = W * H * 36 / avg_s / 1e9
        out["roofline_hbm"] = {"bound": "hbm", "achieved": round(hbm, 3), "peak": 8000.0, "unit": "GB/s",
                               "frac": round(hbm / 8000.0, 6), "traffic": out["roofline"]["traffic"],
                               "algorithmic_bytes": W * H * 36}
        # the same frame with the reference's sequential generator and running-mean blend (one launch, untimed above):
        # samples are taken in index order there, so the wavefronts pay for roulette divergence
        if args.stream_line:
            ctx.set_params(bdpt.default_params(kernel_variant=args.kernel))
            ctx.num_samples = 0
            ctx.render(SPP)
            out["stream_mode"] = {"kernel_ms": round(ctx.last_launch_ms(), 3),
                                  "mpaths_per_s": round(W * H * SPP / ctx.last_launch_ms() / 1e3, 2)}

    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        sys.path.insert(0, os.path.join(ROOT, "oracle"))
        import oracle_bind as oracle   # the checker, timed here as the CPU baseline

        threads = min(os.cpu_count() or 1, int(os.environ.get("OMP_NUM_THREADS", "16")), 16)
        cw, ch, cspp = min(W, 640), min(H, 360), 512
        region = ((W - cw) // 2, (H - ch) // 2, cw, ch)
        st = oracle.State(W, H, bdpt.make_seeds(W, H, 0x5EED))
        p = bdpt.default_params(rng_mode=bdpt.RNG_STRIDED, accum_mode=bdpt.ACCUM_SUM)
        t0 = time.perf_counter()
        oracle.render(st, scene, scene.camera, p, cspp, region=region, n_threads=threads)
        cpu_s = time.perf_counter() - t0
        out["cpu_baseline"] = {
            "value": round(cw * ch * cspp / cpu_s / 1e6, 3), "unit": "Mpaths/s", "cores": threads, "kind": "port",
            "sample": "centre %dx%d crop of the %dx%d frame, %d spp, same scene/seeds/mode; %.1f s" % (cw, ch, W, H, cspp, cpu_s),
            "reference_opencl_on_cpu": cpu_opencl_status(),
        }

    if rank == 0:
        print(json.dumps(out), flush=True)
    ctx.close()
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()

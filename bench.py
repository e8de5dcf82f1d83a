"""BDPT throughput benchmark: Cornell box, 1920x1080, 1024 samples per pixel per step.

    python bench.py --gpus N --steps K --warmup W

With N > 1 and no launcher around it the script starts its N ranks itself (a
`torch.distributed.run` child process, started before anything here touches the
GPU) and passes their output through; under `torch.distributed.run` it is one of
the ranks.

One step = one pass of the hot path over one batch: every rank renders the full
frame for its own `spp` samples of every pixel (STRIDED generator mode, rank r
owns sample indices [r*spp, (r+1)*spp)) into a sum accumulator, then the partial
sums are combined on rank 0 by one RCCL reduce over xGMI
(dist_render.render_sample_split / combine_sample_split).  Per-GPU work is fixed
as N grows (weak scaling); value = all ranks' paths / max-over-ranks time.
One path = one pixel-sample = one execution of the reference's
bi_directional_path_trace (/root/reference/src/kernel/kernel0.cl:571-864).

With N > 1 the weak-scaling steps are followed by as many strong-scaling steps:
the samples of ONE GPU's step dealt to the N ranks (`strong_scaling` in the line).

Rank 0 prints one JSON line.  `roofline` prices the render kernel against the
FP32 vector peak (its `traffic` is measured by two rocprofv3 counter passes run
as child processes, N = 1); `time_to_rmse` (N = 1) is the wall time from a camera reset
until the image is within RMSE 0.02 of a converged reference; `cpu_baseline`
times the CPU oracle on this host's cores on a bounded sample of the same
workload and, when this host has a CPU OpenCL device and the reference's kernel
file, the reference's own kernel on it (tools/cl_baseline, a child process).
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(ROOT, "opencl-path-tracer_amd"))

PEAK_FP32_VALU_TFLOPS = 157.3   # MI355X vector FP32 peak (256 CU x 4 SIMD x 64 lanes x 2 flop x 2.4 GHz)
F_SPHERE, F_PLANE, F_TRIANGLE, F_RAY_FIXED = 35, 37, 113, 30   # algorithmic flops per primitive test / per ray
F_PATH_FULL, F_PATH_EARLY = 500, 100                           # per path outside the closest-hit queries
F_CULL = {4: 15, 5: 29, 6: 14}  # flops of the cheap test in front of the triangle test: bounding sphere (TILED: 7 fma + compare);
                                # conservative plane test (LDS2), for two triangles that share only the plane; see f_cull_lds2;
                                # one box of the hierarchy (BVH: 6 subtractions, 6 products, 2 scalings; the counter counts boxes)


def recorded(name, key):
    """An entry of one of the committed measurement summaries under profiles/ (None if absent)."""
    try:
        with open(os.path.join(ROOT, "profiles", name)) as f:
            return json.load(f).get(key)
    except (OSError, ValueError):
        return None


def counter_pass(workload_args, counters, timeout_s=150):
    """One `rocprofv3 --pmc` pass (a child process running tools/prof_run.py on the given workload): the named hardware
    counters of the run's last render launch.  Returns (dict, None) or (None, reason)."""
    import csv
    import glob
    import shutil
    import tempfile
    exe = shutil.which("rocprofv3")
    if exe is None:
        return None, "rocprofv3 is not on PATH"
    with tempfile.TemporaryDirectory(dir="/tmp") as tmp:
        cmd = [exe, "--pmc"] + list(counters) + ["--output-format", "csv", "-d", tmp, "--",
               sys.executable, os.path.join(ROOT, "tools", "prof_run.py"), "--launches", "1"] + workload_args
        try:
            r = subprocess.run(cmd, cwd="/tmp", env=dict(os.environ, TMPDIR="/tmp"), capture_output=True, text=True, timeout=timeout_s)
        except subprocess.TimeoutExpired:
            return None, "the pass for %s did not finish within %d s" % (" ".join(counters), timeout_s)
        if r.returncode != 0:
            return None, "the pass for %s failed (exit status %d)" % (" ".join(counters), r.returncode)
        got = {}
        for f in sorted(glob.glob(os.path.join(tmp, "**", "*counter_collection.csv"), recursive=True)):
            for row in csv.DictReader(open(f)):
                if "bdpt" in row["Kernel_Name"] and "query" not in row["Kernel_Name"]:
                    got[row["Counter_Name"]] = float(row["Counter_Value"])     # rows come in launch order: the last launch stays
        ms = [ln for ln in r.stdout.splitlines() if ln.startswith("scene=")]
        if ms:
            try:
                got["_kernel_ms"] = float(ms[-1].split("[")[1].split("]")[0].split(",")[-1].strip(" '"))
            except (IndexError, ValueError):
                pass
    missing = [c for c in counters if c not in got]
    if missing:
        return None, "the pass recorded no %s for the render kernel" % " ".join(missing)
    return got, None


def measured_traffic(workload_args):
    """HBM-side bytes of one launch of the render kernel, measured now: FETCH_SIZE and WRITE_SIZE of the same workload in
    two separate counter passes, FETCH_SIZE doubled as gfx950 needs for wide reads.  Returns (bytes, text) or (None, reason)."""
    got = {}
    for counter in ("FETCH_SIZE", "WRITE_SIZE"):
        c, why = counter_pass(workload_args, [counter])
        if c is None:
            return None, why
        got[counter] = c[counter]
    total = int((2.0 * got["FETCH_SIZE"] + got["WRITE_SIZE"]) * 1024.0)
    return total, ("measured by this run: rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE, one pass each, of `tools/prof_run.py %s` "
                   "(the timed launch's workload and launch form); FETCH_SIZE=%.0f KB doubled (gfx950 counts half the bytes of wide "
                   "reads; an upper bound for narrow ones), WRITE_SIZE=%.0f KB; L2-to-fabric boundary, Infinity Cache hits included"
                   % (" ".join(workload_args), got["FETCH_SIZE"], got["WRITE_SIZE"]))


def measured_executed(workload_args):
    """The hardware's view of one launch, measured now in one counter pass: FP32 vector operations the lanes performed
    ((2 FMA + MUL + ADD wave-instructions) x 64 x lane utilisation) per second, and what the SIMDs did meanwhile.
    Packed two-wide instructions are billed once, so a kernel that uses them performs more than this says."""
    c, why = counter_pass(workload_args, ["SQ_INSTS_VALU", "SQ_ACTIVE_INST_VALU", "SQ_THREAD_CYCLES_VALU", "SQ_INSTS_VALU_FMA_F32",
                                          "SQ_INSTS_VALU_MUL_F32", "SQ_INSTS_VALU_ADD_F32", "SQ_WAVE_CYCLES", "SQ_WAIT_ANY", "GRBM_GUI_ACTIVE"])
    if c is None:
        return None, why
    clocks = c["GRBM_GUI_ACTIVE"] / 8.0                                   # the counter adds up the eight dies
    ms = c.get("_kernel_ms", clocks / 2.4e6)
    util = c["SQ_THREAD_CYCLES_VALU"] / (64.0 * c["SQ_ACTIVE_INST_VALU"])
    flops = (2 * c["SQ_INSTS_VALU_FMA_F32"] + c["SQ_INSTS_VALU_MUL_F32"] + c["SQ_INSTS_VALU_ADD_F32"]) * 64 * util
    return {
        "source": "measured by this run: one rocprofv3 --pmc pass of `tools/prof_run.py %s`" % " ".join(workload_args),
        "kernel_ms_under_profiler": round(ms, 2),
        "valu_wave_insts": c["SQ_INSTS_VALU"],
        "fp32_arith_fraction_of_valu": round((c["SQ_INSTS_VALU_FMA_F32"] + c["SQ_INSTS_VALU_MUL_F32"] + c["SQ_INSTS_VALU_ADD_F32"]) / c["SQ_INSTS_VALU"], 3),
        "lane_utilisation": round(util, 3),
        "executed_tflops": round(flops / (ms * 1e-3) / 1e12, 2),
        "executed_frac_of_157.3": round(flops / (ms * 1e-3) / 1e12 / PEAK_FP32_VALU_TFLOPS, 3),
        "simd_cycles_per_valu_inst": round(1024 * clocks / c["SQ_INSTS_VALU"], 2),
        "wavefronts_per_simd_mean": round(c["SQ_WAVE_CYCLES"] * 4.0 / (1024 * clocks), 2),
        "wave_cycles_waiting_fraction": round(c["SQ_WAIT_ANY"] / c["SQ_WAVE_CYCLES"], 3),
    }, None


def reference_on_cpu_opencl(scene, camera, seeds, width, height, frames=16, warmup=2, timeout_s=240):
    """The reference's own kernel on a CPU OpenCL device of this host, through tools/cl_baseline run as a child process
    (a fresh process: nothing of it shares this one's GPU state).  Returns (lines, reason): the harness's JSON lines, one
    per build option, or an empty list and the reason it gave ("pocl baseline unavailable: ...")."""
    import tempfile
    exe = os.path.join(ROOT, "tools", "cl_baseline")
    if not os.path.exists(exe):
        return [], "unavailable: tools/cl_baseline is not built"
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import cl_input
    with tempfile.TemporaryDirectory() as tmp:
        blob = os.path.join(tmp, "scene.bin")
        cl_input.write(blob, scene, camera, seeds, width, height)
        try:
            r = subprocess.run([exe, "--input", blob, "--frames", str(frames), "--warmup", str(warmup)], capture_output=True, text=True,
                               timeout=timeout_s)
        except subprocess.TimeoutExpired:
            return [], "not finished within %d s" % timeout_s
    lines = []
    for ln in r.stdout.splitlines():
        if ln.startswith("{"):
            try:
                lines.append(json.loads(ln))
            except ValueError:
                pass
    if lines:
        return lines, None
    text = (r.stdout.strip() or r.stderr.strip() or "exit status %d" % r.returncode).splitlines()[-1]
    return [], text.replace("pocl baseline unavailable: ", "unavailable: ")


def start_ranks(n):
    """N > 1 without a launcher: run this script again as N ranks under torch.distributed.run, as a child process.
    Nothing in this process has touched the GPU at this point, and nothing will."""
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n),
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    return subprocess.run(cmd).returncode


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
    ap.add_argument("--no-stream-line", action="store_true",
                    help="skip the extra launch of the same frame in STREAM + MEAN mode (the reference's generator and blend)")
    ap.add_argument("--dump-image", default=None, help="rank 0 saves the combined mean image of the last step (.npy)")
    ap.add_argument("--dump-strong-image", default=None, help="N > 1: rank 0 saves the combined mean image of the last strong-scaling step (.npy)")
    ap.add_argument("--no-traffic-pass", action="store_true",
                    help="do not measure roofline.traffic with rocprofv3 child processes (N = 1); quote the committed figure instead")
    ap.add_argument("--no-time-to-rmse", action="store_true", help="skip the time-to-target-RMSE measurement (N = 1, c2)")
    ap.add_argument("--rmse-target", type=float, default=0.02)
    ap.add_argument("--rmse-ref-spp", type=int, default=16384)
    ap.add_argument("--strong-spp", type=int, default=None, help="N > 1: total samples per pixel of the strong-scaling steps (default: --spp)")
    args = ap.parse_args()

    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        sys.exit(start_ranks(args.gpus))

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
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
    import dist_render as dr

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
    bench_params = bdpt.default_params(rng_mode=bdpt.RNG_STRIDED, accum_mode=bdpt.ACCUM_SUM, kernel_variant=args.kernel,
                                       sample_order=order)
    ctx.set_params(bench_params)
    ctx.upload_scene(scene)
    ctx.upload_seeds(bdpt.make_seeds(W, H, 0x5EED))
    ctx.set_camera(scene.camera)
    accum = torch.zeros((H, W, 4), dtype=torch.float32, device="cuda")
    ctx.bind_accum(accum.data_ptr())
    ctx.follow_stream(torch.cuda.current_stream().cuda_stream)   # launches, the reduce and the events below: one stream

    launch_ms, reduce_ms = [], []
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
    image = None

    def step(record):
        nonlocal image
        # a fresh image each step: rank r renders samples [r*SPP, (r+1)*SPP) of world*SPP
        dr.render_sample_split(ctx, world * SPP, world, rank, wait=False)
        ev[0].record()
        image = dr.combine_sample_split(accum, world * SPP, dist if world > 1 else None)
        ev[1].record()
        if record:
            launch_ms.append(ctx.last_launch_ms())
            if world > 1:
                ev[1].synchronize()
                reduce_ms.append(ev[0].elapsed_time(ev[1]))

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
    per_rank_ms = [sum(launch_ms) / len(launch_ms)]
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device="cuda" if args.backend == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
        gathered = [None] * world
        dist.all_gather_object(gathered, per_rank_ms[0])
        per_rank_ms = gathered

    variant = ctx.last_launch_variant()
    variant_name = {1: "DIRECT", 3: "LDS", 4: "TILED", 5: "LDS2", 6: "BVH"}.get(variant, str(variant))
    weak_chunks = ctx.last_launch_chunks()

    # Strong scaling (N > 1): the SAME total work as one GPU's step -- `strong_spp` samples per pixel in all -- dealt to the
    # ranks (rank r renders sample indices [r*S/N, (r+1)*S/N)), combined by the same reduce.  Start-up, table fill, reduce and
    # rank skew weigh N times more against the render here than in the weak-scaling steps above.
    strong = None
    if world > 1:
        strong_spp = args.strong_spp or SPP
        s_launch, s_reduce, s_image = [], [], [None]

        def strong_step(record):
            dr.render_sample_split(ctx, strong_spp, world, rank, wait=False)
            ev[0].record()
            s_image[0] = dr.combine_sample_split(accum, strong_spp, dist)
            ev[1].record()
            if record:
                s_launch.append(ctx.last_launch_ms() if dr.sample_range(strong_spp, world, rank)[1] > 0 else 0.0)
                ev[1].synchronize()
                s_reduce.append(ev[0].elapsed_time(ev[1]))

        for _ in range(max(1, args.warmup)):
            strong_step(False)
        fence()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            strong_step(True)
        fence()
        s_elapsed = time.perf_counter() - t0
        t = torch.tensor([s_elapsed], dtype=torch.float64, device="cuda" if args.backend == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        s_elapsed = float(t.item())
        gathered = [None] * world
        dist.all_gather_object(gathered, sum(s_launch) / len(s_launch))
        strong = {
            "total_spp": strong_spp, "spp_per_rank": [dr.sample_range(strong_spp, world, r)[1] for r in range(world)],
            "ms_per_step": round(s_elapsed / args.steps * 1e3, 3),
            "mpaths_per_s": round(W * H * strong_spp * args.steps / s_elapsed / 1e6, 2),
            "per_rank_kernel_ms": [round(m, 3) for m in gathered],
            "reduce_ms": round(sum(s_reduce) / len(s_reduce), 3),
            "what": "fixed total work: one step renders total_spp samples per pixel in all, dealt to the ranks, one reduce; "
                    "speed-up over one GPU = this mpaths_per_s / the N=1 line's value",
        }
        if rank == 0 and args.dump_strong_image:
            np.save(args.dump_strong_image, s_image[0].cpu().numpy())
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
            "workload": "%s, %dx%d, %d spp per GPU per step, %s" % (scene_name, W, H, SPP, "closest hit through a hierarchy over the triangles (same hits as the exhaustive search)" if variant == bdpt.KERNEL_BVH else "exhaustive closest hit"),
            "rng_mode": "strided", "accumulate": "sum", "sample_order": {"auto": "adaptive (roulette classes for pixels whose first eye ray reaches a surface)", "index": "index",
                             "class": "roulette class", "adaptive": "adaptive"}[args.order],
            "kernel_variant": variant_name,
            "launch": "%d runs per workgroup tile (8x8 pixels), each a workgroup of its own (bdpt_set_launch_chunks: automatic)" % weak_chunks,
            "split": "samples per pixel" if world > 1 else "none",
            "combine": ("%s reduce(sum) of %dx%dx4 f32 to rank 0" % ("RCCL" if args.backend == "nccl" else "gloo (host tensors; rehearsal)", W, H)) if world > 1 else "none",
        },
    }
    if world > 1:
        out["per_rank_kernel_ms"] = [round(m, 3) for m in per_rank_ms]
        out["reduce_ms"] = round(sum(reduce_ms) / len(reduce_ms), 3)     # rank 0: end of its launch -> combined image; includes waiting for the slowest rank
        out["strong_scaling"] = strong

    if rank == 0 and args.dump_image:
        np.save(args.dump_image, image.cpu().numpy())

    if rank == 0:
        # Work of one launch, counted by device atomics in untimed launches of the same workload:
        #  - as the timed launch traces it (no parity buffers: rays whose result nothing reads are left out);
        #  - as the reference formulates it (parity buffers on: every ray of kernel0.cl:571-864 is traced).
        ctx.use_own_stream()
        ctx.bind_accum(None)
        ctx.set_sample_offset(0)
        ctx.enable_counters(True)
        ctx.num_samples = 0
        ctx.render(SPP)
        rays, full, early, n_cull, n_exact = ctx.read_counters()
        ctx.set_params(bdpt.default_params(rng_mode=bdpt.RNG_STRIDED, accum_mode=bdpt.ACCUM_SUM, kernel_variant=args.kernel,
                                           sample_order=order, aov_enable=1))
        ctx.set_camera(scene.camera)
        ctx.render(SPP)
        rays_ref, full_ref, early_ref = ctx.read_counters()[:3]
        ctx.enable_counters(False)
        ctx.set_params(bench_params)
        assert (full, early) == (full_ref, early_ref)
        n_s, n_p, n_t = len(scene.spheres), len(scene.planes), len(scene.triangles)
        f_ray = F_SPHERE * n_s + F_PLANE * n_p + F_TRIANGLE * n_t + F_RAY_FIXED
        f_paths = full * F_PATH_FULL + early * F_PATH_EARLY
        flops_ref = rays_ref * f_ray + f_paths
        f_cull = F_CULL.get(variant)
        if variant == 5:
            # per cull test: the plane part (19 flops) is shared by the two triangles of a pair; a triangle's own part
            # is 19 flops (two coordinate planes, the third coordinate, minimum, margin, compare), 12.5 for each half
            # of a parallelogram, which shares the coordinate planes as well
            # two opposite walls of a box (four triangles) go through one such test of 42 flops
            quads, pairs, singles, slabs = ctx.two_stage_record_counts()
            f_cull = (slabs * 42 + quads * 44 + pairs * 57 + singles * 38) / max(1, 4 * slabs + 2 * quads + 2 * pairs + singles)
        if variant in F_CULL:
            # a cheap conservative test decides most (ray, triangle) pairs; price the tests that were run
            flops = int(rays * (F_SPHERE * n_s + F_PLANE * n_p + F_RAY_FIXED) + n_cull * f_cull + n_exact * F_TRIANGLE + f_paths)
        else:
            flops = rays * f_ray + f_paths
        avg_s = sum(launch_ms) / len(launch_ms) / 1e3
        achieved = flops / avg_s / 1e12
        traffic = recorded("hbm_traffic.json", "%s/%dx%d" % (args.workload, W, H))
        if args.workload == "c4" and variant != bdpt.KERNEL_TILED:
            traffic = None                                   # the figures on file are the exhaustive kernel's
        traffic_bytes = None if traffic is None else traffic["fixed"] + traffic["per_spp"] * SPP + traffic.get("per_extra_run", 0) * (weak_chunks - 1)
        traffic_source = None if traffic is None else traffic.get("source", "profiles/hbm_traffic.json: rocprofv3 --pmc passes of an earlier run of this command, not this run")
        if world == 1 and not args.no_traffic_pass:
            scene_arg = {"c2": "cornell", "c3": "outdoor", "c4": "soup:%d" % args.soup}[args.workload]
            workload_args = ["--spp", str(SPP), "--kernel", str(variant), "--scene", scene_arg, "--width", str(W), "--height", str(H)]
            live, text = measured_traffic(workload_args)
            if live is not None:
                if traffic_bytes is not None:
                    text += "; profiles/hbm_traffic.json (an earlier session) gives %d" % traffic_bytes
                traffic_bytes, traffic_source = live, text
            elif traffic_source is not None:
                traffic_source = "not measured by this run (%s); " % text + traffic_source
        out["roofline"] = {
            "bound": "valu_fp32", "achieved": round(achieved, 3), "peak": PEAK_FP32_VALU_TFLOPS, "unit": "TFLOP/s",
            "frac": round(achieved / PEAK_FP32_VALU_TFLOPS, 4),
            "frac_reference_formulation": round(flops_ref / avg_s / 1e12 / PEAK_FP32_VALU_TFLOPS, 4),
            "traffic": traffic_bytes,
            "traffic_source": traffic_source,
            "kernel_ms": round(avg_s * 1e3, 3), "flops_per_launch": flops, "flops_per_launch_reference_formulation": flops_ref,
            "rays_traced": rays, "rays_reference_formulation": rays_ref, "paths_full": full, "paths_early": early,
            "flops_per_ray": f_ray, "cull_tests": n_cull, "exact_triangle_tests_after_cull": n_exact,
            "note": ("frac prices the tests the timed launch ran (rays traced x per-primitive flops of the reference's tests; for "
                     "kernels with a cheap conservative test (bounding sphere, plane, box of a hierarchy) in front of the triangle test: cull tests x %s + triangle tests "
                     "actually run x 113); frac_reference_formulation prices every ray and every (ray, primitive) pair of "
                     "the reference's formulation, which a kernel that culls does not execute -- it may exceed 1" % ("n/a" if f_cull is None else "%.1f" % f_cull)),
        }
        executed = recorded("pmc_summary.json", "%s/%s" % (args.workload, variant_name))
        if executed is not None:
            out["roofline"]["executed"] = executed       # hardware view of the same launch, from committed PMC passes
        if world == 1 and not args.no_traffic_pass:
            live, why = measured_executed(workload_args)
            if live is not None:
                if executed is not None:
                    live["committed_passes"] = executed                  # the fuller set of counters, from an earlier session
                out["roofline"]["executed"] = live
            elif executed is not None:
                executed["not_measured_by_this_run"] = why
        # the same launch against the HBM roofline, to show that it is not the one that binds:
        # 16 B accumulator read + 16 B write + 4 B seed word per pixel per launch, whatever spp is
        hbm = W * H * 36 / avg_s / 1e9
        out["roofline_hbm"] = {"bound": "hbm", "achieved": round(hbm, 3), "peak": 8000.0, "unit": "GB/s",
                               "frac": round(hbm / 8000.0, 6), "traffic": out["roofline"]["traffic"],
                               "algorithmic_bytes": W * H * 36}
        # the same frame with the reference's sequential generator and running-mean blend (one launch, outside the timed
        # region): samples are taken in index order there, so the wavefronts pay for roulette divergence
        if not args.no_stream_line and world == 1:
            ctx.set_params(bdpt.default_params(kernel_variant=args.kernel))
            names = {bdpt.PATHS_PER_LANE: "per_lane", bdpt.PATHS_POOLED: "pooled_by_phase", bdpt.PATHS_BY_CLASS: "by_class"}

            def stream_launch(sched):
                """one warm-up and one measured launch of the frame in the reference's mode; None if it ran another way"""
                ctx.set_path_scheduling(sched)
                ctx.set_camera(scene.camera)
                ctx.render(SPP)
                ctx.num_samples = 0
                ctx.render(SPP)
                ran = ctx.last_launch_path_scheduling()
                if sched != bdpt.PATHS_AUTO and ran != sched:
                    return None
                return {"kernel_ms": round(ctx.last_launch_ms(), 3),
                        "mpaths_per_s": round(W * H * SPP / ctx.last_launch_ms() / 1e3, 2),
                        "path_scheduling": names[ran], "runs_per_tile": ctx.last_launch_chunks()}

            # as the library runs it when left to itself (pixels of a wavefront sorted by the roulette class of their next
            # sample where the launch allows it), then with each way of putting paths on lanes asked for by name
            line = stream_launch(bdpt.PATHS_AUTO)
            line["what"] = "rng_mode STREAM + accum_mode MEAN: the reference's own generator and running mean, one launch, path scheduling left to the library"
            for sched in (bdpt.PATHS_PER_LANE, bdpt.PATHS_BY_CLASS, bdpt.PATHS_POOLED):
                sub = stream_launch(sched)
                if sub is not None:
                    del sub["path_scheduling"]
                    line[names[sched]] = sub
            ctx.set_path_scheduling(bdpt.PATHS_AUTO)
            out["stream_mode"] = line
        # the many-triangle scene through the exhaustive search as well (every ray meets every triangle's bounding
        # sphere: what this configuration was defined on), beside the hierarchy walk the library chooses by itself
        if args.workload == "c4" and world == 1 and variant == bdpt.KERNEL_BVH:
            ctx.set_params(bdpt.default_params(rng_mode=bdpt.RNG_STRIDED, accum_mode=bdpt.ACCUM_SUM, kernel_variant=bdpt.KERNEL_TILED, sample_order=order))
            ctx.set_camera(scene.camera)
            ctx.render(SPP)
            out["exhaustive_search"] = {"kernel_variant": "TILED", "kernel_ms": round(ctx.last_launch_ms(), 3),
                                        "mpaths_per_s": round(W * H * SPP / ctx.last_launch_ms() / 1e3, 2),
                                        "what": "the same launch with every triangle's bounding sphere tested for every ray (kernel_variant TILED), one launch"}
            ctx.set_params(bench_params)
            ctx.set_camera(scene.camera)
            ctx.set_params(bench_params)

    # The second half of the headline metric: wall time from a camera reset until the accumulated image is within
    # `rmse_target` (RMSE on linear RGB clamped to [0,1]) of a converged reference -- a `rmse_ref_spp` render of the
    # reference's own sequence (STREAM + MEAN) with other seeds -- checked after every 64-sample launch.  Outside the
    # timed steps.  The time includes the RMSE evaluation on the device after every launch.
    if rank == 0 and world == 1 and args.workload == "c2" and not args.no_time_to_rmse:
        ctx.set_params(bench_params)
        ctx.bind_accum(accum.data_ptr())
        ctx.follow_stream(torch.cuda.current_stream().cuda_stream)
        with bdpt.Context(W, H, device=local_rank) as rctx:
            ref_buf = torch.zeros((H, W, 4), dtype=torch.float32, device="cuda")
            rctx.upload_scene(scene)
            rctx.upload_seeds(bdpt.make_seeds(W, H, 0xBEEF))
            rctx.set_camera(scene.camera)
            rctx.bind_accum(ref_buf.data_ptr())
            rctx.render(args.rmse_ref_spp)
            ref = ref_buf[..., :3].clamp(0, 1).clone()
        del ref_buf
        chunk, max_spp = 64, 8192
        total = torch.zeros_like(accum)
        ctx.set_camera(scene.camera)
        ctx.set_sample_offset(0)
        ctx.render(chunk)                         # warm-up of this launch size and of the RMSE evaluation's device code
        total += accum
        torch.sqrt(torch.mean(((total[..., :3] / chunk).clamp(0, 1) - ref) ** 2)).item()
        total.zero_()
        ctx.set_camera(scene.camera)              # the reset the clock starts at
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        done, hit, curve = 0, None, []
        while done < max_spp:
            ctx.set_sample_offset(done)
            ctx.num_samples = 0
            ctx.render_async(chunk)
            done += chunk
            total += accum
            err = torch.sqrt(torch.mean(((total[..., :3] / done).clamp(0, 1) - ref) ** 2)).item()
            t = time.perf_counter() - t0
            curve.append([done, round(t * 1e3, 2), round(err, 5)])
            if err < args.rmse_target:
                hit = (done, t, err)
                break
        out["time_to_rmse"] = {
            "seconds": None if hit is None else round(hit[1], 4), "spp_at_target": None if hit is None else hit[0],
            "rmse_at_target": None if hit is None else round(hit[2], 5), "bound": args.rmse_target,
            "reference_spp": args.rmse_ref_spp, "reference": "STREAM + MEAN render of the same frame with other seeds",
            "chunk_spp": chunk, "falloff": "1/r^2 (kernel0.cl:719)", "n_gpus": 1,
            "curve_spp_ms_rmse": curve[:: max(1, len(curve) // 8)] + curve[-1:],
        }
        ctx.use_own_stream()
        ctx.bind_accum(None)
        ctx.set_sample_offset(0)

    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        sys.path.insert(0, os.path.join(ROOT, "oracle"))
        import oracle_bind as oracle   # the checker, timed here as the CPU baseline

        threads = min(os.cpu_count() or 1, int(os.environ.get("OMP_NUM_THREADS", "16")), 16)
        # a sample with the frame's own mix of paths: a 6 x 4 grid of blocks spread evenly over the frame
        gx, gy, bw, bh, cspp = 6, 4, min(W // 6, 80), min(H // 4, 60), 1024
        st = oracle.State(W, H, bdpt.make_seeds(W, H, 0x5EED))
        p = bdpt.default_params(rng_mode=bdpt.RNG_STRIDED, accum_mode=bdpt.ACCUM_SUM)
        if args.workload == "c4":
            gx, gy, bw, bh, cspp = 4, 3, 16, 8, 1
        t0 = time.perf_counter()
        for j in range(gy):
            for i in range(gx):
                x0 = (W // gx) * i + (W // gx - bw) // 2
                y0 = (H // gy) * j + (H // gy - bh) // 2
                st.num_samples = 0
                oracle.render(st, scene, scene.camera, p, cspp, region=(x0, y0, bw, bh), n_threads=threads)
        cpu_s = time.perf_counter() - t0
        n_paths = gx * gy * bw * bh * cspp
        out["cpu_baseline"] = {
            "value": round(n_paths / cpu_s / 1e6, 6), "unit": "Mpaths/s", "cores": threads, "kind": "port",
            "sample": "%d x %d blocks of %dx%d pixels spread evenly over the %dx%d frame, %d spp, same scene/seeds/mode; %.1f s"
                      % (gx, gy, bw, bh, W, H, cspp, cpu_s),
        }
        # the reference's own kernel on a CPU OpenCL device of this host, when there is one: then THAT is the baseline
        # (kind "reference") and the oracle's figure is kept beside it
        lines, reason = reference_on_cpu_opencl(scene, scene.camera, bdpt.make_seeds(W, H, 0x5EED), W, H)
        if lines:
            port = out["cpu_baseline"]
            first = lines[0]
            out["cpu_baseline"] = {
                "value": round(first["mpaths_per_s"], 6), "unit": "Mpaths/s", "cores": first["compute_units"], "kind": "reference",
                "sample": "the reference's kernel file built with %s on OpenCL device %r (%d compute units, nproc %d): %d frames of %dx%d, 1 spp each, after 2 warm-up frames; %.1f s"
                          % (first["build_options"], first["device"], first["compute_units"], first["nproc"], first["frames"], W, H, first["seconds"]),
                "other_builds": lines[1:], "port": port,
            }
        else:
            out["cpu_baseline"]["reference_opencl_on_cpu"] = reason

    if rank == 0:
        print(json.dumps(out), flush=True)
    ctx.close()
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()

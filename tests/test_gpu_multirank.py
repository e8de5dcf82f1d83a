"""The N > 1 paths with real GPU contexts: two ranks on the one GPU of the test box, gloo for the exchange.
The same dist_render functions drive bench.py and tools/interactive.py over RCCL on a multi-GPU node."""
import json
import os
import socket
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
W, H, SPP = 200, 120, 24          # not a multiple of the 64x32 tiles: ragged right and bottom edges


def _worker(rank, world, port, mode, out_path):
    sys.path.insert(0, os.path.join(ROOT, "opencl-path-tracer_amd"))
    import torch
    import torch.distributed as dist

    import bdpt
    import dist_render as dr

    dist.init_process_group("gloo", init_method="tcp://127.0.0.1:%d" % port, rank=rank, world_size=world)
    torch.cuda.set_device(0)
    scene = bdpt.Scene.cornell()
    image = torch.zeros((H, W, 4), dtype=torch.float32, device="cuda")
    with bdpt.Context(W, H, device=0) as ctx:
        if mode == "samples":
            ctx.set_params(bdpt.default_params(rng_mode=bdpt.RNG_STRIDED, accum_mode=bdpt.ACCUM_SUM))
        ctx.upload_scene(scene)
        ctx.upload_seeds(bdpt.make_seeds(W, H, 0x5EED))
        ctx.set_camera(scene.camera)
        ctx.bind_accum(image.data_ptr())
        ctx.follow_stream(torch.cuda.current_stream().cuda_stream)
        if mode == "samples":
            dr.render_sample_split(ctx, SPP, world, rank, wait=False)
            img = dr.combine_sample_split(image, SPP, dist)
        else:
            for _ in range(3):                  # three progressive frames, gathered every frame
                dr.render_tile_split(ctx, H, W, 2, world, rank, wait=False)
                img = dr.gather_tile_split(image, H, world, dist)
        torch.cuda.synchronize()
    if rank == 0:
        np.save(out_path, img.cpu().numpy())
    else:
        assert img is None
    dist.barrier()
    dist.destroy_process_group()


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _run(mode, tmp_path, world=2):
    import torch.multiprocessing as mp

    out = str(tmp_path / ("%s.npy" % mode))
    mp.spawn(_worker, args=(world, _free_port(), mode, out), nprocs=world, join=True)
    return np.load(out)


def _single(bdpt, spp, **params):
    scene = bdpt.Scene.cornell()
    with bdpt.Context(W, H) as ctx:
        ctx.set_params(bdpt.default_params(**params))
        ctx.upload_scene(scene)
        ctx.upload_seeds(bdpt.make_seeds(W, H, 0x5EED))
        ctx.set_camera(scene.camera)
        for n in spp:
            ctx.render(n)
        return ctx.read_accum()


@pytest.mark.timeout(300)
def test_two_ranks_sample_split_equals_one_rank_rendering_both_ranges(bdpt, tmp_path):
    got = _run("samples", tmp_path)
    want = _single(bdpt, [SPP], rng_mode=bdpt.RNG_STRIDED, accum_mode=bdpt.ACCUM_SUM) / SPP
    np.testing.assert_allclose(got, want, rtol=1e-6, atol=1e-6)      # the same samples, added in another order
    assert got.max() > 0.1


@pytest.mark.timeout(300)
def test_three_ranks_tile_split_is_bit_identical_to_one_rank(bdpt, tmp_path):
    got = _run("tiles", tmp_path, world=3)
    want = _single(bdpt, [2, 2, 2])
    assert np.array_equal(got.view(np.uint32), want.view(np.uint32))


def test_eight_interleaved_shares_make_the_whole_frame(bdpt):
    """Eight contexts, share r of 8 each (tiles r, r+8, ... of the 64x32 tiling), STREAM mode, parity buffers on:
    together they are the frame one context renders, bit for bit, and no share touches another's pixels."""
    import dist_render as dr

    scene = bdpt.Scene.cornell()
    seeds = bdpt.make_seeds(W, H, 7)

    def render(stride, phase):
        with bdpt.Context(W, H) as ctx:
            ctx.set_params(bdpt.default_params(aov_enable=1))
            ctx.upload_scene(scene)
            ctx.upload_seeds(seeds)
            ctx.set_camera(scene.camera)
            ctx.set_tile_interleave(64, 32, stride, phase)
            ctx.render(3)
            ctx.render(2)
            return ctx.read_accum(), ctx.read_aov(bdpt.AOV_HIT_HASH), ctx.read_seeds()

    whole = render(1, 0)
    owner = dr.tile_owner_map(H, W, 8)
    xs, ys = np.meshgrid(np.arange(W), np.arange(H))
    seed_index = (xs * H + ys) * 4 + 3
    for r in range(8):
        acc, hsh, sd = render(8, r)
        mine = owner == r
        assert np.array_equal(acc[mine].view(np.uint32), whole[0][mine].view(np.uint32))
        assert np.array_equal(hsh[mine], whole[1][mine])
        assert np.array_equal(sd[seed_index[mine]], whole[2][seed_index[mine]])
        assert not acc[~mine].any() and not hsh[~mine].any()
        assert np.array_equal(sd[seed_index[~mine]], seeds[seed_index[~mine]])
    with bdpt.Context(W, H) as ctx:
        for bad in ((60, 32, 2, 0), (64, 12, 2, 0), (64, 32, 2, 2), (64, 32, 0, 0)):
            with pytest.raises(bdpt.BdptError):
                ctx.set_tile_interleave(*bad)


def _read_pfm(path):
    raw = open(path, "rb").read()
    head = raw.split(b"\n", 3)
    w, h = (int(v) for v in head[1].split())
    return np.frombuffer(head[3], "<f4").reshape(h, w, 3)


def test_device_group_in_one_process(bdpt, tmp_path):
    """bdpt_group_*: several contexts behind one handle in one process.  On this one-GPU box the group names GPU 0
    three times (partial sums combined by device-to-device copies) and once (combined by a one-rank RCCL reduce,
    which exercises the run-time lookup of RCCL and the reduce call): three progressive frames give the mean of the
    same 20 samples one context renders alone, up to the order of additions; a camera change starts over."""
    scene = bdpt.Scene.cornell()
    seeds = bdpt.make_seeds(W, H, 0x5EED)
    want = _single(bdpt, [20], rng_mode=bdpt.RNG_STRIDED, accum_mode=bdpt.ACCUM_SUM) / 20

    def run(devices, combine):
        with bdpt.Group(devices, W, H, combine) as g:
            assert g.size == len(devices)
            g.upload_scene(scene)
            g.upload_seeds(seeds)
            g.set_camera(scene.camera)
            g.render(2)                     # fewer samples than members: one member has nothing to add yet
            first = g.read_image()
            g.render(11)
            g.render(7)
            assert g.num_samples == 20
            img = g.read_image()
            g.set_camera(scene.camera)
            assert g.num_samples == 0
            g.render(2)
            again = g.read_image()
            with pytest.raises(bdpt.BdptError, match="STRIDED"):
                g.set_params(bdpt.default_params())
            return first, img, again, g.combine

    first, img, again, how = run([0, 0, 0], bdpt.COMBINE_AUTO)
    assert how == bdpt.COMBINE_COPY
    np.testing.assert_allclose(img, want, rtol=1e-6, atol=1e-6)
    np.testing.assert_allclose(first, _single(bdpt, [2], rng_mode=bdpt.RNG_STRIDED, accum_mode=bdpt.ACCUM_SUM) / 2, rtol=1e-6, atol=1e-6)
    assert np.array_equal(first, again)
    first1, img1, _, how1 = run([0], bdpt.COMBINE_RCCL)
    assert how1 == bdpt.COMBINE_RCCL
    np.testing.assert_allclose(img1, want, rtol=1e-6, atol=1e-6)
    with pytest.raises(bdpt.BdptError, match="distinct"):
        bdpt.Group([0, 0], W, H, bdpt.COMBINE_RCCL)
    # one sample per frame (the C++ host's default): the frames go round the members, none idles
    with bdpt.Group([0, 0, 0], W, H, bdpt.COMBINE_AUTO) as g:
        g.upload_scene(scene)
        g.upload_seeds(seeds)
        g.set_camera(scene.camera)
        for _ in range(7):
            g.render(1)
        assert [g.member_num_samples(i) for i in range(3)] == [3, 2, 2]
        g.render(5)                                    # 1 + 2 left over, starting at member 1
        assert [g.member_num_samples(i) for i in range(3)] == [4, 4, 4]
        np.testing.assert_allclose(g.read_image(), _single(bdpt, [12], rng_mode=bdpt.RNG_STRIDED, accum_mode=bdpt.ACCUM_SUM) / 12, rtol=1e-6, atol=1e-6)

    # the C++ host over the same entry points
    exe = os.path.join(ROOT, "opencl-path-tracer_amd", "bdpt_host")
    outs = []
    for devs in ("0,0", "0"):
        out = str(tmp_path / ("g%d.pfm" % len(outs)))
        r = subprocess.run([exe, "--devices", devs, "--width", "96", "--height", "64", "--frames", "3", "--spp", "6", "--out", out,
                            "--quiet"], capture_output=True, text=True, timeout=120)
        assert r.returncode == 0, r.stdout + r.stderr
        assert "18 samples accumulated" in r.stdout and "devices" in r.stdout
        outs.append(_read_pfm(out))
    np.testing.assert_allclose(outs[0], outs[1], rtol=1e-6, atol=1e-6)
    assert outs[0].max() > 0.1


@pytest.mark.timeout(600)
def test_bench_starts_its_own_ranks(tmp_path):
    """`python bench.py --gpus 2` without a launcher around it: the bench starts two ranks itself (before it touches the
    GPU), they share the one GPU here and combine over gloo; one JSON line, and the combined image is the mean of the
    128 samples one context renders alone; the strong-scaling steps that follow render 64 samples in all, 32 per rank."""
    sys.path.insert(0, os.path.join(ROOT, "opencl-path-tracer_amd"))
    import bdpt

    dump = str(tmp_path / "img.npy")
    dump_strong = str(tmp_path / "img_strong.npy")
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--backend", "gloo", "--spp", "64",
                        "--width", "320", "--height", "180", "--steps", "2", "--warmup", "1", "--dump-image", dump, "--dump-strong-image", dump_strong],
                       capture_output=True, text=True, timeout=500, env=env)
    assert r.returncode == 0, r.stdout + r.stderr
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["unit"] == "Mpaths/s" and out["value"] > 0
    assert "gloo" in out["config"]["combine"] and "RCCL" not in out["config"]["combine"]
    assert len(out["per_rank_kernel_ms"]) == 2
    got = np.load(dump)
    scene = bdpt.Scene.cornell()
    with bdpt.Context(320, 180) as ctx:
        ctx.set_params(bdpt.default_params(rng_mode=bdpt.RNG_STRIDED, accum_mode=bdpt.ACCUM_SUM))
        ctx.upload_scene(scene)
        ctx.upload_seeds(bdpt.make_seeds(320, 180, 0x5EED))
        ctx.set_camera(scene.camera)
        ctx.render(128)
        want = ctx.read_accum() / 128
    np.testing.assert_allclose(got, want, rtol=1e-6, atol=2e-6)
    # the strong-scaling steps: the 64 samples of one GPU's step dealt to the two ranks
    ss = out["strong_scaling"]
    assert ss["total_spp"] == 64 and ss["spp_per_rank"] == [32, 32] and len(ss["per_rank_kernel_ms"]) == 2
    assert ss["ms_per_step"] > 0 and abs(ss["mpaths_per_s"] - 320 * 180 * 64 / ss["ms_per_step"] / 1e3) < 0.01 * ss["mpaths_per_s"]
    with bdpt.Context(320, 180) as ctx:
        ctx.set_params(bdpt.default_params(rng_mode=bdpt.RNG_STRIDED, accum_mode=bdpt.ACCUM_SUM))
        ctx.upload_scene(scene)
        ctx.upload_seeds(bdpt.make_seeds(320, 180, 0x5EED))
        ctx.set_camera(scene.camera)
        ctx.render(64)
        want = ctx.read_accum() / 64
    np.testing.assert_allclose(np.load(dump_strong), want, rtol=1e-6, atol=2e-6)

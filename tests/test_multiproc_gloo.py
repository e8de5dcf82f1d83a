"""world_size-2 runs of the multi-GPU driver logic (dist_render.py) over gloo on CPU.
The renderer behind the protocol is the CPU oracle standing in for a GPU context;
the partitioning, offsets and the exchange are the code bench.py runs over RCCL."""
import os
import socket
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
W, H, SPP = 40, 24, 10


class OracleRenderer:
    """Implements the renderer protocol of dist_render with the oracle."""

    def __init__(self, bdpt, oracle, scene, params, seeds):
        self.bdpt, self.oracle, self.scene, self.params = bdpt, oracle, scene, params
        self.state = oracle.State(W, H, seeds)
        self.offset, self.region, self.tiles = 0, None, (0, 0, 1, 0)

    def set_sample_offset(self, first):
        self.offset = first

    def set_region(self, x0, y0, w, h):
        self.region = (x0, y0, w, h)

    def set_tile_interleave(self, tile_w, tile_h, stride, phase):
        self.tiles = (tile_w, tile_h, stride, phase)

    @property
    def num_samples(self):
        return self.state.num_samples

    @num_samples.setter
    def num_samples(self, n):
        self.state.num_samples = n

    def render(self, spp):
        tw, th, stride, phase = self.tiles
        if stride <= 1:
            self.oracle.render(self.state, self.scene, self.scene.camera, self.params, spp, region=self.region,
                               sample_offset=self.offset, n_threads=2)
            return
        # the oracle renders rectangles: one call per owned tile, each from the same sample counter
        x0, y0, w, h = self.region if self.region is not None else (0, 0, W, H)
        tiles_x = (w + tw - 1) // tw
        n0 = self.state.num_samples
        for t in range(phase, tiles_x * ((h + th - 1) // th), stride):
            tx, ty = x0 + (t % tiles_x) * tw, y0 + (t // tiles_x) * th
            self.state.num_samples = n0
            self.oracle.render(self.state, self.scene, self.scene.camera, self.params, spp,
                               region=(tx, ty, min(tw, x0 + w - tx), min(th, y0 + h - ty)), sample_offset=self.offset, n_threads=2)
        self.state.num_samples = n0 + spp

    render_async = render

    def clear_accum(self):
        self.state.accum[...] = 0.0


def _worker(rank, world, port, mode, out_path):
    for p in (os.path.join(ROOT, "opencl-path-tracer_amd"), os.path.join(ROOT, "oracle")):
        sys.path.insert(0, p)
    import torch
    import torch.distributed as dist

    import bdpt
    import dist_render as dr
    import oracle_bind as oracle

    dist.init_process_group("gloo", init_method="tcp://127.0.0.1:%d" % port, rank=rank, world_size=world)
    scene = bdpt.Scene.cornell()
    seeds = bdpt.make_seeds(W, H, 0x5EED)
    if mode in ("samples", "samples_few"):
        total = SPP if mode == "samples" else 1          # one sample for two ranks: rank 1 has nothing to render
        p = bdpt.default_params(rng_mode=bdpt.RNG_STRIDED, accum_mode=bdpt.ACCUM_SUM)
        r = OracleRenderer(bdpt, oracle, scene, p, seeds)
        r.state.accum[...] = 7.0                          # what an earlier frame left behind
        dr.render_sample_split(r, total, world, rank)
        t = torch.from_numpy(r.state.accum)
        img = dr.combine_sample_split(t, total, dist)
    else:
        p = bdpt.default_params()
        r = OracleRenderer(bdpt, oracle, scene, p, seeds)
        dr.render_tile_split(r, H, W, SPP, world, rank, tile=(16, 8))
        img = dr.gather_tile_split(torch.from_numpy(r.state.accum), H, world, dist, tile=(16, 8))
    if rank == 0:
        np.save(out_path, img.numpy())
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


def _run(mode, tmp_path):
    import torch.multiprocessing as mp

    out = str(tmp_path / ("%s.npy" % mode))
    mp.spawn(_worker, args=(2, _free_port(), mode, out), nprocs=2, join=True)
    return np.load(out)


@pytest.mark.timeout(300)
def test_sample_split_reduce_equals_single_rank(bdpt, oracle, tmp_path):
    got = _run("samples", tmp_path)
    scene = bdpt.Scene.cornell()
    p = bdpt.default_params(rng_mode=bdpt.RNG_STRIDED, accum_mode=bdpt.ACCUM_SUM)
    st = oracle.State(W, H, bdpt.make_seeds(W, H, 0x5EED))
    oracle.render(st, scene, scene.camera, p, SPP, n_threads=4)
    # the same samples summed in a different order: fp32 re-association noise only
    np.testing.assert_allclose(got, st.accum / SPP, rtol=1e-5, atol=1e-6)
    assert got.max() > 0


@pytest.mark.timeout(300)
def test_sample_split_with_fewer_samples_than_ranks(bdpt, oracle, tmp_path):
    """A rank without a share contributes zeros, not what its accumulator held before."""
    got = _run("samples_few", tmp_path)
    scene = bdpt.Scene.cornell()
    p = bdpt.default_params(rng_mode=bdpt.RNG_STRIDED, accum_mode=bdpt.ACCUM_SUM)
    st = oracle.State(W, H, bdpt.make_seeds(W, H, 0x5EED))
    oracle.render(st, scene, scene.camera, p, 1, n_threads=4)
    assert np.array_equal(got.view(np.uint32), st.accum.view(np.uint32))


@pytest.mark.timeout(300)
def test_tile_split_gather_is_bit_identical_to_single_rank(bdpt, oracle, tmp_path):
    got = _run("tiles", tmp_path)
    scene = bdpt.Scene.cornell()
    st = oracle.State(W, H, bdpt.make_seeds(W, H, 0x5EED))
    oracle.render(st, scene, scene.camera, bdpt.default_params(), SPP, n_threads=4)
    assert np.array_equal(got.view(np.uint32), st.accum.view(np.uint32))

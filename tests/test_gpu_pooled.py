"""Pooled path scheduling (bdpt_set_path_scheduling, csrc/bdpt_pool.h): a wavefront owns more pixels than lanes and
runs, every round, one phase of up to 64 of their paths.  Which lane works on a pixel must not show anywhere: the bars
are those of tests/test_gpu_parity.py -- ids, hash chains, draw and ray counts, generator state, radiance and
accumulator all bit-exact against the CPU oracle -- in the reference's own mode (sequential generator, running mean,
kernel0.cl:853-856, early returns :645-656) and in every other mode of the library."""
import numpy as np
import pytest

from test_gpu_parity import _compare, _pair

pytestmark = pytest.mark.gpu

LDS, LDS2 = 3, 5


def _pooled_pair(bdpt, oracle, scene, W, H, **kw):
    ctx, st, p = _pair(bdpt, oracle, scene, W, H, **kw)
    ctx.set_path_scheduling(bdpt.PATHS_POOLED)
    return ctx, st, p


@pytest.mark.parametrize("variant", [LDS, LDS2])
@pytest.mark.parametrize("chunks", [1, 3])
def test_pooled_stream_mode_matches_the_oracle(bdpt, oracle, variant, chunks):
    """Cornell box on a frame that is no multiple of the pool's 16 x 6 pixels, the reference's mode, several launches:
    everything the oracle computes, after every launch, unsliced and as three runs per tile."""
    scene = bdpt.Scene.cornell()
    W, H = 70, 45
    ctx, st, p = _pooled_pair(bdpt, oracle, scene, W, H, seed=0x51, kernel_variant=variant)
    ctx.set_launch_chunks(chunks)
    for spp in (1, 7, 12):
        ctx.render(spp)
        assert ctx.last_launch_path_scheduling() == bdpt.PATHS_POOLED
        oracle.render(st, scene, scene.camera, p, spp, n_threads=8)
        _compare(bdpt, ctx, st, "after %d more samples:" % spp)
    ctx.close()


@pytest.mark.parametrize("mode", ["strided_mean", "strided_sum_index", "strided_sum_class", "strided_sum_adaptive"])
def test_pooled_strided_modes_match_the_oracle(bdpt, oracle, mode):
    kw = {"strided_mean": dict(rng_mode=1), "strided_sum_index": dict(rng_mode=1, accum_mode=1, sample_order=1),
          "strided_sum_class": dict(rng_mode=1, accum_mode=1, sample_order=2),
          "strided_sum_adaptive": dict(rng_mode=1, accum_mode=1, sample_order=3)}[mode]
    scene = bdpt.Scene.cornell()
    W, H = 48, 40
    ctx, st, p = _pooled_pair(bdpt, oracle, scene, W, H, seed=0x52, kernel_variant=LDS2, **kw)
    ctx.set_launch_chunks(2)
    for spp in (40, 9):
        ctx.render(spp)
        assert ctx.last_launch_path_scheduling() == bdpt.PATHS_POOLED
        oracle.render(st, scene, scene.camera, p, spp, n_threads=8)
        _compare(bdpt, ctx, st, mode + ":")
    ctx.close()


@pytest.mark.parametrize("scene_name", ["outdoor", "cornell_planes"])
def test_pooled_other_scenes(bdpt, oracle, scene_name):
    """Spheres under an open sky (most eye rays leave the scene: pixels drop out of the pool at once) and the box made
    of planes, LDS kernel."""
    scene = getattr(bdpt.Scene, scene_name)()
    W, H = 80, 42
    ctx, st, p = _pooled_pair(bdpt, oracle, scene, W, H, seed=0x53, kernel_variant=LDS)
    ctx.render(10)
    assert ctx.last_launch_path_scheduling() == bdpt.PATHS_POOLED
    oracle.render(st, scene, scene.camera, p, 10, n_threads=8)
    _compare(bdpt, ctx, st, scene_name + ":")
    ctx.close()


def test_pooled_switches_and_long_subpaths(bdpt, oracle):
    """The quirk switches and loop constants: no skipped light primitive, bounded falloff, indirect only, up to seven
    light-subpath vertices with a roulette that nearly always continues."""
    scene = bdpt.Scene.cornell()
    W, H = 40, 30
    for kw in (dict(light_origin=1), dict(connection_falloff=1, falloff_c=0.5), dict(indirect_only=1),
               dict(max_light_bounces=7, p_continue=0.95), dict(max_light_bounces=1), dict(max_light_bounces=12, p_continue=0.9)):
        ctx, st, p = _pooled_pair(bdpt, oracle, scene, W, H, seed=0x54, kernel_variant=LDS2, **kw)
        ctx.render(6)
        assert ctx.last_launch_path_scheduling() == bdpt.PATHS_POOLED
        oracle.render(st, scene, scene.camera, p, 6, n_threads=8)
        _compare(bdpt, ctx, st, "%r:" % (kw,))
        ctx.close()


def test_pooled_region_and_counters(bdpt, oracle):
    """A region that starts and ends off the pool grid; the work counters of a counting launch equal the per-lane
    launch's (the same rays are traced, only by other lanes)."""
    scene = bdpt.Scene.cornell()
    W, H = 96, 64
    seeds = bdpt.make_seeds(W, H, 0x55)
    out = {}
    for sched in (bdpt.PATHS_PER_LANE, bdpt.PATHS_POOLED):
        with bdpt.Context(W, H) as ctx:
            ctx.set_params(bdpt.default_params(kernel_variant=LDS2))
            ctx.upload_scene(scene); ctx.upload_seeds(seeds); ctx.set_camera(scene.camera)
            ctx.set_path_scheduling(sched)
            ctx.set_region(13, 7, 50, 41)
            ctx.enable_counters(True)
            ctx.render(9)
            assert ctx.last_launch_path_scheduling() == sched
            out[sched] = (ctx.read_accum(), ctx.read_seeds(), ctx.read_counters())
    a, b = out[bdpt.PATHS_PER_LANE], out[bdpt.PATHS_POOLED]
    assert np.array_equal(a[0].view(np.uint32), b[0].view(np.uint32))
    assert np.array_equal(a[1], b[1])
    assert list(a[2]) == list(b[2])
    assert np.count_nonzero(a[0][:7]) == 0 and np.count_nonzero(a[0][:, :13]) == 0      # outside the region nothing is written


def test_launches_that_cannot_be_pooled_run_per_lane(bdpt):
    """Other samplers, the tiled and direct kernels, the interleaved tile split and the old integrator are per-lane
    launches whatever is asked; the call says which it was."""
    scene = bdpt.Scene.cornell()
    W, H = 64, 32
    for kw, split in ((dict(sampler=bdpt.SAMPLER_HEMISPHERE), False), (dict(kernel_variant=bdpt.KERNEL_TILED), False),
                      (dict(kernel_variant=bdpt.KERNEL_DIRECT), False), (dict(), True), (dict(integrator=1), False)):
        with bdpt.Context(W, H) as ctx:
            ctx.set_params(bdpt.default_params(**kw))
            ctx.upload_scene(scene); ctx.upload_seeds(bdpt.make_seeds(W, H, 1)); ctx.set_camera(scene.camera)
            ctx.set_path_scheduling(bdpt.PATHS_POOLED)
            if split:
                ctx.set_tile_interleave(16, 8, 2, 1)
            ctx.render(2)
            assert ctx.last_launch_path_scheduling() == bdpt.PATHS_PER_LANE
    with bdpt.Context(W, H) as ctx:
        with pytest.raises(bdpt.BdptError, match="path_scheduling"):
            ctx.set_path_scheduling(7)


def test_pooled_full_size_frame_equals_per_lane(bdpt):
    """1920 x 1080 x 64 samples in the reference's mode, automatic slicing: every bit of the image and of the generator
    state as from the per-lane launch."""
    scene = bdpt.Scene.cornell()
    W, H = 1920, 1080
    seeds = bdpt.make_seeds(W, H, 0x5EED)
    out = []
    for sched in (bdpt.PATHS_PER_LANE, bdpt.PATHS_POOLED):
        with bdpt.Context(W, H) as ctx:
            ctx.set_params(bdpt.default_params())
            ctx.upload_scene(scene); ctx.upload_seeds(seeds); ctx.set_camera(scene.camera)
            ctx.set_path_scheduling(sched)
            ctx.set_launch_chunks(0)
            ctx.render(64)
            assert ctx.last_launch_path_scheduling() == sched
            out.append((ctx.read_accum(), ctx.read_seeds()))
    assert np.array_equal(out[0][0].view(np.uint32), out[1][0].view(np.uint32))
    assert np.array_equal(out[0][1], out[1][1])

"""Pixels sorted by the length of their next sample (BDPT_PATHS_BY_CLASS, csrc/bdpt_class_pool.h): a wavefront owns
128 pixels and runs, every round, one whole sample for up to 64 of them whose light subpaths Russian roulette will cut
to about the same length.  Launches that take a pixel's samples in index order run this way (on request, or by
themselves from 32 samples per pixel on).  Which lane works on a pixel, and in which order different pixels advance,
must not show anywhere: the bars are those of tests/test_gpu_parity.py -- ids, hash chains, draw and ray counts,
generator state, radiance and accumulator all bit-exact against the CPU oracle -- in the reference's own mode
(sequential generator, running mean, kernel0.cl:853-856, early returns :645-656) and with the strided generator."""
import numpy as np
import pytest

from test_gpu_parity import _compare, _pair

pytestmark = pytest.mark.gpu

DIRECT, LDS, LDS2, BVH = 1, 3, 5, 6


def _by_class_pair(bdpt, oracle, scene, W, H, **kw):
    ctx, st, p = _pair(bdpt, oracle, scene, W, H, **kw)
    ctx.set_path_scheduling(bdpt.PATHS_BY_CLASS)
    return ctx, st, p


@pytest.mark.parametrize("variant", [DIRECT, LDS, LDS2, BVH])
@pytest.mark.parametrize("chunks", [1, 3, 0])
def test_by_class_stream_mode_matches_the_oracle(bdpt, oracle, variant, chunks):
    """Cornell box on a frame that is no multiple of a wavefront's 16 x 8 pixels, the reference's mode, several
    launches: everything the oracle computes, after every launch; unsliced, as three runs per tile (a run ends when its
    share of the tile's samples is spent, wherever the pixels are) and with the library's own choice of runs."""
    scene = bdpt.Scene.cornell()
    W, H = 70, 45
    ctx, st, p = _by_class_pair(bdpt, oracle, scene, W, H, seed=0x61, kernel_variant=variant)
    ctx.set_launch_chunks(chunks)
    for spp in (1, 7, 45):
        ctx.render(spp)
        assert ctx.last_launch_path_scheduling() == bdpt.PATHS_BY_CLASS
        oracle.render(st, scene, scene.camera, p, spp, n_threads=8)
        _compare(bdpt, ctx, st, "after %d more samples:" % spp)
    ctx.close()


@pytest.mark.parametrize("mode", ["strided_mean", "strided_sum_index"])
@pytest.mark.parametrize("variant", [LDS, LDS2])
def test_by_class_strided_generator_matches_the_oracle(bdpt, oracle, mode, variant):
    kw = {"strided_mean": dict(rng_mode=1), "strided_sum_index": dict(rng_mode=1, accum_mode=1, sample_order=1)}[mode]
    scene = bdpt.Scene.cornell()
    W, H = 48, 40
    ctx, st, p = _by_class_pair(bdpt, oracle, scene, W, H, seed=0x62, kernel_variant=variant, **kw)
    ctx.set_launch_chunks(2)
    for spp in (40, 9):
        ctx.render(spp)
        assert ctx.last_launch_path_scheduling() == bdpt.PATHS_BY_CLASS
        oracle.render(st, scene, scene.camera, p, spp, n_threads=8)
        _compare(bdpt, ctx, st, mode + ":")
    ctx.close()


@pytest.mark.parametrize("scene_name", ["outdoor", "cornell_planes"])
@pytest.mark.parametrize("sampler", [0, 1, 2])
def test_by_class_other_scenes_and_samplers(bdpt, oracle, scene_name, sampler):
    """Spheres under an open sky (most eye rays leave the scene: in the reference's mode those pixels are done after one
    sample) and the box made of planes, every direction sampler."""
    scene = getattr(bdpt.Scene, scene_name)()
    W, H = 80, 42
    ctx, st, p = _by_class_pair(bdpt, oracle, scene, W, H, seed=0x63, kernel_variant=LDS, sampler=sampler)
    ctx.set_launch_chunks(2)
    ctx.render(10)
    assert ctx.last_launch_path_scheduling() == bdpt.PATHS_BY_CLASS
    oracle.render(st, scene, scene.camera, p, 10, n_threads=8)
    _compare(bdpt, ctx, st, "%s sampler %d:" % (scene_name, sampler))
    ctx.close()


def test_by_class_switches_and_subpath_lengths(bdpt, oracle):
    """The quirk switches and loop constants the kernel supports: no skipped light primitive, bounded falloff, indirect
    only, one to five light-subpath vertices, a roulette that nearly always continues and one that nearly always stops."""
    scene = bdpt.Scene.cornell()
    W, H = 40, 30
    for kw in (dict(light_origin=1), dict(connection_falloff=1, falloff_c=0.5), dict(indirect_only=1), dict(max_light_bounces=1),
               dict(max_light_bounces=3, p_continue=0.95), dict(max_light_bounces=5, p_continue=0.95), dict(p_continue=0.1)):
        ctx, st, p = _by_class_pair(bdpt, oracle, scene, W, H, seed=0x64, kernel_variant=LDS2, **kw)
        ctx.render(6)
        assert ctx.last_launch_path_scheduling() == bdpt.PATHS_BY_CLASS
        oracle.render(st, scene, scene.camera, p, 6, n_threads=8)
        _compare(bdpt, ctx, st, "%r:" % (kw,))
        ctx.close()


def test_by_class_region_and_counters(bdpt, oracle):
    """A region that starts and ends off the 16 x 8 grid.  The work counters of an unsliced counting launch equal the
    per-lane launch's (the same rays are traced, only by other lanes); in a sliced launch of the reference's mode the
    per-lane kernel traces the eye ray of a finished pixel (:645-656) once per run, this kernel once per launch, so only
    the rays counter differs, by less than pixels x runs."""
    scene = bdpt.Scene.cornell()
    W, H = 96, 64
    seeds = bdpt.make_seeds(W, H, 0x65)
    for chunks in (1, 4):
        out = {}
        for sched in (bdpt.PATHS_PER_LANE, bdpt.PATHS_BY_CLASS):
            with bdpt.Context(W, H) as ctx:
                ctx.set_params(bdpt.default_params(kernel_variant=LDS2))
                ctx.upload_scene(scene); ctx.upload_seeds(seeds); ctx.set_camera(scene.camera)
                ctx.set_path_scheduling(sched)
                ctx.set_launch_chunks(chunks)
                ctx.set_region(13, 7, 50, 41)
                ctx.enable_counters(True)
                ctx.render(12)
                assert ctx.last_launch_path_scheduling() == sched
                out[sched] = (ctx.read_accum(), ctx.read_seeds(), [int(c) for c in ctx.read_counters()])
        a, b = out[bdpt.PATHS_PER_LANE], out[bdpt.PATHS_BY_CLASS]
        assert np.array_equal(a[0].view(np.uint32), b[0].view(np.uint32))
        assert np.array_equal(a[1], b[1])
        assert a[2][1:3] == b[2][1:3]                                        # full and early samples
        if chunks == 1:
            assert a[2] == b[2]
        else:
            assert 0 <= a[2][0] - b[2][0] < 50 * 41 * chunks
        assert np.count_nonzero(a[0][:7]) == 0 and np.count_nonzero(a[0][:, :13]) == 0      # outside the region nothing is written


def test_which_launches_run_by_class(bdpt):
    """Class order (the samples of a pixel sorted instead), more than five light-subpath vertices, the many-triangle
    kernel, the interleaved tile split and the old integrator are per-lane launches whatever is asked.  Left to itself
    the library sorts pixels in index-order launches of 32 samples or more."""
    scene = bdpt.Scene.cornell()
    W, H = 64, 32
    cases = ((dict(rng_mode=1, accum_mode=1, sample_order=2), False, bdpt.PATHS_PER_LANE),
             (dict(max_light_bounces=7), False, bdpt.PATHS_PER_LANE),
             (dict(kernel_variant=bdpt.KERNEL_TILED), False, bdpt.PATHS_PER_LANE),
             (dict(), True, bdpt.PATHS_PER_LANE),
             (dict(integrator=1), False, bdpt.PATHS_PER_LANE),
             (dict(sampler=bdpt.SAMPLER_HEMISPHERE), False, bdpt.PATHS_BY_CLASS),
             (dict(kernel_variant=bdpt.KERNEL_DIRECT), False, bdpt.PATHS_BY_CLASS))
    for kw, split, expect in cases:
        with bdpt.Context(W, H) as ctx:
            ctx.set_params(bdpt.default_params(**kw))
            ctx.upload_scene(scene); ctx.upload_seeds(bdpt.make_seeds(W, H, 1)); ctx.set_camera(scene.camera)
            ctx.set_path_scheduling(bdpt.PATHS_BY_CLASS)
            if split:
                ctx.set_tile_interleave(16, 8, 2, 1)
            ctx.render(2)
            assert ctx.last_launch_path_scheduling() == expect, kw
    with bdpt.Context(W, H) as ctx:
        ctx.set_params(bdpt.default_params())
        ctx.upload_scene(scene); ctx.upload_seeds(bdpt.make_seeds(W, H, 1)); ctx.set_camera(scene.camera)
        for spp, expect in ((1, bdpt.PATHS_PER_LANE), (31, bdpt.PATHS_PER_LANE), (32, bdpt.PATHS_BY_CLASS), (100, bdpt.PATHS_BY_CLASS)):
            ctx.render(spp)
            assert ctx.last_launch_path_scheduling() == expect, spp
        ctx.set_params(bdpt.default_params(rng_mode=1, accum_mode=1))             # class order where the mode allows it
        ctx.render(64)
        assert ctx.last_launch_path_scheduling() == bdpt.PATHS_PER_LANE


def test_by_class_under_graph_capture_allocates_nothing(bdpt):
    """The kernel keeps its state in LDS and in the arrays every launch has: a launch recorded into a HIP graph (no
    allocation allowed) runs by class and replays to the bits of a direct launch."""
    import torch
    scene = bdpt.Scene.cornell()
    W, H = 128, 72
    seeds = bdpt.make_seeds(W, H, 0x66)
    images = []
    for captured in (False, True):
        with bdpt.Context(W, H) as ctx:
            ctx.set_params(bdpt.default_params())
            ctx.upload_scene(scene); ctx.upload_seeds(seeds); ctx.set_camera(scene.camera)
            ctx.set_path_scheduling(bdpt.PATHS_BY_CLASS)
            if captured:
                stream = torch.cuda.Stream()
                ctx.set_stream(stream.cuda_stream)
                ctx.render(1)                                                   # sizes whatever a first launch sizes
                stream.synchronize()
                ctx.upload_seeds(seeds); ctx.set_camera(scene.camera)
                graph = torch.cuda.CUDAGraph()
                with torch.cuda.graph(graph, stream=stream):
                    ctx.render_async(40)
                assert ctx.last_launch_path_scheduling() == bdpt.PATHS_BY_CLASS
                ctx.upload_seeds(seeds); ctx.set_camera(scene.camera)
                graph.replay()
                stream.synchronize()
            else:
                ctx.render(40)
            images.append(ctx.read_accum())
    assert np.array_equal(images[0].view(np.uint32), images[1].view(np.uint32))


def test_by_class_full_size_frame_equals_per_lane(bdpt):
    """1920 x 1080 x 64 samples in the reference's mode, automatic slicing: every bit of the image and of the generator
    state as from the per-lane launch."""
    scene = bdpt.Scene.cornell()
    W, H = 1920, 1080
    seeds = bdpt.make_seeds(W, H, 0x5EED)
    out = []
    for sched in (bdpt.PATHS_PER_LANE, bdpt.PATHS_AUTO):
        with bdpt.Context(W, H) as ctx:
            ctx.set_params(bdpt.default_params())
            ctx.upload_scene(scene); ctx.upload_seeds(seeds); ctx.set_camera(scene.camera)
            ctx.set_path_scheduling(sched)
            ctx.set_launch_chunks(0)
            ctx.render(64)
            assert ctx.last_launch_path_scheduling() == (bdpt.PATHS_BY_CLASS if sched == bdpt.PATHS_AUTO else sched)
            out.append((ctx.read_accum(), ctx.read_seeds()))
    assert np.array_equal(out[0][0].view(np.uint32), out[1][0].view(np.uint32))
    assert np.array_equal(out[0][1], out[1][1])

"""Sliced launches (bdpt_set_launch_chunks): a render call run as several runs per tile must leave every bit of the
pixel state -- accumulator, generator words, parity buffers, counters -- as the unsliced launch does, in every
generator / accumulation / sample-order mode and kernel, and so must equal the CPU oracle."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

W, H = 88, 52                      # ragged against the workgroup tiles
AOVS = ("AOV_EYE_HIT", "AOV_HIT_HASH", "AOV_RADIANCE", "AOV_RNG_DRAWS", "AOV_RAYS")

MODES = {
    "stream_mean": {},
    "strided_mean": dict(rng_mode=1),
    "strided_sum_index": dict(rng_mode=1, accum_mode=1, sample_order=1),
    "strided_sum_class": dict(rng_mode=1, accum_mode=1, sample_order=2),
    "strided_sum_adaptive": dict(rng_mode=1, accum_mode=1, sample_order=3),
}


def _render(bdpt, scene, params, chunks, calls, region=None, interleave=None, counters=False, size=None, order_scratch=None,
            sample_offset=0):
    w, h = size or (W, H)
    with bdpt.Context(w, h) as ctx:
        if order_scratch is not None:
            ctx.set_order_scratch_limit(order_scratch)
        if sample_offset:
            ctx.set_sample_offset(sample_offset)
        ctx.set_params(params)
        ctx.upload_scene(scene)
        ctx.upload_seeds(bdpt.make_seeds(w, h, 4242))
        ctx.set_camera(scene.camera)
        ctx.set_launch_chunks(chunks)
        if region:
            ctx.set_region(*region)
        if interleave:
            ctx.set_tile_interleave(*interleave)
        if counters:
            ctx.enable_counters(True)
        for spp in calls:
            ctx.render(spp)
        out = {"accum": ctx.read_accum(), "seeds": ctx.read_seeds()}
        if params.aov_enable:
            for k in AOVS:
                out[k] = ctx.read_aov(getattr(bdpt, k))
        if counters:
            out["counters"] = np.array(ctx.read_counters(), np.uint64)
        return out


def _same(a, b, what, work_counters=True):
    """work_counters=False: compare only the sample counts.  In STREAM mode a pixel whose eye ray ends the sample
    repeats that sample for the rest of a launch (kernel0.cl:645-656), which the kernels trace once per run."""
    assert a.keys() == b.keys()
    for k in a:
        x, y = np.ascontiguousarray(a[k]), np.ascontiguousarray(b[k])
        if k == "counters" and not work_counters:
            x, y = x[1:3], y[1:3]
        assert x.tobytes() == y.tobytes(), "%s: %s differs between the sliced and the unsliced launch" % (what, k)


def test_mode_constants(bdpt):
    assert (bdpt.RNG_STRIDED, bdpt.ACCUM_SUM, bdpt.ORDER_INDEX, bdpt.ORDER_ROULETTE_CLASS, bdpt.ORDER_ADAPTIVE) == (1, 1, 1, 2, 3)


@pytest.mark.parametrize("variant", [1, 3, 4, 5, 6])
@pytest.mark.parametrize("mode", sorted(MODES))
def test_sliced_launch_is_bit_identical(bdpt, variant, mode):
    scene = bdpt.Scene.soup(60) if variant == 4 else bdpt.Scene.cornell()
    for aov in (1, 0):
        p = bdpt.default_params(kernel_variant=variant, aov_enable=aov, **MODES[mode])
        calls = (3, 37, 8)                               # 37 = 5 runs of 8, 8, 8, 8 and 5 samples
        whole = _render(bdpt, scene, p, 1, calls, counters=not aov)
        for chunks in (5, 37):
            _same(_render(bdpt, scene, p, chunks, calls, counters=not aov), whole, "%s, %d runs, aov %d" % (mode, chunks, aov),
                  work_counters=mode != "stream_mean")


@pytest.mark.parametrize("variant", [4, 5])
def test_sliced_launch_of_a_region_and_of_interleaved_tiles(bdpt, variant):
    scene = bdpt.Scene.soup(60) if variant == 4 else bdpt.Scene.cornell()
    p = bdpt.default_params(kernel_variant=variant, aov_enable=1, **MODES["strided_sum_adaptive"])
    for kw in (dict(region=(9, 5, 61, 40)), dict(interleave=(32, 16, 3, 1)), dict(region=(16, 8, 64, 40), interleave=(16, 8, 2, 1))):
        _same(_render(bdpt, scene, p, 4, (21,), **kw), _render(bdpt, scene, p, 1, (21,), **kw), str(kw))


@pytest.mark.parametrize("variant", [4, 5])
def test_sliced_launch_matches_the_oracle(bdpt, oracle, variant):
    scene = bdpt.Scene.soup(60) if variant == 4 else bdpt.Scene.cornell()
    for mode in ("stream_mean", "strided_sum_adaptive"):
        p = bdpt.default_params(kernel_variant=variant, aov_enable=1, **MODES[mode])
        got = _render(bdpt, scene, p, 6, (2, 23))
        st = oracle.State(W, H, bdpt.make_seeds(W, H, 4242))
        for spp in (2, 23):
            oracle.render(st, scene, scene.camera, p, spp, n_threads=8)
        assert np.array_equal(got["seeds"], st.seeds), mode
        assert np.array_equal(got["AOV_HIT_HASH"], st.hit_hash), mode
        assert np.array_equal(got["AOV_RNG_DRAWS"], st.rng_draws), mode
        assert got["accum"].tobytes() == st.accum.tobytes(), mode


def test_automatic_slicing_leaves_results_alone_and_short_launches_whole(bdpt):
    """chunks = 0: launches of at least 64 samples are sliced by the small-scene kernels (of fewer once the context
    knows how long a sample takes), every launch of more than one
    sample by the many-triangle kernel; results as with one run."""
    for variant, spp in ((5, 70), (4, 9)):
        scene = bdpt.Scene.soup(60) if variant == 4 else bdpt.Scene.cornell()
        p = bdpt.default_params(kernel_variant=variant, aov_enable=0, **MODES["strided_sum_adaptive"])
        _same(_render(bdpt, scene, p, 0, (spp, 1)), _render(bdpt, scene, p, 1, (spp, 1)), "automatic, variant %d" % variant)


@pytest.mark.parametrize("mode", ["stream_mean", "strided_sum_adaptive"])
def test_runs_of_a_tile_on_different_dies_at_full_occupancy(bdpt, mode):
    """A frame of 16 065 tiles (an odd number) with many short runs: workgroups are dealt to the eight dies in turn, so
    consecutive runs of a tile land on different dies, on a chip that is full; every run reads the state the run
    before wrote.  Parity buffers on, so that all the state arrays take part."""
    scene = bdpt.Scene.cornell()
    p = bdpt.default_params(kernel_variant=5, aov_enable=1, **MODES[mode])
    size = (1896, 1080)
    whole = _render(bdpt, scene, p, 1, (48,), size=size)
    for chunks in (48, 7):
        _same(_render(bdpt, scene, p, chunks, (48,), size=size), whole, "%s, %d runs" % (mode, chunks))


@pytest.mark.parametrize("size", [(152, 72), (328, 200), (776, 424)])
def test_runs_of_a_tile_on_different_dies_with_all_state_cache_resident(bdpt, size):
    """Frames of 90, 525 and 2 597 tiles (not multiples of the eight dies, so a tile's runs move from die to die)
    whose whole pixel state fits the second-level caches: a stale cached copy of state that another die has since
    rewritten would survive here.  STREAM mode with parity buffers hands over the 4-byte arrays too, whose cache
    lines hold pixels of several tiles.  Two launches of 40 one-sample runs."""
    scene = bdpt.Scene.cornell()
    for mode, aov in (("stream_mean", 1), ("strided_sum_adaptive", 0)):
        p = bdpt.default_params(kernel_variant=5, aov_enable=aov, **MODES[mode])
        whole = _render(bdpt, scene, p, 1, (40, 40), size=size)
        _same(_render(bdpt, scene, p, 40, (40, 40), size=size), whole, "%s %dx%d" % (mode, size[0], size[1]))


def test_benchmark_sized_launch_sliced_automatically(bdpt):
    """The headline configuration as bench.py runs it: 1920x1080, 1024 samples per pixel, STRIDED + SUM, adaptive
    order, automatic slicing (16 runs per tile) -- every bit of the image as from the unsliced launch."""
    scene = bdpt.Scene.cornell()
    p = bdpt.default_params(aov_enable=0, **MODES["strided_sum_adaptive"])
    size = (1920, 1080)
    with bdpt.Context(*size) as ctx:
        ctx.set_params(p)
        ctx.upload_scene(scene)
        ctx.upload_seeds(bdpt.make_seeds(size[0], size[1], 0x5EED))
        ctx.set_camera(scene.camera)
        ctx.set_launch_chunks(0)                       # automatic, whatever BDPT_LAUNCH_CHUNKS says
        ctx.set_path_scheduling(bdpt.PATHS_PER_LANE)   # ... and one lane per pixel, whatever BDPT_PATH_SCHEDULING says
        ctx.render(1024)
        assert ctx.last_launch_chunks() == 16 and ctx.last_launch_variant() == bdpt.KERNEL_LDS2
        sliced = ctx.read_accum()
        ctx.set_launch_chunks(1)
        ctx.num_samples = 0
        ctx.render(1024)
        assert ctx.last_launch_chunks() == 1
        whole = ctx.read_accum()
    assert sliced.tobytes() == whole.tobytes()
    assert np.isfinite(whole).all() and whole[..., :3].max() > 0


def test_launch_chunks_argument_is_checked(bdpt):
    with bdpt.Context(16, 8) as ctx:
        with pytest.raises(bdpt.BdptError):
            ctx.set_launch_chunks(-1)
        with pytest.raises(bdpt.BdptError):
            ctx.set_launch_chunks(100000)
        ctx.set_launch_chunks(0)


@pytest.mark.parametrize("variant", [1, 3, 5])
@pytest.mark.parametrize("order", ["strided_sum_class", "strided_sum_adaptive"])
def test_class_order_from_the_table_is_the_order_worked_out_on_the_fly(bdpt, variant, order):
    """A launch keeps the classes of its samples in a table in device memory when it may have the memory
    (bdpt_set_order_scratch_limit), and works them out class by class when not: same order, so the same bits, for
    sample counts that end inside a mask word, launches that continue a frame, a sample offset, more subpath
    vertices than the default, sliced and not."""
    scene = bdpt.Scene.cornell()
    for extra in (dict(), dict(max_light_bounces=7), dict(max_light_bounces=2), dict(max_light_bounces=1)):
        p = bdpt.default_params(kernel_variant=variant, aov_enable=1, **MODES[order], **extra)
        for chunks, calls, offset in ((1, (70, 5, 33), 0), (4, (70, 64), 0), (3, (45,), 1000)):
            table = _render(bdpt, scene, p, chunks, calls, counters=False, sample_offset=offset)
            none = _render(bdpt, scene, p, chunks, calls, counters=False, sample_offset=offset, order_scratch=0)
            _same(table, none, "%s, variant %d, %s, %d runs" % (order, variant, extra, chunks))
    # a region that starts off the tile grid, and tiles dealt to one of three shares
    p = bdpt.default_params(kernel_variant=variant, aov_enable=1, **MODES[order])
    for kw in (dict(region=(9, 5, 61, 40)), dict(interleave=(32, 16, 3, 1)), dict(region=(16, 8, 64, 40), interleave=(16, 8, 2, 1))):
        _same(_render(bdpt, scene, p, 3, (45, 20), **kw), _render(bdpt, scene, p, 3, (45, 20), order_scratch=0, **kw), "%s, %s" % (order, kw))
    # a limit just too small for the second launch: the first uses the table, the second does not
    p = bdpt.default_params(kernel_variant=variant, **MODES[order])
    words_32 = W * H * p.max_light_bounces * 4
    _same(_render(bdpt, scene, p, 2, (32, 33), order_scratch=words_32), _render(bdpt, scene, p, 2, (32, 33), order_scratch=0), "limit between two launches")


@pytest.mark.parametrize("variant", [4, 5])
def test_a_launch_that_gives_up_is_reported_on_every_way_of_waiting_and_leaves_nothing_stale(bdpt, variant):
    """With a wait limit of a nanosecond every workgroup that has to wait for the run before it gives up: the launch is
    abandoned (later runs drain without touching the image), and whichever call next waits for it -- bdpt_sync,
    bdpt_read_accum, bdpt_last_launch_ms after bdpt_render_async, or the next bdpt_render_async itself -- says so,
    once.  After a reset the context renders what a fresh context renders."""
    scene = bdpt.Scene.soup(60) if variant == 4 else bdpt.Scene.cornell()
    p = bdpt.default_params(kernel_variant=variant, **MODES["stream_mean"])
    w, h = 512, 256
    with bdpt.Context(w, h) as ctx:
        ctx.set_params(p)
        ctx.upload_scene(scene)
        ctx.upload_seeds(bdpt.make_seeds(w, h, 4242))
        ctx.set_camera(scene.camera)
        ctx.set_launch_chunks(1)
        ctx.render(160)
        want = ctx.read_accum()
        ctx.set_launch_chunks(8)
        for waiter in ("sync", "read_accum", "last_launch_ms", "render_async"):
            ctx.upload_seeds(bdpt.make_seeds(w, h, 4242))
            ctx.set_camera(scene.camera)
            ctx.set_slice_wait_limit(1e-9)
            ctx.render_async(160)
            assert ctx.last_launch_chunks() == 8
            with pytest.raises(bdpt.BdptError, match="did not complete in order"):
                if waiter == "sync":
                    ctx.sync()
                elif waiter == "read_accum":
                    ctx.read_accum()
                elif waiter == "last_launch_ms":
                    ctx.last_launch_ms()
                else:
                    import time
                    time.sleep(1.0)                     # nothing waits: the mark is simply there when the next launch is asked for
                    ctx.render_async(1)
            ctx.sync()                                  # reported once
            # the abandoned image holds only whole runs of whole tiles: every pixel is untouched or a sum that a run left
            ctx.set_slice_wait_limit(0)
            ctx.upload_seeds(bdpt.make_seeds(w, h, 4242))
            ctx.set_camera(scene.camera)
            ctx.render(160)
            assert ctx.read_accum().tobytes() == want.tobytes(), waiter
        with pytest.raises(bdpt.BdptError):
            ctx.set_slice_wait_limit(-1.0)

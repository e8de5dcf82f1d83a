"""GPU parity: the HIP path through the C-ABI against the CPU oracle on the same
seeded inputs.  Bars (stated once, used below):
  - closest-hit primitive ids, hit-id hash chains, generator state and draw
    counts: bit-exact;
  - per-sample radiance and accumulated image: the arithmetic is specified
    operation by operation (include/bdpt_clmath.h), so these are compared
    bit-exact as well; RADIANCE_RTOL/ATOL is the stated floating-point bar
    (fp32 relative 1e-5, absolute 1e-7) and is asserted first so a failure says
    how far off the values are.
"""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

RADIANCE_RTOL = 1e-5
RADIANCE_ATOL = 1e-7


def _pair(bdpt, oracle, scene, W, H, seed=0x5EED, **params):
    seeds = bdpt.make_seeds(W, H, seed)
    ctx = bdpt.Context(W, H)
    p = bdpt.default_params(aov_enable=1, **params)
    ctx.set_params(p)
    ctx.upload_scene(scene)
    ctx.upload_seeds(seeds)
    ctx.set_camera(scene.camera)
    st = oracle.State(W, H, seeds)
    return ctx, st, p


def _compare(bdpt, ctx, st, what=""):
    eye = ctx.read_aov(bdpt.AOV_EYE_HIT)
    assert np.array_equal(eye, st.eye_hit), what + " eye-hit ids differ at %d pixels" % np.count_nonzero((eye != st.eye_hit).any(-1))
    hh = ctx.read_aov(bdpt.AOV_HIT_HASH)
    assert np.array_equal(hh, st.hit_hash), what + " hit-id hash differs at %d pixels" % np.count_nonzero(hh != st.hit_hash)
    assert np.array_equal(ctx.read_aov(bdpt.AOV_RNG_DRAWS), st.rng_draws), what + " draw counts differ"
    assert np.array_equal(ctx.read_aov(bdpt.AOV_RAYS), st.rays), what + " ray counts differ"
    assert np.array_equal(ctx.read_seeds(), st.seeds), what + " generator state differs"
    rad = ctx.read_aov(bdpt.AOV_RADIANCE)
    np.testing.assert_allclose(rad, st.radiance, rtol=RADIANCE_RTOL, atol=RADIANCE_ATOL, err_msg=what + " radiance")
    acc = ctx.read_accum()
    np.testing.assert_allclose(acc, st.accum, rtol=RADIANCE_RTOL, atol=RADIANCE_ATOL, err_msg=what + " accumulator")
    assert np.array_equal(rad.view(np.uint32), st.radiance.view(np.uint32)), what + " radiance not bit-exact"
    assert np.array_equal(acc.view(np.uint32), st.accum.view(np.uint32)), what + " accumulator not bit-exact"


def test_clmath_device_matches_host_bitwise(bdpt, oracle):
    """Every built-in of the numeric contract rounds identically on gfx950 and on the host."""
    rng = np.random.default_rng(1)
    n = 200000
    ctx = bdpt.Context(8, 8)
    x = rng.uniform(0.0, 2 * np.pi, n).astype(np.float32)
    s = np.array([oracle.sincos(v) for v in x[:20000]], np.float32)
    assert np.array_equal(ctx.clmath_eval(0, x[:20000]).astype(np.float32).view(np.uint32), s[:, 0].view(np.uint32))
    assert np.array_equal(ctx.clmath_eval(1, x[:20000]).astype(np.float32).view(np.uint32), s[:, 1].view(np.uint32))
    u = np.concatenate([rng.uniform(-1, 1, 20000), [0.0, 1.0, -1.0, 0.5, -0.5, 0.5000001, 1e-30]]).astype(np.float32)
    a = np.array([oracle.acosf(v) for v in u], np.float32)
    assert np.array_equal(ctx.clmath_eval(2, u).astype(np.float32).view(np.uint32), a.view(np.uint32))
    ud = np.concatenate([rng.uniform(-1, 1, 20000), [0.0, 1.0, -1.0, 0.5, -0.5, 1 - 1e-10, -1 + 1e-10]])
    ad = np.array([oracle.acos_f64(v) for v in ud], np.float64)
    assert np.array_equal(ctx.clmath_eval(3, ud).view(np.uint64), ad.view(np.uint64))
    # IEEE sqrt / div / fma against numpy (correctly rounded on the host)
    v = np.abs(rng.standard_normal(n)).astype(np.float32) * np.float32(10.0) ** rng.integers(-20, 20, n).astype(np.float32)
    assert np.array_equal(ctx.clmath_eval(4, v).astype(np.float32).view(np.uint32), np.sqrt(v).view(np.uint32))
    a32 = (rng.standard_normal(n) * 10.0 ** rng.integers(-15, 15, n)).astype(np.float32)
    b32 = (rng.standard_normal(n) * 10.0 ** rng.integers(-15, 15, n)).astype(np.float32)
    with np.errstate(all="ignore"):
        q = a32 / b32
    got = ctx.clmath_eval(5, np.stack([a32, b32], 1)).astype(np.float32)
    assert np.array_equal(got.view(np.uint32), q.view(np.uint32))
    c32 = rng.standard_normal(n).astype(np.float32)
    a1 = rng.standard_normal(n).astype(np.float32)
    b1 = rng.standard_normal(n).astype(np.float32)
    fma = (a1.astype(np.float64) * b1.astype(np.float64) + c32.astype(np.float64)).astype(np.float32)  # exact product in f64, one rounding (double rounding is negligible but not impossible)
    got = ctx.clmath_eval(6, np.stack([a1, b1, c32], 1)).astype(np.float32)
    assert np.count_nonzero(got.view(np.uint32) != fma.view(np.uint32)) <= n // 100000 + 2
    ctx.close()


@pytest.mark.parametrize("scene_name", ["cornell", "cornell_planes", "outdoor"])
def test_closest_hit_ids_bit_exact_on_identical_rays(bdpt, oracle, scene_name):
    scene = getattr(bdpt.Scene, scene_name)()
    rng = np.random.default_rng(7)
    n = 4000
    o = rng.uniform(-2.9, 2.9, (n, 3)).astype(np.float32)
    d = rng.standard_normal((n, 3)).astype(np.float32)
    d /= np.linalg.norm(d, axis=1, keepdims=True).astype(np.float32)
    org = np.tile(np.array([4, 0], np.int32), (n, 1))
    org[: n // 4] = [0, 0]          # rays leaving the light sphere skip it
    org[n // 4: n // 2] = [2, 3]    # rays leaving triangle 3 skip it
    ctx = bdpt.Context(8, 8)
    ctx.upload_scene(scene)
    ids, pos, nrm = ctx.query_rays(np.concatenate([o, d], 1), org)
    for i in range(n):
        hit, oid, opos, onrm = oracle.ray_intersect(scene, o[i], d[i], org[i])
        assert tuple(ids[i]) == tuple(oid), "ray %d: gpu %s oracle %s" % (i, ids[i], oid)
        if hit:
            assert np.array_equal(pos[i].view(np.uint32), opos.view(np.uint32))
            assert np.array_equal(nrm[i].view(np.uint32), onrm.view(np.uint32))
    ctx.close()


def test_near_ties_resolve_like_the_reference_order(bdpt, oracle):
    """Rays aimed at the diagonals and edges shared by two wall triangles: both triangles report a hit at (almost)
    the same distance, and the winner must be the one the sequential strict-'<' search keeps."""
    scene = bdpt.Scene.cornell()
    rng = np.random.default_rng(11)
    tri = scene.triangles
    targets = []
    for k in range(0, 12, 2):                       # the two triangles of wall k//2 share the edge p1-p2 of the first
        a, b = tri["p1"][k, :3], tri["p2"][k, :3]
        for t in rng.uniform(0.02, 0.98, 60):
            targets.append(a + np.float32(t) * (b - a))
    for _ in range(120):                            # box edges shared by two walls
        c = rng.choice([-3.0, 3.0], 3)
        c[rng.integers(0, 3)] = rng.uniform(-2.9, 2.9)
        targets.append(c.astype(np.float32))
    targets = np.array(targets, np.float32)
    o = rng.uniform(-1.0, 1.0, targets.shape).astype(np.float32)
    o[::7] = 0.0                                    # exact ties from the centre
    d = targets - o
    d = (d / np.linalg.norm(d, axis=1, keepdims=True)).astype(np.float32)
    org = np.tile(np.array([4, 0], np.int32), (len(d), 1))
    ctx = bdpt.Context(8, 8)
    ctx.upload_scene(scene)
    ids, pos, nrm = ctx.query_rays(np.concatenate([o, d], 1), org)
    exact_ties = 0
    for i in range(len(d)):
        hit, oid, opos, onrm = oracle.ray_intersect(scene, o[i], d[i], org[i])
        assert tuple(ids[i]) == tuple(oid), "ray %d: gpu %s oracle %s" % (i, ids[i], oid)
        assert np.array_equal(pos[i].view(np.uint32), opos.view(np.uint32))
        if oid[0] == 2:
            other = oid[1] ^ 1
            h2, p2 = oracle.ray_primitive("triangle", tri[other:other + 1], o[i], d[i])
            exact_ties += int(h2 and np.linalg.norm(p2 - o[i]) == np.linalg.norm(opos - o[i]))
    assert exact_ties > 10
    ctx.close()


VARIANTS = [1, 3, 4, 5, 6]   # DIRECT, LDS, TILED, LDS2, BVH: every kernel variant must give the same bits


@pytest.mark.parametrize("variant", VARIANTS)
def test_cornell_per_sample_parity_stream_mode(bdpt, oracle, variant):
    """64x64 Cornell, 16 samples one launch each: ids, hashes, radiance, accumulator and seeds after every sample."""
    scene = bdpt.Scene.cornell()
    W = H = 64
    ctx, st, p = _pair(bdpt, oracle, scene, W, H, kernel_variant=variant)
    for s in range(16):
        ctx.render(1)
        oracle.render(st, scene, scene.camera, p, 1, n_threads=8)
        _compare(bdpt, ctx, st, "sample %d:" % s)
    assert ctx.num_samples == 16
    ctx.close()


@pytest.mark.parametrize("variant", VARIANTS)
def test_batched_launch_equals_sample_by_sample(bdpt, oracle, variant):
    """One launch of 24 samples gives the bits of 24 one-sample launches (the oracle runs them one by one)."""
    scene = bdpt.Scene.cornell()
    W, H = 96, 56
    ctx, st, p = _pair(bdpt, oracle, scene, W, H, seed=99, kernel_variant=variant)
    ctx.render(24)
    for _ in range(24):
        oracle.render(st, scene, scene.camera, p, 1, n_threads=8)
    _compare(bdpt, ctx, st, "24 spp:")
    ctx.render(5)
    oracle.render(st, scene, scene.camera, p, 5, n_threads=8)
    _compare(bdpt, ctx, st, "24+5 spp:")
    ctx.close()


@pytest.mark.parametrize("variant", VARIANTS)
@pytest.mark.parametrize("scene_name,W,H", [("cornell_planes", 64, 48), ("outdoor", 80, 48)])
def test_other_scenes_parity(bdpt, oracle, scene_name, W, H, variant):
    scene = getattr(bdpt.Scene, scene_name)()
    ctx, st, p = _pair(bdpt, oracle, scene, W, H, seed=3, kernel_variant=variant)
    ctx.render(8)
    oracle.render(st, scene, scene.camera, p, 8, n_threads=8)
    _compare(bdpt, ctx, st, scene_name + ":")
    ctx.close()


@pytest.mark.parametrize("variant", [0, 1, 4, 6])
def test_soup_scene_parity(bdpt, oracle, variant):
    """Cornell + 3000 soup triangles, 32x32 x 2 spp (the searches over many primitives: plain, exhaustive with bounding
    spheres, hierarchy; AUTO must pick the hierarchy)."""
    scene = bdpt.Scene.soup(3000)
    ctx, st, p = _pair(bdpt, oracle, scene, 32, 32, seed=5, kernel_variant=variant)
    ctx.render(2)
    oracle.render(st, scene, scene.camera, p, 2, n_threads=8)
    _compare(bdpt, ctx, st, "soup:")
    ctx.close()


@pytest.mark.parametrize("variant", [0, 1, 4])
def test_full_size_soup_region_against_the_oracle(bdpt, oracle, variant):
    """Cornell box + 100 000 soup triangles on the 1920x1080 frame, a 32x32 region x 2 spp, in one launch: the hierarchy
    (AUTO must pick it), the tiled kernel over all 521 tiles and the plain walk, against the oracle run on the same region."""
    scene = bdpt.Scene.soup(100000)
    W, H = 1920, 1080
    region = (1100, 380, 32, 32)
    ctx, st, p = _pair(bdpt, oracle, scene, W, H, seed=0xC4, kernel_variant=variant)
    ctx.set_region(*region)
    ctx.render(2)
    assert ctx.last_launch_variant() == {0: bdpt.KERNEL_BVH, 1: bdpt.KERNEL_DIRECT, 4: bdpt.KERNEL_TILED}[variant]
    oracle.render(st, scene, scene.camera, p, 2, region=region, n_threads=16)
    _compare(bdpt, ctx, st, "100k soup region:")
    assert (st.eye_hit[380:412, 1100:1132, 1] >= 12).any()
    ctx.close()


@pytest.mark.parametrize("variant", VARIANTS)
@pytest.mark.parametrize("region", [(1240, 300, 64, 64), (1856, 0, 64, 64)])
def test_cornell_region_of_the_1080p_frame_strided_sum(bdpt, oracle, variant, region):
    """A 64x64 region of the 1920x1080 Cornell frame x 16 spp in the benchmark's mode (STRIDED + SUM, adaptive
    order, a sample offset as a second rank would have): pixel coordinates and seed index at the full frame size."""
    scene = bdpt.Scene.cornell()
    W, H = 1920, 1080
    ctx, st, p = _pair(bdpt, oracle, scene, W, H, seed=0x5EED, kernel_variant=variant, rng_mode=bdpt.RNG_STRIDED,
                       accum_mode=bdpt.ACCUM_SUM)
    ctx.set_region(*region)
    ctx.set_sample_offset(1024)
    ctx.render(16)
    oracle.render(st, scene, scene.camera, p, 16, region=region, sample_offset=1024, n_threads=16)
    _compare(bdpt, ctx, st, "1080p region %r:" % (region,))
    ctx.close()


@pytest.mark.parametrize("variant", VARIANTS)
def test_strided_mode_and_sample_split(bdpt, oracle, variant):
    """STRIDED + SUM: two contexts rendering samples [0,6) and [6,12) add up to one context rendering [0,12)."""
    scene = bdpt.Scene.cornell()
    W, H = 64, 40
    kw = dict(rng_mode=bdpt.RNG_STRIDED, accum_mode=bdpt.ACCUM_SUM, kernel_variant=variant)
    ctx, st, p = _pair(bdpt, oracle, scene, W, H, seed=11, **kw)
    ctx.render(12)
    oracle.render(st, scene, scene.camera, p, 12, n_threads=8)
    _compare(bdpt, ctx, st, "strided 12:")
    whole = ctx.read_accum()
    ctx.close()

    parts = []
    for first in (0, 6):
        c, s2, p2 = _pair(bdpt, oracle, scene, W, H, seed=11, **kw)
        c.set_sample_offset(first)
        c.render(6)
        oracle.render(s2, scene, scene.camera, p2, 6, sample_offset=first, n_threads=8)
        _compare(bdpt, c, s2, "strided part %d:" % first)
        parts.append(c.read_accum())
        c.close()
    np.testing.assert_allclose(parts[0] + parts[1], whole, rtol=1e-5, atol=1e-6)


@pytest.mark.parametrize("variant", VARIANTS)
def test_strided_sum_launch_adds_by_roulette_class(bdpt, oracle, variant):
    """A 40-sample STRIDED + SUM launch: the device takes the samples class by class; so does the oracle, bit for bit.
    The classes only reorder the additions: the sum matches the index-ordered MEAN of the same samples to rounding."""
    scene = bdpt.Scene.cornell()
    W, H, SPP = 64, 40, 40
    ctx, st, p = _pair(bdpt, oracle, scene, W, H, seed=17, rng_mode=bdpt.RNG_STRIDED, accum_mode=bdpt.ACCUM_SUM,
                       kernel_variant=variant)
    ctx.set_sample_offset(3)
    ctx.render(SPP)
    oracle.render(st, scene, scene.camera, p, SPP, sample_offset=3, n_threads=8)
    _compare(bdpt, ctx, st, "class-ordered sum:")
    total = ctx.read_accum()
    ctx.close()
    ctx2, st2, p2 = _pair(bdpt, oracle, scene, W, H, seed=17, rng_mode=bdpt.RNG_STRIDED, kernel_variant=variant)
    ctx2.set_sample_offset(3)
    ctx2.render(SPP)
    np.testing.assert_allclose(total / SPP, ctx2.read_accum(), rtol=2e-5, atol=1e-6)
    ctx2.close()


@pytest.mark.parametrize("variant", VARIANTS)
@pytest.mark.parametrize("p_continue,max_l", [(0.8, 5), (0.5, 7), (0.93, 3), (0.8, 1)])
def test_sample_orders_take_the_same_samples(bdpt, oracle, variant, p_continue, max_l):
    """STRIDED + SUM with sample_order INDEX and ROULETTE_CLASS: each bit-exact against the oracle in its own order,
    the same draw and ray counts in both (every sample is taken exactly once), sums equal to rounding."""
    scene = bdpt.Scene.cornell()
    W, H, SPP = 40, 24, 70          # 70 samples: more than two 32-sample look-ahead windows
    res = {}
    for order in (bdpt.ORDER_INDEX, bdpt.ORDER_ROULETTE_CLASS):
        ctx, st, p = _pair(bdpt, oracle, scene, W, H, seed=23, rng_mode=bdpt.RNG_STRIDED, accum_mode=bdpt.ACCUM_SUM,
                           kernel_variant=variant, sample_order=order, p_continue=p_continue, max_light_bounces=max_l)
        ctx.render(SPP)
        oracle.render(st, scene, scene.camera, p, SPP, n_threads=8)
        _compare(bdpt, ctx, st, "order %d:" % order)
        res[order] = (ctx.read_accum(), ctx.read_aov(bdpt.AOV_RNG_DRAWS), ctx.read_aov(bdpt.AOV_RAYS))
        ctx.close()
    a, b = res[bdpt.ORDER_INDEX], res[bdpt.ORDER_ROULETTE_CLASS]
    np.testing.assert_array_equal(a[1], b[1])
    np.testing.assert_array_equal(a[2], b[2])
    np.testing.assert_allclose(a[0], b[0], rtol=2e-5, atol=1e-5)


@pytest.mark.parametrize("variant", [1, 3, 4, 5, 6])
@pytest.mark.parametrize("sampler", [1, 2, 3])
def test_alternative_samplers_match_the_oracle(bdpt, oracle, variant, sampler):
    """HEMISPHERE, REFLECTIVE and LAMBERT_IMP bounce samplers: per-sample parity in STREAM mode and a class-ordered STRIDED + SUM
    launch, on every kernel variant that has them."""
    scene = bdpt.Scene.cornell()
    W, H = 48, 40
    extra = dict(imp_light_weight=0.4) if sampler == 3 else {}      # both branches of brdf_lambertian_imp
    ctx, st, p = _pair(bdpt, oracle, scene, W, H, seed=41, kernel_variant=variant, sampler=sampler, **extra)
    for s in range(6):
        ctx.render(1)
        oracle.render(st, scene, scene.camera, p, 1, n_threads=8)
        _compare(bdpt, ctx, st, "sampler %d sample %d:" % (sampler, s))
    ctx.render(5)
    oracle.render(st, scene, scene.camera, p, 5, n_threads=8)
    _compare(bdpt, ctx, st, "sampler %d batch:" % sampler)
    ctx.close()
    ctx, st, p = _pair(bdpt, oracle, scene, W, H, seed=43, kernel_variant=variant, sampler=sampler,
                       rng_mode=bdpt.RNG_STRIDED, accum_mode=bdpt.ACCUM_SUM, **extra)
    ctx.render(45)
    oracle.render(st, scene, scene.camera, p, 45, n_threads=8)
    _compare(bdpt, ctx, st, "sampler %d strided sum:" % sampler)
    ctx.close()


@pytest.mark.parametrize("scene_name", ["cornell", "outdoor", "cornell_planes"])
def test_path_trace_integrator_matches_the_oracle(bdpt, oracle, scene_name):
    """The unidirectional kernel the reference keeps commented out (kernel0.cl:872-1001): launches of 1, 3 and 6
    samples (one discarded draw per launch, as its sample loop has it), every parity output and the image."""
    scene = getattr(bdpt.Scene, scene_name)()
    W, H = 72, 48
    ctx, st, p = _pair(bdpt, oracle, scene, W, H, seed=51, integrator=bdpt.INTEGRATOR_PATH_TRACE)
    for spp in (1, 3, 6):
        ctx.render(spp)
        oracle.render(st, scene, scene.camera, p, spp, n_threads=8)
        _compare(bdpt, ctx, st, "path_trace %s spp %d:" % (scene_name, spp))
    assert ctx.num_samples == 10
    ctx.close()
    with bdpt.Context(16, 16) as c2:
        with pytest.raises(bdpt.BdptError, match="PATH_TRACE"):
            c2.set_params(bdpt.default_params(integrator=bdpt.INTEGRATOR_PATH_TRACE, rng_mode=bdpt.RNG_STRIDED))


def test_unknown_variants_and_samplers_are_refused(bdpt):
    ctx = bdpt.Context(16, 16)
    for bad in (2, 7, -1):                      # 2 was a variant that no longer exists
        with pytest.raises(bdpt.BdptError, match="kernel_variant"):
            ctx.set_params(bdpt.default_params(kernel_variant=bad))
    with pytest.raises(bdpt.BdptError, match="sampler"):
        ctx.set_params(bdpt.default_params(sampler=7))
    ctx.close()


def _mixed_soup(bdpt, n, seed):
    """Cornell box plus n triangles of every size and shape: specks, slivers, wall-sized sheets, some far outside."""
    rng = np.random.default_rng(seed)
    sc = bdpt.Scene.cornell()
    tri = np.zeros(n, bdpt.triangle_dtype)
    size = 10.0 ** rng.uniform(-3.0, 0.6, n)
    centre = rng.uniform(-2.8, 2.8, (n, 3))
    centre[::17] *= 40.0                                        # far from everything, large coordinates
    e1 = rng.standard_normal((n, 3)) * size[:, None]
    e2 = rng.standard_normal((n, 3)) * size[:, None]
    sliver = rng.random(n) < 0.2
    thin = 10.0 ** rng.uniform(-7.0, -1.0, (sliver.sum(), 1))   # down to where the computed normal is noise
    e2[sliver] = e1[sliver] * rng.uniform(0.2, 3.0, (sliver.sum(), 1)) + rng.standard_normal((sliver.sum(), 3)) * size[sliver, None] * thin
    tri["p0"][:, :3] = centre
    tri["p1"][:, :3] = centre + e1
    tri["p2"][:, :3] = centre + e2
    tri["surface"] = rng.choice([0, 1, 3, 4], n)
    sc.triangles = np.concatenate([sc.triangles, tri])
    return sc


@pytest.mark.parametrize("seed,n", [(1, 700), (2, 1500), (3, 260)])
@pytest.mark.parametrize("variant", [4, 6])
def test_tiled_kernel_bounding_sphere_cull_never_drops_a_hit(bdpt, oracle, seed, n, variant):
    """The tiled kernel skips a triangle for a ray that clearly passes its bounding sphere, and the hierarchy skips whole
    boxes of them.  Triangles from 1e-3 to 4 units, slivers and far-away ones, seen by eye, connection and bounce rays:
    every id, count and bit as the oracle."""
    scene = _mixed_soup(bdpt, n, seed)
    W, H = 56, 40
    ctx, st, p = _pair(bdpt, oracle, scene, W, H, seed=seed, kernel_variant=variant)
    for s in range(3):
        ctx.render(2)
        oracle.render(st, scene, scene.camera, p, 2, n_threads=8)
        _compare(bdpt, ctx, st, "mixed soup %d launch %d:" % (seed, s))
    ctx.close()
    # the same scene through the kernel without the cull
    ctx, st2, p2 = _pair(bdpt, oracle, scene, W, H, seed=seed, kernel_variant=bdpt.KERNEL_DIRECT)
    ctx.render(6)
    assert np.array_equal(ctx.read_aov(bdpt.AOV_HIT_HASH), st.hit_hash)
    ctx.close()


@pytest.mark.parametrize("seed", [1, 2, 3, 4])
def test_two_stage_kernel_conservative_test_never_drops_a_hit(bdpt, oracle, seed):
    """The two-stage LDS kernel rules a triangle out for a ray with an approximate test and a margin.  Specks,
    slivers down to where the normal is noise, wall-sized sheets, triangles far outside, seen by eye, connection
    and bounce rays (grazing ones included): every id, count and bit as the oracle."""
    scene = _mixed_soup(bdpt, 110, seed)
    W, H = 64, 48
    ctx, st, p = _pair(bdpt, oracle, scene, W, H, seed=seed, kernel_variant=bdpt.KERNEL_LDS2)
    for s in range(3):
        ctx.render(3)
        oracle.render(st, scene, scene.camera, p, 3, n_threads=8)
        _compare(bdpt, ctx, st, "two-stage, mixed soup %d launch %d:" % (seed, s))
    ctx.close()


def _quad_walls(bdpt, n_quads, n_loose, seed):
    """Cornell box plus parallelograms of every size, two thirds of them rectangles with sides along the axes (the
    others seldom give both halves the same plane bit for bit), each cut into two triangles along either diagonal, with the vertices in any rotation and either winding (the halves of some wound differently, so that
    their normals differ), a few with the fourth corner nudged off the parallelogram within its plane, and loose triangles between."""
    rng = np.random.default_rng(seed)
    sc = bdpt.Scene.cornell()
    tris = []
    for q in range(n_quads):
        size = 10.0 ** rng.uniform(-1.5, 0.8)
        p = rng.uniform(-2.5, 2.5, 3).astype(np.float32)
        e1 = (rng.standard_normal(3) * size).astype(np.float32)
        e2 = (rng.standard_normal(3) * size).astype(np.float32)
        if q % 3 != 0:                                                  # axis-aligned: both halves get the same normal and a.n bit for bit
            i, j = rng.choice(3, 2, replace=False)
            e1 = np.zeros(3, np.float32); e1[i] = np.float32(rng.choice([-1, 1]) * size * rng.uniform(0.3, 3.0))
            e2 = np.zeros(3, np.float32); e2[j] = np.float32(rng.choice([-1, 1]) * size * rng.uniform(0.3, 3.0))
        corner = [p, p + e1, p + e1 + e2, p + e2]                      # float32 sums: the fourth corner is rounded
        if q % 5 == 4:                                                  # in its plane: from a rounding error's worth to a plain trapezoid
            corner[2] = (corner[2] + e1 * np.float32(10.0 ** rng.uniform(-7.5, -0.5))).astype(np.float32)
        halves = ([(0, 1, 2), (0, 2, 3)], [(1, 2, 3), (1, 3, 0)])[int(rng.integers(2))]
        flip_both = rng.random() < 0.5
        for k, h in enumerate(halves):
            r = int(rng.integers(3))
            h = h[r:] + h[:r]
            if flip_both != (k == 1 and q % 7 == 6):
                h = (h[0], h[2], h[1])
            tris.append([corner[i] for i in h])
        for _ in range(int(rng.integers(0, 2)) if n_loose else 0):
            c = rng.uniform(-2.8, 2.8, 3)
            tris.append([c, c + rng.standard_normal(3) * 0.5, c + rng.standard_normal(3) * 0.5])
            n_loose -= 1
    tri = np.zeros(len(tris), bdpt.triangle_dtype)
    for i, t in enumerate(tris):
        tri["p0"][i, :3], tri["p1"][i, :3], tri["p2"][i, :3] = t
    tri["surface"] = rng.choice([0, 1, 3, 4], len(tris))
    sc.triangles = np.concatenate([sc.triangles, tri])
    return sc


@pytest.mark.parametrize("seed", [11, 12, 13])
def test_two_stage_kernel_parallelogram_pairs_never_drop_a_hit(bdpt, oracle, seed):
    """The two halves of a parallelogram share the plane part and the coordinate planes of the conservative test (the
    second half's coordinates are derived from the first one's).  Every id, count and bit as the oracle, for
    parallelograms of all shapes, cuts, windings, and near-misses of the shape."""
    scene = _quad_walls(bdpt, 34, 16, seed)
    W, H = 64, 48
    ctx, st, p = _pair(bdpt, oracle, scene, W, H, seed=seed, kernel_variant=bdpt.KERNEL_LDS2)
    quads, pairs, singles, slabs = ctx.two_stage_record_counts()
    assert quads >= 16 and pairs >= 2 and singles >= 16 and 4 * slabs + 2 * (quads + pairs) + singles == len(scene.triangles), (quads, pairs, singles, slabs)
    for s in range(3):
        ctx.render(3)
        oracle.render(st, scene, scene.camera, p, 3, n_threads=8)
        _compare(bdpt, ctx, st, "two-stage, parallelograms %d launch %d:" % (seed, s))
    ctx.close()


def _boxes(bdpt, n_boxes, seed):
    """Cornell box plus boxes with sides along the axes, seen from inside or outside (either winding), every face cut
    into two triangles along either diagonal with the vertices in any rotation: opposite faces mostly along the same
    diagonal, some along the other one, some faces wound against the rest of their box, some moved within their plane
    or made smaller than the face opposite, and the triangles of some boxes in shuffled order."""
    rng = np.random.default_rng(seed)
    sc = bdpt.Scene.cornell()
    tris = []
    for b in range(n_boxes):
        c = rng.uniform(-2.0, 2.0, 3).astype(np.float32)
        h = (10.0 ** rng.uniform(-1.0, 0.3, 3)).astype(np.float32)
        if b % 2 == 0:                                                  # coordinates that add and subtract without rounding
            c = np.round(c * 4).astype(np.float32) / 4
            h = np.float32(2.0) ** np.round(np.log2(h)).astype(np.float32)
        inward = rng.random() < 0.5
        box = []
        for axis in range(3):
            i, j = [(1, 2), (2, 0), (0, 1)][axis]
            same_cut = rng.random() < 0.7
            cut = int(rng.integers(2))
            for side in (-1, 1):
                o = c.copy(); o[axis] = c[axis] + side * h[axis]
                ei = np.zeros(3, np.float32); ei[i] = h[i]
                ej = np.zeros(3, np.float32); ej[j] = h[j]
                if side == 1 and b % 5 == 3 and axis == 0:
                    o = (o + ei * np.float32(0.25)).astype(np.float32)  # moved within its plane
                if side == 1 and b % 5 == 4 and axis == 1:
                    ej = (ej * np.float32(0.5)).astype(np.float32)      # smaller than the face opposite
                corner = [o - ei - ej, o + ei - ej, o + ei + ej, o - ei + ej]     # counter-clockwise seen from +axis
                this_cut = cut if (same_cut or side == -1) else 1 - cut
                halves = ([(0, 1, 2), (0, 2, 3)], [(1, 2, 3), (1, 3, 0)])[this_cut]
                flip = (side == 1) == inward                            # outward normals unless the box is seen from inside
                if b % 7 == 6 and axis == 2 and side == 1:
                    flip = not flip                                     # one face against the rest
                for hv in halves:
                    r = int(rng.integers(3))
                    hv = hv[r:] + hv[:r]
                    if flip:
                        hv = (hv[0], hv[2], hv[1])
                    box.append([corner[k] for k in hv])
        if b % 3 == 2:
            rng.shuffle(box)
        tris += box
    tri = np.zeros(len(tris), bdpt.triangle_dtype)
    for i, t in enumerate(tris):
        tri["p0"][i, :3], tri["p1"][i, :3], tri["p2"][i, :3] = t
    tri["surface"] = rng.choice([0, 1, 3, 4], len(tris))
    sc.triangles = np.concatenate([sc.triangles, tri])
    return sc


@pytest.mark.parametrize("seed", [21, 22, 23])
def test_two_stage_kernel_opposite_walls_never_drop_a_hit(bdpt, oracle, seed):
    """Two opposite walls of a box, cut along the same diagonal, share one record of the conservative test: a ray is
    tested against the one whose normal it runs against.  Every id, count and bit as the oracle, for boxes seen from
    inside and from outside, with the exceptions that must fall back to records of their own."""
    scene = _boxes(bdpt, 7, seed)
    W, H = 64, 48
    ctx, st, p = _pair(bdpt, oracle, scene, W, H, seed=seed, kernel_variant=bdpt.KERNEL_LDS2)
    quads, pairs, singles, slabs = ctx.two_stage_record_counts()
    assert slabs >= 8 and quads >= 4 and 4 * slabs + 2 * (quads + pairs) + singles == len(scene.triangles), (quads, pairs, singles, slabs)
    for s in range(3):
        ctx.render(3)
        oracle.render(st, scene, scene.camera, p, 3, n_threads=8)
        _compare(bdpt, ctx, st, "two-stage, boxes %d launch %d:" % (seed, s))
    ctx.close()


def test_the_cornell_box_is_three_pairs_of_opposite_walls(bdpt):
    with bdpt.Context(32, 32) as ctx:
        ctx.upload_scene(bdpt.Scene.cornell())
        assert ctx.two_stage_record_counts() == (0, 0, 0, 3)


def test_auto_picks_the_kernel_by_the_shape_of_the_scene(bdpt):
    """AUTO: the Cornell box (12 wall-sized triangles) stays in LDS (two-stage kernel), also with a few more triangles; a few dozen
    small triangles tip it to the tiled kernel with its bounding-sphere test; a scene too large for LDS always."""
    def variant_for(scene):
        with bdpt.Context(32, 32) as ctx:
            ctx.upload_scene(scene)
            ctx.upload_seeds(bdpt.make_seeds(32, 32, 1))
            ctx.set_camera(scene.camera)
            assert ctx.last_launch_variant() == 0
            ctx.render(1)
            return ctx.last_launch_variant()
    assert variant_for(bdpt.Scene.cornell()) == bdpt.KERNEL_LDS2
    assert variant_for(bdpt.Scene.cornell_planes()) == bdpt.KERNEL_LDS     # no triangles: nothing for two stages to do
    assert variant_for(bdpt.Scene.outdoor()) == bdpt.KERNEL_LDS
    assert variant_for(bdpt.Scene.soup(8)) == bdpt.KERNEL_LDS2
    assert variant_for(bdpt.Scene.soup(40)) == bdpt.KERNEL_TILED            # too few triangles for a hierarchy to pay
    assert variant_for(bdpt.Scene.soup(2000)) == bdpt.KERNEL_BVH


def test_class_ordered_launches_over_random_parameters(bdpt, oracle):
    """24 random (spp, max_light_bounces, p_continue, sample offset, sampler) combinations of a class-ordered launch,
    two launches each so that the second starts from a non-zero counter: every one bit for bit like the oracle."""
    rng = np.random.default_rng(2024)
    scene = bdpt.Scene.cornell()
    W, H = 24, 16
    for case in range(24):
        spp = int(rng.choice([1, 2, 3, 31, 32, 33, 64, 65, 97, 150]))
        kw = dict(rng_mode=bdpt.RNG_STRIDED, accum_mode=bdpt.ACCUM_SUM, max_light_bounces=int(rng.integers(1, 10)),
                  p_continue=float(rng.uniform(0.05, 0.97)), sampler=int(rng.choice([0, 0, 1, 2, 3])),
                  kernel_variant=int(rng.choice([0, 1, 3, 4, 5])))
        # STRIDED keeps the samples of a pixel 32 draws apart: at most 7 vertices (5 with the REFLECTIVE sampler)
        kw["max_light_bounces"] = min(kw["max_light_bounces"], 5 if kw["sampler"] >= 2 else 7)
        offset = int(rng.integers(0, 5000))
        ctx, st, p = _pair(bdpt, oracle, scene, W, H, seed=100 + case, **kw)
        ctx.set_sample_offset(offset)
        for launch in range(2):
            ctx.render(spp)
            oracle.render(st, scene, scene.camera, p, spp, sample_offset=offset, n_threads=8)
            _compare(bdpt, ctx, st, "case %d %r spp %d offset %d launch %d:" % (case, kw, spp, offset, launch))
        ctx.close()


@pytest.mark.parametrize("variant", [0, 1, 3, 4, 5])
@pytest.mark.parametrize("mode", ["stream", "strided_sum", "strided_mean"])
def test_launches_without_parity_buffers_give_the_same_image_and_seeds(bdpt, oracle, variant, mode):
    """With aov_enable = 0 the kernels leave out work only the parity buffers can see (the bounce ray of the last
    vertex, bounce directions that roulette discards): accumulator and generator state must not notice."""
    scene = bdpt.Scene.soup(40) if variant == 4 else bdpt.Scene.cornell()
    W, H = 64, 48
    kw = {"stream": {}, "strided_sum": dict(rng_mode=bdpt.RNG_STRIDED, accum_mode=bdpt.ACCUM_SUM),
          "strided_mean": dict(rng_mode=bdpt.RNG_STRIDED)}[mode]
    for extra in ({}, dict(max_light_bounces=2, p_continue=0.95), dict(sampler=bdpt.SAMPLER_REFLECTIVE), dict(indirect_only=1)):
        seeds = bdpt.make_seeds(W, H, 77)
        p = bdpt.default_params(kernel_variant=variant, **kw, **extra)
        with bdpt.Context(W, H) as ctx:
            ctx.set_params(p)
            ctx.upload_scene(scene)
            ctx.upload_seeds(seeds)
            ctx.set_camera(scene.camera)
            ctx.render(1)
            ctx.render(9)
            acc, out_seeds = ctx.read_accum(), ctx.read_seeds()
        st = oracle.State(W, H, seeds)
        oracle.render(st, scene, scene.camera, p, 1, n_threads=8)
        oracle.render(st, scene, scene.camera, p, 9, n_threads=8)
        assert np.array_equal(out_seeds, st.seeds), (mode, extra)
        assert np.array_equal(acc.view(np.uint32), st.accum.view(np.uint32)), (mode, extra)


@pytest.mark.parametrize("variant", VARIANTS)
@pytest.mark.parametrize("which", ["cornell", "outdoor"])
def test_adaptive_order_decides_per_pixel(bdpt, oracle, variant, which):
    """ADAPTIVE: pixels whose first eye ray ends the sample take index order, the others class order.  Bit for bit like
    the oracle; equal to INDEX on the first kind of pixel and to ROULETTE_CLASS on the second."""
    scene = getattr(bdpt.Scene, which)()
    W, H, SPP = 48, 32, 40
    res = {}
    for order in (bdpt.ORDER_ADAPTIVE, bdpt.ORDER_INDEX, bdpt.ORDER_ROULETTE_CLASS):
        ctx, st, p = _pair(bdpt, oracle, scene, W, H, seed=61, rng_mode=bdpt.RNG_STRIDED, accum_mode=bdpt.ACCUM_SUM,
                           kernel_variant=variant, sample_order=order)
        ctx.set_sample_offset(7)
        ctx.render(SPP)
        oracle.render(st, scene, scene.camera, p, SPP, sample_offset=7, n_threads=8)
        _compare(bdpt, ctx, st, "%s order %d:" % (which, order))
        res[order] = ctx.read_accum().view(np.uint32)
        ctx.close()
    # the first sample index alone, to see which pixels it ends at the eye ray
    ctx, st, p = _pair(bdpt, oracle, scene, W, H, seed=61, rng_mode=bdpt.RNG_STRIDED, accum_mode=bdpt.ACCUM_SUM,
                       kernel_variant=variant, sample_order=bdpt.ORDER_INDEX)
    ctx.set_sample_offset(7)
    ctx.render(1)
    eye = ctx.read_aov(bdpt.AOV_EYE_HIT)
    ctx.close()
    early = (eye[..., 0] < 0) | ((eye[..., 0] == 0) & (eye[..., 1] == 0))     # missed, or saw the light (sphere 0)
    assert early.any() and (~early).any()
    assert np.array_equal(res[bdpt.ORDER_ADAPTIVE][early], res[bdpt.ORDER_INDEX][early])
    assert np.array_equal(res[bdpt.ORDER_ADAPTIVE][~early], res[bdpt.ORDER_ROULETTE_CLASS][~early])


def test_strided_mode_refuses_subpaths_whose_draws_would_overlap(bdpt):
    """STRIDED spaces samples 32 draws apart; a sample draws 5 + 4 per vertex (5 + 5 with REFLECTIVE), the first
    discarded: 7 vertices (5 REFLECTIVE) is the most that keeps the numbers of neighbouring samples apart."""
    ctx = bdpt.Context(16, 16)
    ctx.set_params(bdpt.default_params(rng_mode=bdpt.RNG_STRIDED, max_light_bounces=7))
    ctx.set_params(bdpt.default_params(rng_mode=bdpt.RNG_STRIDED, max_light_bounces=5, sampler=bdpt.SAMPLER_REFLECTIVE))
    ctx.set_params(bdpt.default_params(max_light_bounces=40))                       # STREAM has no such limit
    with pytest.raises(bdpt.BdptError, match="STRIDED"):
        ctx.set_params(bdpt.default_params(rng_mode=bdpt.RNG_STRIDED, max_light_bounces=8))
    with pytest.raises(bdpt.BdptError, match="STRIDED"):
        ctx.set_params(bdpt.default_params(rng_mode=bdpt.RNG_STRIDED, max_light_bounces=6, sampler=bdpt.SAMPLER_REFLECTIVE))
    ctx.close()


@pytest.mark.parametrize("scale", [0.5, 2.0])
def test_tiled_cull_with_plane_normals_that_are_not_unit_length(bdpt, oracle, scale):
    """The reference takes plane normals as given (primitives.cpp does not normalise them).  A bounce off a plane
    whose normal has length 0.5 or 2 starts a ray whose direction is not a unit vector; the tiled kernel's
    bounding-sphere test must not drop hits for such rays.  Six walls as planes plus small soup triangles."""
    rng = np.random.default_rng(5)
    sc = bdpt.Scene.cornell_planes()
    sc.planes = sc.planes.copy()
    sc.planes["normal"][:, :3] *= np.float32(scale)
    n = 400
    tri = np.zeros(n, bdpt.triangle_dtype)
    centre = rng.uniform(-2.7, 2.7, (n, 3))
    tri["p0"][:, :3] = centre
    tri["p1"][:, :3] = centre + rng.uniform(-0.15, 0.15, (n, 3))
    tri["p2"][:, :3] = centre + rng.uniform(-0.15, 0.15, (n, 3))
    tri["surface"] = rng.choice([0, 1, 3, 4], n)
    sc.triangles = np.concatenate([sc.triangles, tri])
    W, H = 56, 40
    for variant in (bdpt.KERNEL_TILED, bdpt.KERNEL_BVH, bdpt.KERNEL_DIRECT):
        ctx, st, p = _pair(bdpt, oracle, sc, W, H, seed=8, kernel_variant=variant)
        ctx.render(6)
        oracle.render(st, sc, sc.camera, p, 6, n_threads=8)
        _compare(bdpt, ctx, st, "normals x%g variant %d:" % (scale, variant))
        ctx.close()
    assert (st.rays > 6 * 3).any()          # some paths did bounce off the walls


def test_roulette_class_order_needs_strided_sum(bdpt):
    ctx = bdpt.Context(16, 16)
    for kw in (dict(), dict(rng_mode=bdpt.RNG_STRIDED), dict(accum_mode=bdpt.ACCUM_SUM)):
        with pytest.raises(bdpt.BdptError, match="sample_order"):
            ctx.set_params(bdpt.default_params(sample_order=bdpt.ORDER_ROULETTE_CLASS, **kw))
    ctx.set_params(bdpt.default_params(sample_order=bdpt.ORDER_ROULETTE_CLASS, rng_mode=bdpt.RNG_STRIDED, accum_mode=bdpt.ACCUM_SUM))
    ctx.close()


@pytest.mark.parametrize("variant", VARIANTS)
def test_region_renders_only_its_pixels(bdpt, oracle, variant):
    scene = bdpt.Scene.cornell()
    W, H = 64, 64
    ctx, st, p = _pair(bdpt, oracle, scene, W, H, seed=21, kernel_variant=variant)
    region = (19, 7, 30, 41)   # deliberately not a multiple of the 8x8 wavefront tile
    ctx.set_region(*region)
    ctx.render(3)
    oracle.render(st, scene, scene.camera, p, 3, region=region, n_threads=8)
    _compare(bdpt, ctx, st, "region:")
    acc = ctx.read_accum()
    mask = np.ones((H, W), bool)
    mask[7:48, 19:49] = False
    assert not acc[mask].any()
    ctx.close()


def test_camera_move_resets_accumulation(bdpt, oracle):
    scene = bdpt.Scene.cornell()
    W, H = 48, 48
    ctx, st, p = _pair(bdpt, oracle, scene, W, H, seed=31)
    ctx.render(4)
    oracle.render(st, scene, scene.camera, p, 4, n_threads=8)
    cams, changed = bdpt.camera_orbit(8.0, (0, -1, 0), (1.0, 0.1, -0.1), 0.03, 0.03, 0.5 / 0.03, 0.0, 2)
    assert changed.all()
    ctx.set_camera(cams[2])
    assert ctx.num_samples == 0
    st.num_samples = 0
    st.reset_aovs()
    ctx.render(2)
    oracle.render(st, scene, cams[2], p, 2, n_threads=8)
    _compare(bdpt, ctx, st, "after move:")
    ctx.close()


def test_error_paths(bdpt):
    ctx = bdpt.Context(16, 16)
    with pytest.raises(bdpt.BdptError) as e:
        ctx.render(1)
    assert e.value.code == -3
    with pytest.raises(bdpt.BdptError):
        ctx.set_params(max_light_bounces=0)
    with pytest.raises(bdpt.BdptError):
        ctx.set_region(8, 8, 16, 16)
    scene = bdpt.Scene.cornell()
    bad = bdpt.Scene.cornell()
    bad.triangles["surface"][3] = 99
    with pytest.raises(bdpt.BdptError):
        ctx.upload_scene(bad)
    # a fourth component other than 0 (homogeneous w = 1) would change the reference's four-component dot and
    # distance (kernel0.cl:331, :336, :398): refused, not dropped
    for arr, field in (("spheres", "center"), ("triangles", "p0"), ("triangles", "p1"), ("triangles", "p2")):
        bad = bdpt.Scene.cornell()
        getattr(bad, arr)[field][1, 3] = 1.0
        with pytest.raises(bdpt.BdptError, match="fourth component"):
            ctx.upload_scene(bad)
    for field in ("normal", "point"):
        bad = bdpt.Scene.cornell_planes()
        bad.planes[field][0, 3] = 1.0
        with pytest.raises(bdpt.BdptError, match="fourth component"):
            ctx.upload_scene(bad)
    ok = bdpt.Scene.cornell()
    ok.spheres["center"][:, 3] = -0.0
    ctx.upload_scene(ok)
    ctx.upload_scene(scene)
    ctx.upload_seeds(bdpt.make_seeds(16, 16))
    for field in ("pos", "look_at", "up"):
        cam = scene.camera.copy()
        cam[field][3] = 1.0
        with pytest.raises(bdpt.BdptError, match="must be 0"):
            ctx.set_camera(cam)
    ctx.set_camera(scene.camera)
    with pytest.raises(bdpt.BdptError):
        ctx.render(0)
    ctx.render(1)
    ctx.close()
    with pytest.raises(bdpt.BdptError):
        bdpt.Context(16, 16, device=4096)


def test_lds_variant_refuses_a_scene_that_does_not_fit(bdpt):
    scene = bdpt.Scene.soup(3000)
    ctx = bdpt.Context(16, 16)
    ctx.set_params(kernel_variant=bdpt.KERNEL_LDS)
    ctx.upload_scene(scene)
    ctx.upload_seeds(bdpt.make_seeds(16, 16))
    ctx.set_camera(scene.camera)
    with pytest.raises(bdpt.BdptError):
        ctx.render(1)
    ctx.close()

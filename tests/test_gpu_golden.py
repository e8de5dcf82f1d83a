"""GPU path against the committed golden fixtures (no oracle involved at run time)."""
import numpy as np
import pytest

import golden_util as gu

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("variant", [0, 1, 3, 4, 5])
@pytest.mark.parametrize("case", gu.RENDER_CASES)
def test_render_matches_golden(bdpt, case, variant):
    g = gu.load(case)
    scene, p, W, H, spp, off = gu.unpack(bdpt, g)
    p.kernel_variant = variant
    if variant in (bdpt.KERNEL_LDS, bdpt.KERNEL_LDS2) and len(scene.triangles) * (96 + (80 if variant == bdpt.KERNEL_LDS2 else 0)) > 20000:
        pytest.skip("scene too large for the LDS-resident kernels")
    with bdpt.Context(W, H) as ctx:
        ctx.set_params(p)
        ctx.upload_scene(scene)
        ctx.upload_seeds(g["seeds"])
        ctx.set_camera(scene.camera)
        ctx.set_sample_offset(off)
        for s in range(spp):
            ctx.render(1)
            assert np.array_equal(ctx.read_aov(bdpt.AOV_EYE_HIT), g["out_eye_hit"][s])
            assert np.array_equal(ctx.read_aov(bdpt.AOV_HIT_HASH), g["out_hit_hash"][s])
            assert np.array_equal(ctx.read_aov(bdpt.AOV_RNG_DRAWS), g["out_rng_draws"][s])
            assert np.array_equal(ctx.read_aov(bdpt.AOV_RAYS), g["out_rays"][s])
            rad = ctx.read_aov(bdpt.AOV_RADIANCE)
            np.testing.assert_allclose(rad, g["out_radiance"][s], rtol=1e-5, atol=1e-7)
            assert np.array_equal(rad.view(np.uint32), g["out_radiance"][s].view(np.uint32))
            assert np.array_equal(ctx.read_accum().view(np.uint32), g["out_accum"][s].view(np.uint32))
        assert np.array_equal(ctx.read_seeds(), g["seeds_after"])


def test_closest_hit_matches_golden(bdpt):
    g = gu.load("closest_hit_cornell")
    with bdpt.Context(8, 8) as ctx:
        ctx.upload_scene(bdpt.Scene.cornell())
        ids, pos, nrm = ctx.query_rays(np.concatenate([g["origins"], g["dirs"]], 1), g["origin_ids"])
    assert np.array_equal(ids, g["ids"])
    hit = ids[:, 0] >= 0
    assert np.array_equal(pos[hit].view(np.uint32), g["pos"][hit].view(np.uint32))
    assert np.array_equal(nrm[hit].view(np.uint32), g["normals"][hit].view(np.uint32))


def _region_variants(case):
    # the 100 000-triangle scene does not fit the LDS-resident kernels; AUTO must resolve to the hierarchy there
    return [0, 1, 4, 6] if "soup100k" in case else [0, 1, 3, 4, 5, 6]


@pytest.mark.parametrize("case,variant", [(c, v) for c in gu.REGION_CASES for v in _region_variants(c)])
def test_region_of_the_1080p_frame_matches_golden(bdpt, case, variant):
    """Regions of the 1920x1080 frame of the benchmark configurations (Cornell box 64x64 x 16 spp; Cornell box +
    100 000-triangle soup 32x32 x 2 spp), one launch per sample: every output of every sample against the fixture,
    pixels outside the region untouched."""
    g = gu.load(case)
    scene, p, W, H, spp, region, seed = gu.unpack_region(bdpt, g)
    x0, y0, w, h = region
    p.kernel_variant = variant
    crop = lambda a: a[y0:y0 + h, x0:x0 + w]
    with bdpt.Context(W, H) as ctx:
        ctx.set_params(p)
        ctx.upload_scene(scene)
        ctx.upload_seeds(bdpt.make_seeds(W, H, seed))
        ctx.set_camera(scene.camera)
        ctx.set_region(*region)
        for s in range(spp):
            ctx.render(1)
            assert np.array_equal(crop(ctx.read_aov(bdpt.AOV_EYE_HIT)), g["out_eye_hit"][s])
            assert np.array_equal(crop(ctx.read_aov(bdpt.AOV_HIT_HASH)), g["out_hit_hash"][s])
            assert np.array_equal(crop(ctx.read_aov(bdpt.AOV_RNG_DRAWS)), g["out_rng_draws"][s])
            assert np.array_equal(crop(ctx.read_aov(bdpt.AOV_RAYS)), g["out_rays"][s])
            assert np.array_equal(crop(ctx.read_aov(bdpt.AOV_RADIANCE)).view(np.uint32), g["out_radiance"][s].view(np.uint32))
            assert np.array_equal(crop(ctx.read_accum()).view(np.uint32), g["out_accum"][s].view(np.uint32))
        if "soup100k" in case:
            assert ctx.last_launch_variant() == {0: bdpt.KERNEL_BVH, 1: bdpt.KERNEL_DIRECT, 4: bdpt.KERNEL_TILED, 6: bdpt.KERNEL_BVH}[variant]
        seeds_after = ctx.read_seeds()
        xs, ys = np.meshgrid(np.arange(x0, x0 + w), np.arange(y0, y0 + h))
        assert np.array_equal(seeds_after[(xs * H + ys) * 4 + 3], g["seed_words_after"])
        acc = ctx.read_accum()
        acc[y0:y0 + h, x0:x0 + w] = 0
        assert not acc.any()


@pytest.mark.parametrize("case,variant", [(c, v) for c in gu.LAUNCH_CASES for v in [0, 1, 3, 4]])
def test_region_of_the_4k_outdoor_frame_matches_golden(bdpt, case, variant):
    """The outdoor sphere arrangement on its 3840x2160 frame as one rank of an 8-way sample split renders it (STRIDED +
    SUM, one launch of 8 samples, first sample index 0 or 896): 64x64 regions on a sphere's silhouette, where four
    spheres meet, and in the frame's bottom-right corner.  Pixel coordinates with the fp64 aspect at 3840/2160 and the
    seed index x*2160 + y (kernel0.cl:590, :600-604); AUTO must take the LDS kernel (no triangles)."""
    g = gu.load(case)
    scene, p, W, H, spp, region, seed = gu.unpack_region(bdpt, g)
    assert (W, H) == (3840, 2160) and p.rng_mode == bdpt.RNG_STRIDED and p.accum_mode == bdpt.ACCUM_SUM
    x0, y0, w, h = region
    p.kernel_variant = variant
    crop = lambda a: a[y0:y0 + h, x0:x0 + w]
    seeds = bdpt.make_seeds(W, H, seed)
    with bdpt.Context(W, H) as ctx:
        ctx.set_params(p)
        ctx.upload_scene(scene)
        ctx.upload_seeds(seeds)
        ctx.set_camera(scene.camera)
        ctx.set_region(*region)
        ctx.set_sample_offset(int(g["sample_offset"]))
        ctx.render(spp)
        if variant == 0:
            assert ctx.last_launch_variant() == bdpt.KERNEL_LDS
        assert np.array_equal(crop(ctx.read_aov(bdpt.AOV_EYE_HIT)), g["out_eye_hit"])
        assert np.array_equal(crop(ctx.read_aov(bdpt.AOV_HIT_HASH)), g["out_hit_hash"])
        assert np.array_equal(crop(ctx.read_aov(bdpt.AOV_RNG_DRAWS)), g["out_rng_draws"])
        assert np.array_equal(crop(ctx.read_aov(bdpt.AOV_RAYS)), g["out_rays"])
        assert np.array_equal(crop(ctx.read_aov(bdpt.AOV_RADIANCE)).view(np.uint32), g["out_radiance"].view(np.uint32))
        acc = ctx.read_accum()
        assert np.array_equal(crop(acc).view(np.uint32), g["out_accum"].view(np.uint32))
        xs, ys = np.meshgrid(np.arange(x0, x0 + w), np.arange(y0, y0 + h))
        assert np.array_equal(ctx.read_seeds()[(xs * H + ys) * 4 + 3], g["seed_words_after"])
        assert np.array_equal(g["seed_words_after"], seeds[(xs * H + ys) * 4 + 3])      # STRIDED: the seeds stay
        acc[y0:y0 + h, x0:x0 + w] = 0
        assert not acc.any()

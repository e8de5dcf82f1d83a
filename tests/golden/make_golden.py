"""Regenerates the golden fixtures in this directory with the CPU oracle.

    python tests/golden/make_golden.py

PARITY UNPINNED: the reference ships no fixtures and cannot be executed in this
environment, so these vectors are outputs of the oracle (oracle/bdpt_oracle.c),
not of the reference.  They pin the oracle against regressions and let the GPU
path be checked without the oracle present.  Each .npz holds the inputs (scene
arrays, camera, seeds, parameter fields) and the expected outputs.
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, os.path.join(ROOT, "opencl-path-tracer_amd"))
sys.path.insert(0, os.path.join(ROOT, "oracle"))

import bdpt  # noqa: E402  (host-side scene/camera/seed helpers only; no GPU call is made)
import oracle_bind as oracle  # noqa: E402

PARAM_FIELDS = [f for f, _ in bdpt.Params._fields_]


def params_dict(p):
    return {"param_" + f: np.asarray(getattr(p, f)) for f in PARAM_FIELDS}


def render_case(name, scene, W, H, spp, seed, sample_offset=0, camera=None, **pkw):
    cam = scene.camera if camera is None else camera
    seeds = bdpt.make_seeds(W, H, seed)
    p = bdpt.default_params(aov_enable=1, **pkw)
    st = oracle.State(W, H, seeds)
    per = {k: [] for k in ("eye_hit", "hit_hash", "radiance", "accum", "rng_draws", "rays")}
    for _ in range(spp):
        oracle.render(st, scene, cam, p, 1, sample_offset=sample_offset, n_threads=4)
        per["eye_hit"].append(st.eye_hit.copy())
        per["hit_hash"].append(st.hit_hash.copy())
        per["radiance"].append(st.radiance.copy())
        per["accum"].append(st.accum.copy())
        per["rng_draws"].append(st.rng_draws.copy())
        per["rays"].append(st.rays.copy())
    out = dict(width=np.asarray(W), height=np.asarray(H), spp=np.asarray(spp), sample_offset=np.asarray(sample_offset),
               spheres=scene.spheres, planes=scene.planes, triangles=scene.triangles, surfaces=scene.surfaces,
               camera=np.asarray(cam), seeds=seeds, seeds_after=st.seeds, **params_dict(p))
    for k, v in per.items():
        out["out_" + k] = np.stack(v)
    np.savez_compressed(os.path.join(HERE, name + ".npz"), **out)
    return st, per


def region_seed_words(seeds, W, H, region):
    """The generator words (word 3 of each uint4, index x*H + y) of the pixels of a region, as an (h, w) array."""
    x0, y0, w, h = region
    xs, ys = np.meshgrid(np.arange(x0, x0 + w), np.arange(y0, y0 + h))
    return seeds[(xs * H + ys) * 4 + 3]


def render_region_case(name, scene, recipe, W, H, region, spp, seed, **pkw):
    """A region of a large frame (the frame sizes of the benchmark configurations): the per-pixel state lives in
    the full W x H frame, only the region is rendered, and the fixture keeps the region's crop of every output
    after each of `spp` one-sample launches.  The scene is named by its recipe (kind, soup triangles, soup seed)
    instead of being stored: the C++ host builds it deterministically."""
    x0, y0, w, h = region
    seeds = bdpt.make_seeds(W, H, seed)
    p = bdpt.default_params(aov_enable=1, **pkw)
    st = oracle.State(W, H, seeds)
    crop = lambda a: a[y0:y0 + h, x0:x0 + w].copy()
    per = {k: [] for k in ("eye_hit", "hit_hash", "radiance", "accum", "rng_draws", "rays")}
    for _ in range(spp):
        oracle.render(st, scene, scene.camera, p, 1, region=region, n_threads=8)
        for k, a in (("eye_hit", st.eye_hit), ("hit_hash", st.hit_hash), ("radiance", st.radiance), ("accum", st.accum),
                     ("rng_draws", st.rng_draws), ("rays", st.rays)):
            per[k].append(crop(a))
    out = dict(width=np.asarray(W), height=np.asarray(H), spp=np.asarray(spp), region=np.asarray(region),
               global_seed=np.asarray(seed), scene_kind=np.asarray(recipe[0]), n_soup=np.asarray(recipe[1]),
               soup_seed=np.asarray(recipe[2]), camera=np.asarray(scene.camera),
               seed_words_after=region_seed_words(st.seeds, W, H, region), **params_dict(p))
    for k, v in per.items():
        out["out_" + k] = np.stack(v)
    np.savez_compressed(os.path.join(HERE, name + ".npz"), **out)
    return st


def region_cases():
    """Regions of the 1920x1080 frame of the benchmark configurations: pixel coordinates (fp64 aspect at 1920/1080,
    kernel0.cl:600-604) and the column-major seed index (kernel0.cl:590) at that size."""
    cornell = bdpt.Scene.cornell()
    W, H = 1920, 1080
    # Cornell box at 1080p, 64x64 x 16 spp: around the light sphere, on the box's silhouette, bottom-right corner
    st = render_region_case("region_c2_light_928_700", cornell, (bdpt.SCENE_CORNELL, 0, 0), W, H, (928, 700, 64, 64), 16, 0x5EED)
    eye = st.eye_hit[700:764, 928:992]
    assert ((eye[..., 0] == 0) & (eye[..., 1] == 0)).any() and (eye[..., 0] == 2).any()
    st = render_region_case("region_c2_silhouette_1616_1000", cornell, (bdpt.SCENE_CORNELL, 0, 0), W, H, (1616, 1000, 64, 64), 16, 0x5EED)
    eye = st.eye_hit[1000:1064, 1616:1680]
    assert (eye[..., 0] < 0).any() and (eye[..., 0] == 2).any()
    render_region_case("region_c2_corner_1856_1016", cornell, (bdpt.SCENE_CORNELL, 0, 0), W, H, (1856, 1016, 64, 64), 16, 0x5EED)
    # Cornell box + 100 000-triangle soup at 1080p, 32x32 x 2 spp
    soup = bdpt.Scene.soup(100000, 0xC0FFEE)
    assert len(soup.triangles) == 100012
    st = render_region_case("region_c4_soup100k_944_524", soup, (bdpt.SCENE_SOUP, 100000, 0xC0FFEE), W, H, (944, 524, 32, 32), 2, 0x5EED)
    eye = st.eye_hit[524:556, 944:976]
    assert (eye[..., 1] >= 12).any(), "the region must see soup triangles"


def render_launch_case(name, scene, recipe, W, H, region, spp, seed, sample_offset, **pkw):
    """A region of a large frame rendered as ONE launch of `spp` samples (the order in which a STRIDED + SUM launch
    adds its samples is part of the result), with a sample offset as a rank of a sample split has.  Keeps the
    region's crop of every output after the launch."""
    x0, y0, w, h = region
    seeds = bdpt.make_seeds(W, H, seed)
    p = bdpt.default_params(aov_enable=1, **pkw)
    st = oracle.State(W, H, seeds)
    oracle.render(st, scene, scene.camera, p, spp, region=region, sample_offset=sample_offset, n_threads=8)
    crop = lambda a: a[y0:y0 + h, x0:x0 + w].copy()
    out = dict(width=np.asarray(W), height=np.asarray(H), spp=np.asarray(spp), region=np.asarray(region),
               sample_offset=np.asarray(sample_offset), global_seed=np.asarray(seed), scene_kind=np.asarray(recipe[0]),
               n_soup=np.asarray(recipe[1]), soup_seed=np.asarray(recipe[2]), camera=np.asarray(scene.camera),
               seed_words_after=region_seed_words(st.seeds, W, H, region), out_eye_hit=crop(st.eye_hit),
               out_hit_hash=crop(st.hit_hash), out_radiance=crop(st.radiance), out_accum=crop(st.accum),
               out_rng_draws=crop(st.rng_draws), out_rays=crop(st.rays), **params_dict(p))
    np.savez_compressed(os.path.join(HERE, name + ".npz"), **out)
    return st


def c3_cases():
    """The outdoor sphere arrangement on its own 3840x2160 frame, as one rank of an 8-way sample split renders it:
    STRIDED + SUM, one launch of 8 samples, first sample index 0 (rank 0) and 896 (rank 7 of 8 x 128).  64x64 regions:
    on the upper silhouette of a sphere, where four spheres meet around a hole the light shows through, and the frame's bottom-right
    corner (every eye ray leaves the scene; x*2160 + y at its largest)."""
    outdoor = bdpt.Scene.outdoor()
    W, H = 3840, 2160
    kw = dict(rng_mode=bdpt.RNG_STRIDED, accum_mode=bdpt.ACCUM_SUM)
    for label, region in (("silhouette_1712_528", (1712, 528, 64, 64)), ("junction_1956_1060", (1956, 1060, 64, 64)),
                          ("corner_3776_2096", (3776, 2096, 64, 64))):
        for off in (0, 896):
            st = render_launch_case("launch4k_c3_%s_from%d" % (label, off), outdoor, (bdpt.SCENE_OUTDOOR, 0, 0), W, H, region,
                                    8, 0x5EED, off, **kw)
            x0, y0, w, h = region
            eye = st.eye_hit[y0:y0 + h, x0:x0 + w]
            if label.startswith("silhouette"):
                assert (eye[..., 0] < 0).any() and (eye[..., 0] == 0).any()
            elif label.startswith("junction"):
                assert len(np.unique(eye[..., 1][eye[..., 0] == 0])) >= 3 and st.accum[y0:y0 + h, x0:x0 + w].max() > 0
            else:
                assert (eye[..., 0] < 0).all()


def main():
    if "--c3-only" in sys.argv:
        c3_cases()
        return
    if "--regions-only" not in sys.argv:
        small_cases()
    region_cases()
    c3_cases()
    print("fixtures written to", HERE)


def small_cases():
    # generator known answers
    seeds = np.array([1, 12345, 0x7FFFFFFF], np.uint32)
    fl, wd = zip(*[oracle.rng_sequence(int(s), 32) for s in seeds])
    np.savez_compressed(os.path.join(HERE, "rng_kat.npz"), seeds=seeds, floats=np.stack(fl), words=np.stack(wd))

    cornell = bdpt.Scene.cornell()
    st, per = render_case("cornell_64x36_4spp", cornell, 64, 36, 4, 0x5EED)
    eye = per["eye_hit"][-1]
    assert ((eye[..., 0] == 0) & (eye[..., 1] == 0)).any(), "fixture must contain a pixel that sees the light directly"
    assert (eye[..., 0] == -1).any(), "fixture must contain a pixel whose eye ray misses"
    render_case("cornell_strided_sum_32x32_3spp_from5", cornell, 32, 32, 3, 77, sample_offset=5,
                rng_mode=bdpt.RNG_STRIDED, accum_mode=bdpt.ACCUM_SUM)
    render_case("cornell_variants_24x24_3spp", cornell, 24, 24, 3, 5, light_origin=bdpt.LIGHT_ORIGIN_NONE,
                connection_falloff=bdpt.FALLOFF_INV_SQ_PLUS_C, falloff_c=0.5, indirect_only=1, max_light_bounces=3,
                p_continue=0.6, blur_factor=1.0)
    render_case("planes_24x24_3spp", bdpt.Scene.cornell_planes(), 24, 24, 3, 9)
    render_case("outdoor_32x24_3spp", bdpt.Scene.outdoor(), 32, 24, 3, 13)
    render_case("soup500_16x16_2spp", bdpt.Scene.soup(500), 16, 16, 2, 21)
    cams, _ = bdpt.camera_orbit(8.0, (0, -1, 0), (1.0, 0.1, -0.1), 0.03, 0.03, 0.5 / 0.03, 0.25 / 0.03, 7)
    render_case("cornell_moved_camera_24x24_2spp", cornell, 24, 24, 2, 3, camera=cams[7])

    # closest-hit regression vectors on the triangle box
    rng = np.random.default_rng(123)
    n = 256
    o = rng.uniform(-2.9, 2.9, (n, 3)).astype(np.float32)
    d = rng.standard_normal((n, 3)).astype(np.float32)
    d /= np.linalg.norm(d, axis=1, keepdims=True).astype(np.float32)
    org = np.tile(np.array([4, 0], np.int32), (n, 1))
    org[:64] = [2, 5]
    ids, pos, nrm = [], [], []
    for i in range(n):
        _, a, b, c = oracle.ray_intersect(cornell, o[i], d[i], org[i])
        ids.append(a.copy()); pos.append(b.copy()); nrm.append(c.copy())
    np.savez_compressed(os.path.join(HERE, "closest_hit_cornell.npz"), origins=o, dirs=d, origin_ids=org,
                        ids=np.stack(ids), pos=np.stack(pos), normals=np.stack(nrm))


if __name__ == "__main__":
    main()

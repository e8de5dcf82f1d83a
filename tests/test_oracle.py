"""CPU tests of the oracle: analytic known answers for every function on the
path, the built-ins against libm, invariants of the sample loop, and the
committed golden fixtures.  (The reference ships no vectors of its own, so none
can be checked here: parity unpinned.)"""
import math

import numpy as np
import pytest

import golden_util as gu


def ulp_diff32(a, b):
    a = np.asarray(a, np.float32).view(np.int32).astype(np.int64)
    b = np.asarray(b, np.float32).view(np.int32).astype(np.int64)
    a = np.where(a < 0, -(a & 0x7FFFFFFF), a)
    b = np.where(b < 0, -(b & 0x7FFFFFFF), b)
    return np.abs(a - b)


# ---- generator: kernel0.cl:106-112 -------------------------------------------

@pytest.mark.parametrize("seed", [1, 12345, 0x7FFFFFFF, 0xFFFFFFFF])
def test_rng_is_the_48271_multiplicative_generator(oracle, seed):
    floats, words = oracle.rng_sequence(seed, 40)
    w = seed
    for k in range(40):
        w = (w * 48271) & 0xFFFFFFFF
        assert int(words[k]) == w
        assert floats[k] == np.float32(np.float32(w) / np.float32(4294967296.0))
        assert 0.0 <= floats[k] <= 1.0


def test_rng_golden(oracle):
    g = gu.load("rng_kat")
    for s, fl, wd in zip(g["seeds"], g["floats"], g["words"]):
        f2, w2 = oracle.rng_sequence(int(s), 32)
        assert np.array_equal(w2, wd) and np.array_equal(f2.view(np.uint32), fl.view(np.uint32))


# ---- built-ins against libm, within OpenCL's bounds ---------------------------

def test_sincos_within_2ulp_on_the_kernel_domain(oracle):
    xs = np.concatenate([np.linspace(0, 2 * np.pi, 20001), [0.0, np.pi / 2, np.pi, 1.5 * np.pi, 6.2831855]]).astype(np.float32)
    got = np.array([oracle.sincos(x) for x in xs], np.float32)
    ref_s = np.sin(xs.astype(np.float64))
    ref_c = np.cos(xs.astype(np.float64))
    # absolute error near the zeros of sin/cos, ulp elsewhere
    assert np.all((ulp_diff32(got[:, 0], ref_s.astype(np.float32)) <= 2) | (np.abs(got[:, 0] - ref_s) < 2e-7))
    assert np.all((ulp_diff32(got[:, 1], ref_c.astype(np.float32)) <= 2) | (np.abs(got[:, 1] - ref_c) < 2e-7))


def test_acosf_within_4ulp(oracle):
    xs = np.concatenate([np.linspace(-1, 1, 20001), [0.5, -0.5, 0.50000006, 1.0, -1.0, 0.0]]).astype(np.float32)
    got = np.array([oracle.acosf(x) for x in xs], np.float32)
    ref = np.arccos(xs.astype(np.float64))
    assert np.all((ulp_diff32(got, ref.astype(np.float32)) <= 4) | (np.abs(got - ref) < 3e-7))
    assert oracle.acosf(1.0) == 0.0


def test_acos_f64_matches_libm(oracle):
    xs = np.concatenate([np.linspace(-1, 1, 20001), [0.5, -0.5, 1.0, -1.0, 1 - 1e-12, -1 + 1e-12]])
    got = np.array([oracle.acos_f64(x) for x in xs])
    ref = np.arccos(xs)
    assert np.max(np.abs(got - ref)) < 1e-15 * 8
    assert oracle.acos_f64(1.0) == 0.0 and abs(oracle.acos_f64(-1.0) - math.pi) < 1e-15


# ---- intersectors: kernel0.cl:232-361 -----------------------------------------

def _sphere(bdpt, c, r):
    s = np.zeros(1, bdpt.sphere_dtype)
    s["center"][0, :3] = c
    s["radius"] = r
    return s


def _tri(bdpt, a, b, c):
    t = np.zeros(1, bdpt.triangle_dtype)
    t["p0"][0, :3], t["p1"][0, :3], t["p2"][0, :3] = a, b, c
    return t


def _plane(bdpt, n, pt):
    p = np.zeros(1, bdpt.plane_dtype)
    p["normal"][0, :3], p["point"][0, :3] = n, pt
    return p


def test_sphere_known_answers(bdpt, oracle):
    s = _sphere(bdpt, (0, 0, 0), 1.0)
    hit, pos = oracle.ray_primitive("sphere", s, (0, 0, -5), (0, 0, 1))
    assert hit and np.allclose(pos, (0, 0, -1), atol=1e-6)
    # pointing away: dot(d, o-c) > 0 (kernel0.cl:333)
    assert not oracle.ray_primitive("sphere", s, (0, 0, -5), (0, 0, -1))[0]
    # origin inside: the near root is behind the origin (kernel0.cl:350)
    assert not oracle.ray_primitive("sphere", s, (0, 0, 0.5), (0, 0, -1))[0]
    assert not oracle.ray_primitive("sphere", s, (0, 0, 0), (0, 0, 1))[0]
    # miss: negative discriminant (kernel0.cl:341)
    assert not oracle.ray_primitive("sphere", s, (0, 2, -5), (0, 0, 1))[0]
    # grazing: discriminant exactly 0 is accepted
    hit, pos = oracle.ray_primitive("sphere", s, (0, 1, -5), (0, 0, 1))
    assert hit and np.allclose(pos, (0, 1, 0), atol=1e-6)


def test_plane_known_answers(bdpt, oracle):
    p = _plane(bdpt, (0, 1, 0), (0, -3, 0))
    hit, pos = oracle.ray_primitive("plane", p, (1, 0, 2), (0, -1, 0))
    assert hit and np.allclose(pos, (1, -3, 2))
    # back face culled, and a ray along the plane is culled too: angle >= 0 (kernel0.cl:252)
    assert not oracle.ray_primitive("plane", p, (1, -5, 2), (0, 1, 0))[0]
    assert not oracle.ray_primitive("plane", p, (1, 0, 2), (1, 0, 0))[0]
    # facing the plane from behind it: desc <= 0 (kernel0.cl:258)
    assert not oracle.ray_primitive("plane", p, (1, -5, 2), (0, -1, 0))[0]


def test_triangle_known_answers(bdpt, oracle):
    # normal = normalize(cross(b-a, c-a)) = +z; rays travelling along -z see its front
    t = _tri(bdpt, (0, 0, 0), (1, 0, 0), (0, 1, 0))
    hit, pos = oracle.ray_primitive("triangle", t, (0.25, 0.25, 1), (0, 0, -1))
    assert hit and np.allclose(pos, (0.25, 0.25, 0))
    assert not oracle.ray_primitive("triangle", t, (0.25, 0.25, -1), (0, 0, 1))[0]      # dot(d, n) > 0 culled (:288)
    assert not oracle.ray_primitive("triangle", t, (0.25, 0.25, -1), (0, 0, -1))[0]     # t < 0 (:292)
    assert not oracle.ray_primitive("triangle", t, (0.8, 0.8, 1), (0, 0, -1))[0]        # outside (:304)
    # on an edge and on a vertex the edge products are exactly 0, which is inside (:304 uses '< 0')
    assert oracle.ray_primitive("triangle", t, (0.5, 0.0, 1), (0, 0, -1))[0]
    assert oracle.ray_primitive("triangle", t, (0.0, 0.0, 1), (0, 0, -1))[0]
    assert oracle.ray_primitive("triangle", t, (0.5, 0.5, 1), (0, 0, -1))[0]
    # a ray in the triangle's plane divides by zero and can never be kept
    hit, pos = oracle.ray_primitive("triangle", t, (-1, 0.25, 0), (1, 0, 0))
    assert (not hit) or not np.isfinite(pos).all()


def test_degenerate_triangle_is_never_hit(bdpt, oracle):
    sc = bdpt.Scene.cornell()
    sc.triangles = np.concatenate([sc.triangles, _tri(bdpt, (0, 0, 0), (1, 1, 1), (2, 2, 2))])
    sc.triangles["surface"][-1] = 1
    rng = np.random.default_rng(0)
    for _ in range(200):
        d = rng.standard_normal(3)
        d /= np.linalg.norm(d)
        _, ids, _, _ = oracle.ray_intersect(sc, rng.uniform(-2, 2, 3), d)
        assert tuple(ids) != (2, 12)


# ---- ray_intersect: kernel0.cl:365-478 ----------------------------------------

def test_tie_break_keeps_the_lowest_index(bdpt, oracle):
    """(-3,0,0) lies on the diagonal shared by triangles 0 and 1 of the x=-3 wall:
    both report a hit at the same distance and strict '<' keeps triangle 0."""
    sc = bdpt.Scene.cornell()
    hit, ids, pos, nrm = oracle.ray_intersect(sc, (0, 0, 0), (-1, 0, 0))
    assert hit and tuple(ids) == (2, 0) and np.allclose(pos, (-3, 0, 0))
    assert oracle.ray_primitive("triangle", sc.triangles[0:1], (0, 0, 0), (-1, 0, 0))[0]
    assert oracle.ray_primitive("triangle", sc.triangles[1:2], (0, 0, 0), (-1, 0, 0))[0]
    # the normal faces the incoming ray (kernel0.cl:471-473)
    assert np.allclose(nrm, (1, 0, 0))
    # skipping triangle 0 as the origin primitive hands the hit to triangle 1 (kernel0.cl:436)
    hit, ids, _, _ = oracle.ray_intersect(sc, (0, 0, 0), (-1, 0, 0), (2, 0))
    assert hit and tuple(ids) == (2, 1)


def test_search_order_and_max_distance(bdpt, oracle):
    sc = bdpt.Scene.cornell()
    # straight down from the centre: the yellow sphere (index 1) is nearer than the floor
    hit, ids, pos, nrm = oracle.ray_intersect(sc, (0, 0, 0), (0, -1, 0))
    assert hit and tuple(ids) == (0, 1) and np.allclose(pos, (0, -1, 0), atol=1e-6)
    assert np.allclose(nrm, (0, 1, 0), atol=1e-6)
    # the camera sits outside the box and looks in through the culled +x wall
    cam = sc.camera
    hit, ids, _, _ = oracle.ray_intersect(sc, cam["pos"][:3], cam["look_at"][:3])
    assert hit and ids[0] in (0, 2) and not (ids[0] == 2 and ids[1] in (2, 3))
    # nothing beyond 10000 units is kept (kernel0.cl:374)
    far = bdpt.Scene.cornell()
    far.spheres["center"][:, 0] += 20000.0
    far.triangles = far.triangles[:0]
    assert not oracle.ray_intersect(far, (0, 0, 0), (1, 0, 0))[0]
    # every wall of the box faces inward: from the centre every direction hits something
    rng = np.random.default_rng(3)
    for _ in range(300):
        d = rng.standard_normal(3)
        d /= np.linalg.norm(d)
        assert oracle.ray_intersect(sc, (0.1, 0.2, -0.3), d)[0]


def _closest_hit_f64(sc, o, d):
    """Analytic closest hit in float64 with the reference's acceptance rules (kernel0.cl:232-361, :365-450), written
    with numpy independently of the C restatement: returns (type, index, distance) or (-1, -1, inf)."""
    o, d = np.asarray(o, np.float64), np.asarray(d, np.float64)
    best = (-1, -1, 1.0e4)
    for i, s in enumerate(sc.spheres):
        oc = o - s["center"][:3].astype(np.float64)
        b = d @ oc
        disc = b * b - oc @ oc + float(s["radius"]) ** 2
        if b > 0 or disc < 0:
            continue
        t = -b - np.sqrt(disc)
        if t >= 0 and t < best[2]:
            best = (0, i, t)
    for i, p in enumerate(sc.planes):
        n, pt = p["normal"][:3].astype(np.float64), p["point"][:3].astype(np.float64)
        den = d @ n
        if den >= 0:
            continue
        t = ((pt - o) @ n) / den
        if t > 0 and t < best[2]:
            best = (1, i, t)
    for i, tr in enumerate(sc.triangles):
        a, b, c = (tr[k][:3].astype(np.float64) for k in ("p0", "p1", "p2"))
        n = np.cross(b - a, c - a)
        if not n.any():
            continue
        n /= np.linalg.norm(n)
        den = d @ n
        if den > 0 or den == 0:
            continue
        t = ((a - o) @ n) / den
        if t < 0:
            continue
        q = o + t * d
        if min(np.cross(q - a, c - a) @ n, np.cross(q - c, b - c) @ n, np.cross(q - b, a - b) @ n) < 0:
            continue
        if t < best[2]:
            best = (2, i, t)
    return best if best[0] >= 0 else (-1, -1, np.inf)


@pytest.mark.parametrize("which", ["cornell", "cornell_planes", "soup"])
def test_closest_hit_agrees_with_float64_geometry(bdpt, oracle, which):
    """800 random rays per scene: the restated search and an independent float64 computation name the same primitive
    at the same point, except where float64 itself finds two candidates closer than 1e-4 (ties, edges, grazing hits)."""
    sc = bdpt.Scene.soup(120) if which == "soup" else getattr(bdpt.Scene, which)()
    rng = np.random.default_rng(11)
    disagree = 0
    for _ in range(800):
        o = rng.uniform(-2.8, 2.8, 3).astype(np.float32)
        d = rng.standard_normal(3)
        d = (d / np.linalg.norm(d)).astype(np.float32)
        hit, ids, pos, _ = oracle.ray_intersect(sc, o, d)
        ty, ix, t = _closest_hit_f64(sc, o, d)
        if hit and (ids[0], ids[1]) == (ty, ix):
            assert np.allclose(pos, o.astype(np.float64) + t * d.astype(np.float64), atol=2e-5)
        else:
            disagree += 1
    assert disagree <= 2, disagree


def test_closest_hit_golden(bdpt, oracle):
    g = gu.load("closest_hit_cornell")
    sc = bdpt.Scene.cornell()
    for i in range(len(g["ids"])):
        _, ids, pos, nrm = oracle.ray_intersect(sc, g["origins"][i], g["dirs"][i], g["origin_ids"][i])
        assert np.array_equal(ids, g["ids"][i])
        assert np.array_equal(pos.view(np.uint32), g["pos"][i].view(np.uint32))
        assert np.array_equal(nrm.view(np.uint32), g["normals"][i].view(np.uint32))


# ---- samplers: kernel0.cl:155-217, :547-569 -----------------------------------

def test_cone_sampler_stays_in_the_hemisphere(oracle):
    rng = np.random.default_rng(5)
    for _ in range(500):
        n = rng.standard_normal(3)
        n /= np.linalg.norm(n)
        d, w = oracle.sample_cone(int(rng.integers(1, 2**32) | 1), n, np.float32(3.1415 * 0.5))
        assert abs(np.linalg.norm(d) - 1) < 1e-5
        assert np.dot(d, n.astype(np.float32)) > -1e-4


def test_light_sampler_lies_on_the_sphere(bdpt, oracle):
    s = bdpt.Scene.cornell().spheres[0:1]
    rng = np.random.default_rng(6)
    ys = []
    for _ in range(2000):
        pos, nrm, _ = oracle.sample_sphere(int(rng.integers(1, 2**32) | 1), s)
        assert abs(np.linalg.norm(pos - s["center"][0, :3]) - 0.2) < 1e-6
        assert abs(np.linalg.norm(nrm) - 1) < 1e-5
        ys.append(nrm[1])
    assert abs(np.mean(ys)) < 0.05      # uniform over the sphere: mean height 0


def _lcg_floats(seed, n):
    out, w = [], seed
    for _ in range(n):
        w = (w * 48271) & 0xFFFFFFFF
        out.append(w / 4294967296.0)
    return out, w


def test_samplers_follow_the_reference_formulas_in_float64(bdpt, oracle):
    """The two samplers against their formulas (kernel0.cl:155-185, :547-569) evaluated in float64 from the same
    draws: cone = cos(a) n + sin(a) (cos(b) t1 + sin(b) t2), a = (1 - acos(u0) 2/pi) half_angle, b = 2 pi u1,
    t1 = normalize(n x (2 u2 - 1, 0, 1)), t2 = n x t1; light point = c + r (sin th sin ph, cos th, sin th cos ph),
    th = acos(1 - 2 u0), ph = 2 pi u1."""
    rng = np.random.default_rng(8)
    light = bdpt.Scene.cornell().spheres[0:1]
    c, r = light["center"][0, :3].astype(np.float64), float(light["radius"][0])
    for _ in range(400):
        seed = int(rng.integers(1, 2**32)) | 1
        n = rng.standard_normal(3)
        n = (n / np.linalg.norm(n)).astype(np.float32)
        half = np.float32(3.1415 * 0.5)
        got, w_after = oracle.sample_cone(seed, n, half)
        (u0, u1, u2), w = _lcg_floats(seed, 3)
        assert w_after == w                                      # three draws, in this order
        n64 = n.astype(np.float64)
        t1 = np.cross(n64, (2 * u2 - 1, 0.0, 1.0))
        t1 /= np.linalg.norm(t1)
        t2 = np.cross(n64, t1)
        a, b = (1 - math.acos(u0) * 2 / math.pi) * float(half), 6.283185307 * u1
        want = math.cos(a) * n64 + math.sin(a) * (math.cos(b) * t1 + math.sin(b) * t2)
        assert np.allclose(got, want, atol=3e-6), (seed, got, want)

        pos, nrm, w_after = oracle.sample_sphere(seed, light)
        (v0, v1), w = _lcg_floats(seed, 2)
        assert w_after == w
        th, ph = math.acos(1 - 2 * v0), 2 * math.pi * v1
        dirn = np.array([math.sin(th) * math.sin(ph), math.cos(th), math.sin(th) * math.cos(ph)])
        assert np.allclose(nrm, dirn, atol=3e-6) and np.allclose(pos, c + r * dirn, atol=3e-6)


def test_camera_ray_through_the_screen_centre_is_the_look_direction(bdpt, oracle):
    cam = bdpt.Scene.cornell().camera
    d = oracle.shoot_ray(cam, 0.0, 0.0)
    assert np.allclose(d, cam["look_at"][:3], atol=1e-6)
    up = oracle.shoot_ray(cam, 0.0, 1.0)     # vertical half field of view = atan(1/2)
    assert abs(np.dot(up, cam["look_at"][:3]) - math.cos(math.atan(0.5))) < 1e-5


# ---- the sample loop: kernel0.cl:571-864 --------------------------------------

def _render(bdpt, oracle, scene, W, H, spp, seed=1, per_sample=False, **pkw):
    seeds = bdpt.make_seeds(W, H, seed)
    st = oracle.State(W, H, seeds)
    p = bdpt.default_params(**pkw)
    rads = []
    for _ in range(spp if per_sample else 1):
        oracle.render(st, scene, scene.camera, p, 1 if per_sample else spp, n_threads=4)
        rads.append(st.radiance.copy())
    return st, seeds, rads


def test_early_out_pixels_freeze_their_seed_and_overwrite_history(bdpt, oracle):
    """kernel0.cl:645-656: a sample whose eye ray misses or hits the light writes the raw value and returns before the seed write-back."""
    sc = bdpt.Scene.cornell()
    W, H = 64, 36
    seeds = bdpt.make_seeds(W, H, 1)
    st = oracle.State(W, H, seeds)
    p = bdpt.default_params()
    oracle.render(st, sc, sc.camera, p, 1, n_threads=4)
    # classify by the FIRST sample: once a sample returns early the pixel repeats it forever
    miss = st.eye_hit[..., 0] == -1
    light = (st.eye_hit[..., 0] == 0) & (st.eye_hit[..., 1] == 0)
    assert miss.any() and light.any()
    hash1 = st.hit_hash.copy()
    oracle.render(st, sc, sc.camera, p, 5, n_threads=4)
    early = miss | light
    assert np.array_equal(st.eye_hit[..., 0][miss], np.full(miss.sum(), -1))
    w0 = seeds.reshape(W, H, 4)[:, :, 3].T
    w1 = st.seeds.reshape(W, H, 4)[:, :, 3].T
    assert np.array_equal(w0[early], w1[early])
    assert (w0[~early] != w1[~early]).all()
    assert not st.accum[miss].any()
    assert (st.accum[light] == 1.0).all()                   # the light's surface colour, not its emission
    assert (st.rng_draws[early] == 3 * 6).all()             # 1 discarded + 2 jitter draws per sample
    assert (st.rays[early] == 6).all()
    assert (st.hit_hash[early] != hash1[early]).all()       # the chain keeps absorbing the repeated id
    # x, y, z of every seed are never touched
    assert np.array_equal(seeds.reshape(-1, 4)[:, :3], st.seeds.reshape(-1, 4)[:, :3])


def test_draw_and_ray_counts_follow_the_loop_structure(bdpt, oracle):
    sc = bdpt.Scene.cornell()
    st, _, _ = _render(bdpt, oracle, sc, 48, 48, 1)
    full = st.eye_hit[..., 0] >= 0
    full &= ~((st.eye_hit[..., 0] == 0) & (st.eye_hit[..., 1] == 0))
    d, r = st.rng_draws[full].astype(int), st.rays[full].astype(int)
    it = (d - 5) // 4                                      # 5 fixed draws + 4 per light-loop iteration
    assert ((d - 5) % 4 == 0).all() and (it >= 1).all() and (it <= 5).all()
    # one eye ray, one connection ray per iteration, one bounce ray per iteration except possibly the last
    assert ((r == 1 + 2 * it) | (r == 2 * it)).all()


def test_alternative_samplers_change_the_draw_pattern_not_the_structure(bdpt, oracle):
    """HEMISPHERE draws like the live sampler (4 per iteration); REFLECTIVE draws one more number per vertex
    (kernel0.cl:493).  At the light the incoming direction is the zero vector (:659), so a mirror pick there (p4 > 0.6)
    yields a NaN direction that hits nothing: the walk ends after that bounce ray."""
    sc = bdpt.Scene.cornell()
    base, _, _ = _render(bdpt, oracle, sc, 48, 48, 1)
    full = base.eye_hit[..., 0] >= 0
    full &= ~((base.eye_hit[..., 0] == 0) & (base.eye_hit[..., 1] == 0))
    hemi, _, _ = _render(bdpt, oracle, sc, 48, 48, 1, sampler=bdpt.SAMPLER_HEMISPHERE)
    d = hemi.rng_draws[full].astype(int)
    assert ((d - 5) % 4 == 0).all() and np.array_equal(hemi.eye_hit, base.eye_hit)
    assert not np.array_equal(hemi.radiance, base.radiance) and np.isfinite(hemi.radiance).all()
    refl, _, _ = _render(bdpt, oracle, sc, 48, 48, 1, sampler=bdpt.SAMPLER_REFLECTIVE)
    d, r = refl.rng_draws[full].astype(int), refl.rays[full].astype(int)
    it = (d - 5) // 5
    assert ((d - 5) % 5 == 0).all() and (it >= 1).all() and (it <= 5).all()
    assert ((r == 1 + 2 * it) | (r == 2 * it)).all() and np.isfinite(refl.radiance).all()
    # more walks stop after one vertex than Russian roulette alone would stop (20 %)
    assert (it == 1).mean() > 0.4
    # a STRIDED + SUM launch takes the same samples in either order (the class follows draws 10, 15, ...)
    kw = dict(rng_mode=bdpt.RNG_STRIDED, accum_mode=bdpt.ACCUM_SUM, sampler=bdpt.SAMPLER_REFLECTIVE)
    a, _, _ = _render(bdpt, oracle, sc, 32, 24, 24, sample_order=bdpt.ORDER_INDEX, **kw)
    b, _, _ = _render(bdpt, oracle, sc, 32, 24, 24, sample_order=bdpt.ORDER_ROULETTE_CLASS, **kw)
    assert np.array_equal(a.rng_draws, b.rng_draws) and np.array_equal(a.rays, b.rays)
    np.testing.assert_allclose(a.accum, b.accum, rtol=2e-5, atol=1e-5)


def test_running_mean_is_the_mean_of_the_samples(bdpt, oracle):
    sc = bdpt.Scene.cornell()
    st, _, rads = _render(bdpt, oracle, sc, 40, 24, 12, per_sample=True)
    full = st.eye_hit[..., 0] >= 0
    full &= ~((st.eye_hit[..., 0] == 0) & (st.eye_hit[..., 1] == 0))
    mean = np.mean(np.stack(rads).astype(np.float64), axis=0)
    assert np.allclose(st.accum[full], mean[full], rtol=2e-5, atol=1e-6)


def test_batched_and_threaded_runs_are_bit_identical(bdpt, oracle):
    sc = bdpt.Scene.cornell()
    a, _, _ = _render(bdpt, oracle, sc, 40, 24, 5)
    b, _, _ = _render(bdpt, oracle, sc, 40, 24, 5, per_sample=True)
    assert np.array_equal(a.accum.view(np.uint32), b.accum.view(np.uint32))
    assert np.array_equal(a.seeds, b.seeds) and np.array_equal(a.hit_hash, b.hit_hash)
    seeds = bdpt.make_seeds(40, 24, 1)
    c = oracle.State(40, 24, seeds)
    oracle.render(c, sc, sc.camera, bdpt.default_params(), 5, n_threads=1)
    assert np.array_equal(a.accum.view(np.uint32), c.accum.view(np.uint32))


def test_strided_sample_ranges_add_up(bdpt, oracle):
    sc = bdpt.Scene.cornell()
    W, H = 32, 20
    kw = dict(rng_mode=bdpt.RNG_STRIDED, accum_mode=bdpt.ACCUM_SUM)
    seeds = bdpt.make_seeds(W, H, 8)
    p = bdpt.default_params(**kw)
    whole = oracle.State(W, H, seeds)
    oracle.render(whole, sc, sc.camera, p, 9, n_threads=4)
    parts = []
    for first, cnt in ((0, 4), (4, 5)):
        s = oracle.State(W, H, seeds)
        oracle.render(s, sc, sc.camera, p, cnt, sample_offset=first, n_threads=4)
        parts.append(s.accum.astype(np.float64))
        assert np.array_equal(s.seeds, seeds)              # strided mode never writes the seeds back
    assert np.allclose(parts[0] + parts[1], whole.accum, rtol=1e-5, atol=1e-6)
    # the stride is 48271^32: sample 1 of a pixel starts where 32 draws of sample 0 would end
    f, w = oracle.rng_sequence(int(seeds[3]), 32)
    assert int(w[-1]) == (int(seeds[3]) * pow(48271, 32, 2**32)) % 2**32


def test_image_is_plausible(bdpt, oracle):
    sc = bdpt.Scene.cornell()
    st, _, _ = _render(bdpt, oracle, sc, 64, 36, 32, blur_factor=0.0)   # no jitter: every sample of a pixel sees the same primitive
    img = st.accum
    assert np.isfinite(img).all() and img.min() >= 0
    hit = st.eye_hit[..., 0] >= 0
    assert 0.02 < img[hit][:, :3].mean() < 2.0
    red = (st.eye_hit[..., 0] == 2) & np.isin(st.eye_hit[..., 1], (8, 9))
    green = (st.eye_hit[..., 0] == 2) & np.isin(st.eye_hit[..., 1], (10, 11))
    assert red.any() and green.any()
    assert img[red][:, 1].max() == 0 and img[red][:, 2].max() == 0      # red wall reflects only red
    assert img[green][:, 0].max() == 0 and img[green][:, 2].max() == 0


@pytest.mark.parametrize("case", gu.RENDER_CASES)
def test_render_golden(bdpt, oracle, case):
    g = gu.load(case)
    scene, p, W, H, spp, off = gu.unpack(bdpt, g)
    st = oracle.State(W, H, g["seeds"])
    for s in range(spp):
        oracle.render(st, scene, scene.camera, p, 1, sample_offset=off, n_threads=4)
        assert np.array_equal(st.eye_hit, g["out_eye_hit"][s])
        assert np.array_equal(st.hit_hash, g["out_hit_hash"][s])
        assert np.array_equal(st.rng_draws, g["out_rng_draws"][s]) and np.array_equal(st.rays, g["out_rays"][s])
        assert np.array_equal(st.radiance.view(np.uint32), g["out_radiance"][s].view(np.uint32))
        assert np.array_equal(st.accum.view(np.uint32), g["out_accum"][s].view(np.uint32))
    assert np.array_equal(st.seeds, g["seeds_after"])


@pytest.mark.parametrize("case", gu.REGION_CASES)
def test_region_golden(bdpt, oracle, case):
    """Regions of the 1920x1080 frame (Cornell 64x64 x 16 spp, 100 000-triangle soup 32x32 x 2 spp)."""
    g = gu.load(case)
    scene, p, W, H, spp, region, seed = gu.unpack_region(bdpt, g)
    x0, y0, w, h = region
    st = oracle.State(W, H, bdpt.make_seeds(W, H, seed))
    for s in range(spp):
        oracle.render(st, scene, scene.camera, p, 1, region=region, n_threads=8)
    s = spp - 1
    crop = lambda a: a[y0:y0 + h, x0:x0 + w]
    assert np.array_equal(crop(st.eye_hit), g["out_eye_hit"][s]) and np.array_equal(crop(st.hit_hash), g["out_hit_hash"][s])
    assert np.array_equal(crop(st.rng_draws), g["out_rng_draws"][s]) and np.array_equal(crop(st.rays), g["out_rays"][s])
    assert np.array_equal(crop(st.accum).view(np.uint32), g["out_accum"][s].view(np.uint32))


def test_path_trace_sky_and_light_known_answers(bdpt, oracle):
    """path_trace (kernel0.cl:872-1001): an eye ray that leaves the scene returns the sky colour of :979, one that
    sees the light its emission / (2 pi) (:971 with wt = 1), and every sample draws 2 + 4 numbers per loop pass."""
    scene = bdpt.Scene.outdoor()
    W, H = 64, 36
    p = bdpt.default_params(aov_enable=1, integrator=bdpt.INTEGRATOR_PATH_TRACE)
    st = oracle.State(W, H, bdpt.make_seeds(W, H, 3))
    oracle.render(st, scene, scene.camera, p, 1, n_threads=4)
    miss = st.eye_hit[..., 0] < 0
    assert miss.any() and (~miss).any()
    assert np.array_equal(st.radiance[miss], np.tile(np.float32([0.3, 0.3, 0.4, 0.3]), (miss.sum(), 1)))
    assert (st.rays[miss] == 1).all() and (st.rng_draws[miss] == 1 + 2 + 4).all()
    light = (st.eye_hit[..., 0] == 0) & (st.eye_hit[..., 1] == 0)
    if light.any():
        e = scene.surfaces["emissive"][scene.spheres["surface"][0]]
        np.testing.assert_allclose(st.radiance[light], np.tile(e / np.float32(2 * np.pi), (light.sum(), 1)), rtol=1e-6)
    # four rays: either the fourth ended the path (2 + 4*4 draws) or the fifth vertex was reached, whose sampler
    # still draws before the loop ends with the factor 0.3 of :957 (2 + 5*4 draws)
    deep = st.rays == 4
    assert deep.any() and set(np.unique(st.rng_draws[deep])) <= {1 + 2 + 16, 1 + 2 + 20}
    # a second launch discards one more draw (:892) and blends with the running mean (:990)
    before = st.accum.copy()
    oracle.render(st, scene, scene.camera, p, 1, n_threads=4)
    both = miss & (st.eye_hit[..., 0] < 0)          # sky in both samples (a silhouette pixel may hit the second time)
    np.testing.assert_allclose(st.accum[both], before[both], rtol=1e-6)
    assert (st.rng_draws[both] == 2 * (1 + 2 + 4)).all()

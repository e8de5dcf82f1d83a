"""Host-side renderer surface (C++): scenes, seeds, orbit camera and its change flag."""
import numpy as np

import dist_render as dr


def test_cornell_scene_is_the_reference_scene(bdpt):
    sc = bdpt.Scene.cornell()        # src/renderer/main.cpp:270-422
    assert (len(sc.spheres), len(sc.planes), len(sc.triangles), len(sc.surfaces)) == (3, 0, 12, 5)
    assert np.allclose(sc.spheres["center"][0], (0, 1.5, 0, 0)) and np.isclose(sc.spheres["radius"][0], 0.2)
    assert sc.spheres["surface"].tolist() == [2, 0, 0]
    assert np.allclose(sc.surfaces["emissive"][2], 5.0) and not sc.surfaces["emissive"][[0, 1, 3, 4]].any()
    assert sc.triangles["surface"].tolist() == [1] * 8 + [3, 3, 4, 4]
    # every wall faces the inside of the box: normalize(cross(p1-p0, p2-p0)) points at the origin
    a, b, c = (sc.triangles[k][:, :3] for k in ("p0", "p1", "p2"))
    n = np.cross(b - a, c - a)
    centroid = (a + b + c) / 3
    assert (np.einsum("ij,ij->i", n, -centroid) > 0).all()
    assert np.abs(np.stack([a, b, c])).max() == 3.0
    # pairs of triangles tile each wall: 6 walls x 36 units of area
    assert np.isclose(0.5 * np.linalg.norm(n, axis=1).sum(), 6 * 36.0)


def test_other_scenes(bdpt):
    pl = bdpt.Scene.cornell_planes()     # main.cpp:306-328
    assert (len(pl.spheres), len(pl.planes), len(pl.triangles)) == (3, 6, 0)
    assert (np.einsum("ij,ij->i", pl.planes["normal"][:, :3], -pl.planes["point"][:, :3]) > 0).all()
    od = bdpt.Scene.outdoor()
    assert (len(od.spheres), len(od.planes), len(od.triangles)) == (6, 0, 0) and od.spheres["surface"][0] == 2
    s1, s2 = bdpt.Scene.soup(1000), bdpt.Scene.soup(1000)
    assert len(s1.triangles) == 1012 and np.array_equal(s1.triangles, s2.triangles)
    assert np.array_equal(s1.triangles[:12], bdpt.Scene.cornell().triangles)
    soup = s1.triangles[12:]
    assert np.abs(soup["p0"][:, :3]).max() <= 2.5 and np.abs(soup["p1"][:, :3] - soup["p0"][:, :3]).max() <= 0.05 + 1e-6
    assert set(soup["surface"].tolist()) == {0, 1, 3, 4}
    assert not np.array_equal(bdpt.Scene.soup(1000, seed=1).triangles, s1.triangles)


def test_default_camera_follows_model_camera_reset(bdpt):
    cam = bdpt.Scene.cornell().camera     # camera.cpp:44-53 with main.cpp:517-519
    d = np.array([1.0, 0.1, -0.1])
    pos = d / np.linalg.norm(d) * 8.0
    look = -pos / np.linalg.norm(pos)
    up = np.cross(look, np.cross(look, (0, -1, 0)))
    up /= np.linalg.norm(up)
    assert np.allclose(cam["pos"][:3], pos, atol=1e-5) and np.allclose(cam["look_at"][:3], look, atol=1e-6)
    assert np.allclose(cam["up"][:3], up, atol=1e-6)
    assert cam["pos"][3] == 0 and cam["look_at"][3] == 0 and cam["up"][3] == 0
    assert abs(np.dot(cam["up"][:3], cam["look_at"][:3])) < 1e-6


def test_orbit_keeps_radius_and_reports_change_while_held(bdpt):
    cams, changed = bdpt.camera_orbit(8.0, (0, -1, 0), (1.0, 0.1, -0.1), 0.03, 0.03, 0.5 / 0.03, 0.0, 60)
    assert changed.all()                                  # isChanged() == button held (camera.cpp:62-65)
    pos = cams["pos"][:, :3].astype(np.float64)
    assert np.allclose(np.linalg.norm(pos, axis=1), 8.0, atol=1e-4)
    assert np.allclose(cams["look_at"][:, :3], -pos / 8.0, atol=1e-5)
    # 0.5 degrees per frame about the up vector
    cosang = np.einsum("ij,ij->i", pos[1:], pos[:-1]) / 64.0
    assert np.allclose(np.degrees(np.arccos(np.clip(cosang, -1, 1))), 0.5, atol=0.02)
    assert np.array_equal(cams[0], bdpt.camera_reset(8.0, (0, -1, 0), (1.0, 0.1, -0.1)))


def test_seeds_are_deterministic_and_odd(bdpt):
    a, b = bdpt.make_seeds(37, 21, 0x5EED), bdpt.make_seeds(37, 21, 0x5EED)
    assert np.array_equal(a, b) and a.size == 37 * 21 * 4
    assert (a.reshape(-1, 4)[:, 3] & 1).all()
    assert not np.array_equal(a, bdpt.make_seeds(37, 21, 0x5EEE))
    assert len(np.unique(a.reshape(-1, 4)[:, 3])) > 0.99 * 37 * 21


def test_pfm_writer(bdpt, tmp_path):
    img = np.arange(4 * 3 * 4, dtype=np.float32).reshape(3, 4, 4)
    path = str(tmp_path / "x.pfm")
    bdpt.write_pfm(path, img)
    raw = open(path, "rb").read()
    head, body = raw[:raw.index(b"-1.0\n") + 5], raw[raw.index(b"-1.0\n") + 5:]
    assert head == b"PF\n4 3\n-1.0\n"
    data = np.frombuffer(body, "<f4").reshape(3, 4, 3)
    assert np.array_equal(data[::-1], img[..., :3])


def test_sample_and_row_partitions_cover_everything_once():
    for total, world in ((1024, 8), (10, 4), (3, 8), (0, 2), (1, 1)):
        got = [dr.sample_range(total, world, r) for r in range(world)]
        assert sum(c for _, c in got) == total
        pos = 0
        for first, cnt in got:
            assert first == pos
            pos += cnt
    for height, world in ((1080, 8), (600, 7), (36, 8), (8, 3), (5, 2)):
        bands = [dr.row_band(height, world, r) for r in range(world)]
        pos = 0
        for y0, h in bands:
            if h:
                assert y0 == pos and (y0 % 8 == 0)
                pos += h
        assert pos == height

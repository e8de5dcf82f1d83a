"""Host-side renderer surface (C++): scenes, seeds, orbit camera and its change flag."""
import numpy as np
import pytest

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


def test_middle_button_drag_changes_the_radius_only(bdpt):
    cams = bdpt.camera_zoom(8.0, (0, -1, 0), (1.0, 0.1, -0.1), 0.05, 10.0, 6)      # camera.cpp:77-89
    pos = cams["pos"][:, :3].astype(np.float64)
    assert np.allclose(np.linalg.norm(pos, axis=1), 8.0 - 0.5 * np.arange(7), atol=1e-4)
    unit = pos / np.linalg.norm(pos, axis=1, keepdims=True)
    assert np.allclose(unit, unit[0], atol=1e-6) and np.allclose(cams["look_at"][:, :3], -unit, atol=1e-6)
    assert np.allclose(cams["up"][:, :3], cams["up"][0, :3], atol=1e-6)


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


# ---- scene files, OBJ meshes, image output --------------------------------------------------------------------------

_CUBE_OBJ = """# unit cube, quads, mixed corner forms
o cube
v -1 -1 -1
v  1 -1 -1
v  1  1 -1
v -1  1 -1
v -1 -1  1
v  1 -1  1
v  1  1  1
v -1  1  1
vt 0 0
vn 0 0 1
s off
f 1 2 3 4
f 5/1 8/1 7/1 6/1
f 1//1 5//1 6//1 2//1
f 2/1/1 6/1/1 7/1/1 3/1/1
f -6 -2 -1 -5
f 4 8 5
f 4 5 1
"""


def test_obj_faces_become_triangles_in_file_order(bdpt, tmp_path):
    obj = tmp_path / "cube.obj"
    obj.write_text(_CUBE_OBJ)
    base = tmp_path / "base.scn"
    bdpt.Scene.cornell().save(str(base))
    sc = bdpt.Scene.load(str(base), objs=[(str(obj), 3, 0.5, (1.0, 2.0, 3.0))])
    tri = sc.triangles[12:]
    assert len(tri) == 5 * 2 + 2 and (tri["surface"] == 3).all()      # five quads fan-triangulated + two triangles
    v = np.array([[-1, -1, -1], [1, -1, -1], [1, 1, -1], [-1, 1, -1], [-1, -1, 1], [1, -1, 1], [1, 1, 1], [-1, 1, 1]], np.float32)
    place = lambda i: v[i - 1] * np.float32(0.5) + np.array([1, 2, 3], np.float32)
    # quad 1 2 3 4 -> (1,2,3), (1,3,4): the winding of the file is kept (loader.cpp:66-71)
    for t, (i, j, k) in zip(tri[:2], [(1, 2, 3), (1, 3, 4)]):
        assert np.array_equal(t["p0"][:3], place(i)) and np.array_equal(t["p1"][:3], place(j)) and np.array_equal(t["p2"][:3], place(k))
    # relative indices: -6 -2 -1 -5 with 8 vertices read = 3 7 8 4
    assert np.array_equal(tri[8]["p0"][:3], place(3)) and np.array_equal(tri[8]["p1"][:3], place(7))
    assert np.array_equal(tri[9]["p2"][:3], place(4))
    assert np.array_equal(tri[11]["p0"][:3], place(4)) and np.array_equal(tri[11]["p2"][:3], place(1))
    assert np.array_equal(sc.triangles[:12], bdpt.Scene.cornell().triangles)


@pytest.mark.parametrize("text,what", [
    ("v 0 0 0\nv 1 0 0\nv 0 1 0\nf 1 2 4\n", "bad vertex reference"),
    ("v 0 0 0\nv 1 0 0\nv 0 1 0\nf 1 2 0\n", "bad vertex reference"),
    ("v 0 0\n", "three numbers"),
    ("v 0 0 0\nv 1 0 0\nf 1 2\n", "three corners"),
])
def test_obj_errors_name_the_line(bdpt, tmp_path, text, what):
    obj = tmp_path / "bad.obj"
    obj.write_text(text)
    base = tmp_path / "base.scn"
    bdpt.Scene.cornell().save(str(base), binary=True)
    with pytest.raises(ValueError, match=what) as e:
        bdpt.Scene.load(str(base), objs=[(str(obj), 0, 1.0, (0, 0, 0))])
    assert "bad.obj:%d" % text.count("\n") in str(e.value)
    with pytest.raises(ValueError, match="cannot open"):
        bdpt.Scene.load(str(base), objs=[(str(tmp_path / "missing.obj"), 0, 1.0, (0, 0, 0))])
    with pytest.raises(ValueError, match="surface does not exist"):
        bdpt.Scene.load(str(base), objs=[(str(obj), 99, 1.0, (0, 0, 0))])


@pytest.mark.parametrize("binary", [False, True])
@pytest.mark.parametrize("which", ["cornell", "cornell_planes", "outdoor", "soup"])
def test_scene_files_round_trip_every_bit(bdpt, tmp_path, which, binary):
    sc = bdpt.Scene.soup(500) if which == "soup" else getattr(bdpt.Scene, which)()
    path = tmp_path / "scene.scn"
    sc.save(str(path), binary=binary)
    back = bdpt.Scene.load(str(path))
    for name in ("spheres", "planes", "triangles", "surfaces"):
        assert np.array_equal(getattr(sc, name), getattr(back, name)), name
    assert back.view == sc.view and np.array_equal(back.camera, sc.camera)


def test_text_scene_grammar_and_errors(bdpt, tmp_path):
    (tmp_path / "tri.obj").write_text("v 0 0 0\nv 1 0 0\nv 0 1 0\nf 1 2 3\n")
    good = tmp_path / "good.scn"
    good.write_text("# a light, a floor and a mesh\n"
                    "sphere 0 2 0 0.25 0     # names surface 0 before it is declared\n"
                    "surface 1 1 1 1  9 9 9 9\n"
                    "surface 0.5 0.25 0.125 1  0 0 0 0\n"
                    "plane 0 1 0  0 -1 0  1\n"
                    "triangle 0 0 0  1 0 0  0 1 0  1\n"
                    "obj tri.obj 1 scale 2 translate 0 0 -1\n"
                    "view 0 0.5 1 6\n")
    sc = bdpt.Scene.load(str(good))
    assert (len(sc.spheres), len(sc.planes), len(sc.triangles), len(sc.surfaces)) == (1, 1, 2, 2)
    assert sc.spheres["radius"][0] == 0.25 and sc.surfaces["emissive"][0].tolist() == [9, 9, 9, 9]
    assert sc.triangles["p1"][1][:3].tolist() == [2, 0, -1] and sc.view == (0.0, 0.5, 1.0, 6.0)
    assert np.allclose(np.linalg.norm(sc.camera["pos"][:3]), 6.0)
    for text, what in [("sphere 0 0 0 1 0\n", "surface that does not exist"),
                       ("surface 1 1 1 1 0 0 0 0\n", "no sphere"),
                       ("surface 1 1 1 1 0 0 0 0\nsphere 0 0 0 -1 0\n", "scn:2: sphere needs"),
                       ("surface 1 1 1 1 0 0 0 0\nsphere 0 0 0 1 0\ncone 1 2 3\n", "scn:3: unknown record 'cone'"),
                       ("surface 1 1 1 1 0 0 0 0\nsphere 0 0 0 1 0\nobj nope.obj 0\n", "scn:3: .*nope.obj: cannot open"),
                       ("surface 1 1 1 x 0 0 0 0\n", "scn:1: surface needs 8 numbers")]:
        bad = tmp_path / "bad.scn"
        bad.write_text(text)
        with pytest.raises(ValueError, match=what):
            bdpt.Scene.load(str(bad))
    trunc = tmp_path / "trunc.scn"
    bdpt.Scene.cornell().save(str(trunc), binary=True)
    trunc.write_bytes(trunc.read_bytes()[:-7])
    with pytest.raises(ValueError, match="bad header or size"):
        bdpt.Scene.load(str(trunc))


def _read_png(path):
    import struct
    import zlib
    data = open(path, "rb").read()
    assert data[:8] == b"\x89PNG\r\n\x1a\n"
    pos, chunks = 8, []
    while pos < len(data):
        n, kind = struct.unpack(">I4s", data[pos:pos + 8])
        body = data[pos + 8:pos + 8 + n]
        assert struct.unpack(">I", data[pos + 8 + n:pos + 12 + n])[0] == zlib.crc32(kind + body)
        chunks.append((kind, body))
        pos += 12 + n
    assert [k for k, _ in chunks] == [b"IHDR", b"IDAT", b"IEND"]
    w, h, depth, colour, comp, filt, lace = struct.unpack(">IIBBBBB", chunks[0][1])
    assert (depth, colour, comp, filt, lace) == (8, 2, 0, 0, 0)
    raw = np.frombuffer(zlib.decompress(chunks[1][1]), np.uint8).reshape(h, 1 + 3 * w)
    assert not raw[:, 0].any()
    return raw[:, 1:].reshape(h, w, 3)


@pytest.mark.parametrize("shape", [(5, 7), (150, 200)])       # the second needs more than one 64 KB stored block
def test_png_ppm_pfm_hold_the_accumulator(bdpt, tmp_path, shape):
    h, w = shape
    rng = np.random.default_rng(5)
    img = rng.uniform(-0.2, 1.3, (h, w, 4)).astype(np.float32)
    img[0, 0, 0] = np.nan
    expect = np.rint(np.clip(np.nan_to_num(img[..., :3], nan=0.0), 0, 1) * 255).astype(np.uint8)
    bdpt.write_image(str(tmp_path / "a.png"), img)
    assert np.array_equal(_read_png(tmp_path / "a.png"), expect)
    bdpt.write_image(str(tmp_path / "a.ppm"), img)
    blob = (tmp_path / "a.ppm").read_bytes()
    head = ("P6\n%d %d\n255\n" % (w, h)).encode()
    assert blob.startswith(head) and np.array_equal(np.frombuffer(blob[len(head):], np.uint8).reshape(h, w, 3), expect)
    bdpt.write_image(str(tmp_path / "g.png"), img, scale=0.5, gamma=2.2)
    g = np.rint(np.clip(np.nan_to_num(img[..., :3], nan=0.0) * np.float32(0.5), 0, 1) ** (1 / 2.2) * 255)
    assert np.abs(_read_png(tmp_path / "g.png").astype(int) - g).max() <= 1
    bdpt.write_image(str(tmp_path / "a.pfm"), img)
    blob = (tmp_path / "a.pfm").read_bytes()
    head = ("PF\n%d %d\n-1.0\n" % (w, h)).encode()
    body = np.frombuffer(blob[len(head):], "<f4").reshape(h, w, 3)[::-1]
    assert blob.startswith(head) and np.array_equal(body, img[..., :3], equal_nan=True)
    with pytest.raises(OSError):
        bdpt.write_image(str(tmp_path / "a.gif"), img)

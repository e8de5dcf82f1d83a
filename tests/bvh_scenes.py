"""Scenes made to trip a hierarchy over the triangles (tests/test_gpu_bvh.py, tools/fuzz_bvh.py): the Cornell box plus
triangles from specks to sheets, slivers, far-away ones with large coordinates, axis-aligned ones (boxes of no thickness),
exact duplicates and coplanar overlaps (equal distances: the lowest index must win), clusters at one point, and the
whole scene scaled and moved away from the origin."""
import numpy as np

KINDS = ["plain", "cluster", "far", "axis_aligned", "duplicates", "moved"]


def mesh(bdpt, rng, cells=40, flat=False):
    """The Cornell box with a connected surface in it: a cells x cells grid of quads over the floor, two triangles each,
    heights random (or none: every triangle in one plane, every inner vertex shared by six).  Rays through shared edges
    and vertices get the same distance from two or more triangles: the lowest index must win."""
    sc = bdpt.Scene.cornell()
    xs = np.linspace(-2.5, 2.5, cells + 1, dtype=np.float32)
    h = np.zeros((cells + 1, cells + 1), np.float32) if flat else (rng.random((cells + 1, cells + 1)) * 0.6).astype(np.float32)
    tri = np.zeros(2 * cells * cells, bdpt.triangle_dtype)
    k = 0
    for i in range(cells):
        for j in range(cells):
            p = lambda a, b: (xs[a], np.float32(-1.5) + h[a, b], xs[b])
            for corners in ((p(i, j), p(i, j + 1), p(i + 1, j)), (p(i + 1, j), p(i, j + 1), p(i + 1, j + 1))):    # facing +y
                tri["p0"][k, :3], tri["p1"][k, :3], tri["p2"][k, :3] = corners
                k += 1
    tri["surface"] = rng.choice([0, 1, 3, 4], len(tri))
    order = rng.permutation(len(tri))                      # neighbours in space are not neighbours in index
    sc.triangles = np.concatenate([sc.triangles, tri[order]])
    return sc


def scene_for(bdpt, rng, kind=None, n=None):
    sc = bdpt.Scene.cornell()
    n = int(rng.choice([1, 3, 17, 120, 700, 3000])) if n is None else n
    tri = np.zeros(n, bdpt.triangle_dtype)
    size = 10.0 ** rng.uniform(-3.0, 0.5, n)
    centre = rng.uniform(-2.7, 2.7, (n, 3))
    kind = int(rng.integers(0, 6)) if kind is None else KINDS.index(kind)
    if kind == 1:
        centre[:] = centre[0] + rng.standard_normal((n, 3)) * 0.02          # a cluster
    if kind == 2:
        centre[::7] *= 50.0                                                    # some far away
    e1 = rng.standard_normal((n, 3)) * size[:, None]
    e2 = rng.standard_normal((n, 3)) * size[:, None]
    if kind == 3:                                                              # axis-aligned: one coordinate constant
        ax = rng.integers(0, 3, n)
        e1[np.arange(n), ax] = 0.0
        e2[np.arange(n), ax] = 0.0
        centre[np.arange(n), ax] = np.round(centre[np.arange(n), ax] * 4) / 4  # and many in the same planes
    sliver = rng.random(n) < 0.15
    thin = 10.0 ** rng.uniform(-7.0, -1.0, (int(sliver.sum()), 1))
    e2[sliver] = e1[sliver] * rng.uniform(0.2, 3.0, (int(sliver.sum()), 1)) + rng.standard_normal((int(sliver.sum()), 3)) * size[sliver, None] * thin
    tri["p0"][:, :3] = centre
    tri["p1"][:, :3] = centre + e1
    tri["p2"][:, :3] = centre + e2
    tri["surface"] = rng.choice([0, 1, 3, 4], n)
    if kind == 4 and n > 1:                                                    # duplicates and coplanar overlaps, shuffled
        dup = tri[: max(1, n // 3)].copy()
        shifted = dup.copy()
        shifted["p1"][:, :3] = dup["p0"][:, :3] + 0.7 * (dup["p1"][:, :3] - dup["p0"][:, :3])
        tri = np.concatenate([tri, dup, shifted])
        tri["surface"] = rng.choice([0, 1, 3, 4], len(tri))
        rng.shuffle(tri)
    sc.triangles = np.concatenate([sc.triangles, tri])
    if kind == 5:                                                              # the whole scene elsewhere and at another scale
        s, t = np.float32(10.0 ** rng.uniform(-1.5, 1.5)), rng.uniform(-30, 30, 3).astype(np.float32)
        for f in ("p0", "p1", "p2"):
            sc.triangles[f][:, :3] = sc.triangles[f][:, :3] * s + t
        sc.spheres["center"][:, :3] = sc.spheres["center"][:, :3] * s + t
        sc.spheres["radius"] *= s
        cam = sc.camera.copy()
        cam["pos"][:3] = cam["pos"][:3] * s + t
        cam["look_at"][:3] = cam["look_at"][:3] * s + t
        sc.camera = cam
    return sc

"""The hierarchy walk of the many-triangle kernel (BDPT_KERNEL_BVH, csrc/bdpt_bvh.h): a ray visits the triangles near
its line in whatever order the boxes come, and must still return the reference's hit -- the nearest, and among equally
near ones the lowest index (kernel0.cl:434-449 walks in index order with a strict "<").  Scenes made to trip it
(tests/bvh_scenes.py) against the CPU oracle, every id, count and bit; and the statements the builder makes about its
own output."""
import numpy as np
import pytest

import bvh_scenes
from test_gpu_parity import _compare, _pair

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("kind", bvh_scenes.KINDS)
@pytest.mark.parametrize("n", [3, 400])
@pytest.mark.parametrize("share", ["0", "1000000"])
def test_hierarchy_returns_the_reference_hit(bdpt, oracle, monkeypatch, kind, n, share):
    """share: hierarchies of this many triangles or more are walked with the lanes of a wavefront sharing work
    (BDPT_BVH_SHARE_MIN, read when a context is created): always, never."""
    monkeypatch.setenv("BDPT_BVH_SHARE_MIN", share)
    rng = np.random.default_rng(bvh_scenes.KINDS.index(kind) * 100 + n)
    scene = bvh_scenes.scene_for(bdpt, rng, kind, n)
    W, H = 48, 36
    ctx, st, p = _pair(bdpt, oracle, scene, W, H, seed=7 + n, kernel_variant=bdpt.KERNEL_BVH)
    for s in range(2):
        ctx.render(3)
        assert ctx.last_launch_variant() == bdpt.KERNEL_BVH
        oracle.render(st, scene, scene.camera, p, 3, n_threads=8)
        _compare(bdpt, ctx, st, "%s %d launch %d:" % (kind, n, s))
    ctx.close()


def test_equal_distances_go_to_the_lowest_index(bdpt, oracle):
    """Forty copies of one large triangle that faces the camera, in shuffled order among other triangles: every eye
    ray that hits the stack must report the copy with the lowest index."""
    rng = np.random.default_rng(5)
    scene = bvh_scenes.scene_for(bdpt, rng, "plain", 200)
    sheet = np.zeros(40, bdpt.triangle_dtype)
    corners = np.float32([(1.0, -1.2, 1.5), (1.0, -1.2, -1.5), (1.0, 1.6, 0.0)])     # the camera looks along -x
    sheet["p0"][:, :3], sheet["p1"][:, :3], sheet["p2"][:, :3] = corners
    sheet["surface"] = 1
    tri = np.concatenate([scene.triangles[12:], sheet])
    rng.shuffle(tri)
    scene.triangles = np.concatenate([scene.triangles[:12], tri])
    copies = [i for i in range(len(scene.triangles)) if all(np.array_equal(scene.triangles[f][i, :3], c) for f, c in zip(("p0", "p1", "p2"), corners))]
    assert len(copies) == 40
    W, H = 64, 48
    ctx, st, p = _pair(bdpt, oracle, scene, W, H, seed=11, kernel_variant=bdpt.KERNEL_BVH)
    ctx.render(2)
    oracle.render(st, scene, scene.camera, p, 2, n_threads=8)
    _compare(bdpt, ctx, st, "stack of copies:")
    eye = ctx.read_aov(bdpt.AOV_EYE_HIT)
    on_stack = np.isin(eye[..., 1], copies) & (eye[..., 0] == 2)
    assert on_stack.sum() > 50
    assert set(np.unique(eye[..., 1][on_stack])) == {min(copies)}
    ctx.close()


@pytest.mark.parametrize("limit", ["2", "2,64", "3,7"])
@pytest.mark.parametrize("share", ["0", "1000000"])
def test_a_walk_whose_stack_runs_over(bdpt, oracle, monkeypatch, limit, share):
    """A ray can pass all four boxes of node after node and need three stack entries per level; a lane whose LDS stack
    runs over walks again with a stack in its private memory, and if that runs over too it meets every triangle
    instead.  BDPT_BVH_STACK_LIMIT (read when a context is created) makes the stacks small enough for that to happen
    in nearly every search of this scene -- "2": both stacks two entries; "2,64": the second walk always succeeds;
    "3,7": some of each -- and every bit stays."""
    monkeypatch.setenv("BDPT_BVH_STACK_LIMIT", limit)
    monkeypatch.setenv("BDPT_BVH_SHARE_MIN", share)
    rng = np.random.default_rng(21)
    scene = bvh_scenes.scene_for(bdpt, rng, "duplicates", 300)
    W, H = 40, 30
    ctx, st, p = _pair(bdpt, oracle, scene, W, H, seed=3, kernel_variant=bdpt.KERNEL_BVH)
    assert ctx.hierarchy_info()["stack_entries"] == int(limit.split(",")[0])
    ctx.render(4)
    assert ctx.last_launch_variant() == bdpt.KERNEL_BVH
    oracle.render(st, scene, scene.camera, p, 4, n_threads=8)
    _compare(bdpt, ctx, st, "stack limit %s:" % limit)
    ctx.close()


@pytest.mark.parametrize("flat", [False, True])
def test_connected_mesh_with_shared_edges(bdpt, oracle, monkeypatch, flat):
    """3200 triangles that share every edge and vertex with neighbours of unrelated index (tests/bvh_scenes.py: mesh),
    seen from above by eye rays and from everywhere by connection and bounce rays: a ray through a shared edge or vertex
    gets equal distances from several triangles and must report the lowest index, as the reference's in-order walk does."""
    monkeypatch.setenv("BDPT_BVH_SHARE_MIN", "0")           # lanes share work, as in scenes of many more triangles
    rng = np.random.default_rng(9)
    scene = bvh_scenes.mesh(bdpt, rng, 40, flat)
    W, H = 64, 48
    for variant in (bdpt.KERNEL_BVH, bdpt.KERNEL_TILED):
        ctx, st, p = _pair(bdpt, oracle, scene, W, H, seed=13, kernel_variant=variant)
        for s in range(2):
            ctx.render(4)
            oracle.render(st, scene, scene.camera, p, 4, n_threads=8)
            _compare(bdpt, ctx, st, "mesh flat=%s variant %d launch %d:" % (flat, variant, s))
        assert (st.eye_hit[..., 1][st.eye_hit[..., 0] == 2] >= 12).any()         # the mesh is in view
        ctx.close()


def test_hierarchy_info_and_reserved_stack(bdpt):
    """What the builder says about its tree: a walk pushes at most three entries per level of four-wide nodes, the
    context reserves that many per lane (never more than 48), the oversized walls of the box stay outside, and a scene
    without triangles has an empty hierarchy."""
    with bdpt.Context(32, 32) as ctx:
        assert ctx.hierarchy_info() is None                                   # no scene yet
        ctx.upload_scene(bdpt.Scene.soup(5000))
        info = ctx.hierarchy_info()
        assert info["outside"] == 12 and 5000 // 4 < info["nodes"] < 5000
        assert 13 <= info["depth"] <= 39                                      # between ceil(log2(5000)) and cuts that leave an eighth on one side
        assert 7 <= info["stack_entries"] <= 3 * ((info["depth"] + 1) // 2) + 3
        ctx.upload_scene(bdpt.Scene.outdoor())
        assert ctx.hierarchy_info() == dict(nodes=0, depth=0, stack_entries=1, outside=0)

"""Loads a golden fixture (tests/golden/*.npz) back into scene / params / expected outputs."""
import glob
import os

import numpy as np

GOLDEN_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")

RENDER_CASES = sorted(os.path.basename(p)[:-4] for p in glob.glob(os.path.join(GOLDEN_DIR, "*spp*.npz")))


REGION_CASES = sorted(os.path.basename(p)[:-4] for p in glob.glob(os.path.join(GOLDEN_DIR, "region_*.npz")))
LAUNCH_CASES = sorted(os.path.basename(p)[:-4] for p in glob.glob(os.path.join(GOLDEN_DIR, "launch4k_*.npz")))


def load(name):
    return np.load(os.path.join(GOLDEN_DIR, name + ".npz"), allow_pickle=False)


def unpack(bdpt, g):
    scene = bdpt.Scene(g["spheres"], g["planes"], g["triangles"], g["surfaces"], g["camera"])
    p = bdpt.default_params()
    for f, _ in bdpt.Params._fields_:
        if "param_" + f in g.files:          # a parameter newer than the fixture keeps its default
            setattr(p, f, g["param_" + f].item())
    return scene, p, int(g["width"]), int(g["height"]), int(g["spp"]), int(g["sample_offset"])


def unpack_region(bdpt, g):
    """A region fixture names its scene by recipe (the C++ host builds it deterministically)."""
    scene = bdpt.Scene.build(int(g["scene_kind"]), int(g["n_soup"]), int(g["soup_seed"]))
    assert np.asarray(scene.camera).tobytes() == np.asarray(g["camera"]).tobytes(), "the host's scene builder changed"
    p = bdpt.default_params()
    for f, _ in bdpt.Params._fields_:
        if "param_" + f in g.files:
            setattr(p, f, g["param_" + f].item())
    return scene, p, int(g["width"]), int(g["height"]), int(g["spp"]), tuple(int(v) for v in g["region"]), int(g["global_seed"])

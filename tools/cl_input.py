"""Writes the input file of tools/cl_baseline (the reference kernel on a CPU OpenCL device): the reference's host
structs byte for byte, camera and seeds, behind a 32-byte header (see the comment at the top of cl_baseline.c)."""
import numpy as np

MAGIC = 0x42445054


def write(path, scene, camera, seeds, width, height):
    seeds = np.ascontiguousarray(seeds, np.uint32)
    assert seeds.size == width * height * 4
    head = np.array([MAGIC, width, height, len(scene.spheres), len(scene.planes), len(scene.triangles), len(scene.surfaces), 0], "<i4")
    with open(path, "wb") as f:
        f.write(head.tobytes())
        for a in (scene.spheres, scene.planes, scene.triangles, scene.surfaces):
            f.write(np.ascontiguousarray(a).tobytes())
        f.write(np.ascontiguousarray(camera).tobytes())
        f.write(seeds.tobytes())

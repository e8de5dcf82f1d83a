"""tools/cl_baseline: the harness that times the reference's own OpenCL kernel on a CPU OpenCL device.  The kernel file
is not part of this repository and no image this was developed on has a CPU OpenCL device, so what is checked is
(a) on any host: the program builds, parses its input and either reports "pocl baseline unavailable: <reason>" with
exit status 0 or, where a CPU device does exist, runs the stand-in kernel below correctly; (b) on the GPU box: every
OpenCL call of the harness on a real device (the vendor's GPU OpenCL runtime) with a stand-in kernel of the same
signature whose output the test predicts.  The run with the reference's kernel itself is untested."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EXE = os.path.join(ROOT, "tools", "cl_baseline")
PROBE = os.path.join(ROOT, "tests", "data", "cl_probe_kernel.cl")
sys.path.insert(0, os.path.join(ROOT, "tools"))

W, H, FRAMES = 96, 40, 5


def _input(bdpt, tmp_path):
    import cl_input

    scene = bdpt.Scene.cornell()
    seeds = bdpt.make_seeds(W, H, 77)
    path = str(tmp_path / "scene.bin")
    cl_input.write(path, scene, scene.camera, seeds, W, H)
    return path, scene, seeds


def _expected(scene, seeds):
    """What tests/data/cl_probe_kernel.cl leaves after FRAMES frames, in float32 as the kernel computes it."""
    xs, ys = np.meshgrid(np.arange(W), np.arange(H))
    w = seeds[(xs * H + ys) * 4 + 3].astype(np.uint64)
    from_scene = np.float32(scene.spheres["radius"][-1] * 4.0 + scene.triangles["p0"][-1, 0] + scene.surfaces["emissive"][2, 0] + len(scene.planes))
    mean = np.zeros((H, W, 4), np.float32)
    for n in range(FRAMES):
        w = (w * 48271) & 0xFFFFFFFF
        value = np.stack([(xs + 2 * ys).astype(np.float32), (w >> 24).astype(np.float32), np.full((H, W), from_scene, np.float32),
                          np.full((H, W), scene.camera["pos"][0], np.float32)], -1)
        mean = value if n == 0 else mean * (np.float32(n) / np.float32(n + 1)) + value / np.float32(n + 1)
    return mean


def _run_probe(bdpt, tmp_path, device_type):
    path, scene, seeds = _input(bdpt, tmp_path)
    out = str(tmp_path / "image.bin")
    r = subprocess.run([EXE, "--input", path, "--kernel", PROBE, "--frames", str(FRAMES), "--warmup", "1", "--out", out,
                        "--device-type", device_type], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    if "pocl baseline unavailable" in r.stdout:
        return r.stdout.strip()
    lines = [json.loads(ln) for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert [ln["build_options"] for ln in lines] == ["-cl-opt-disable", "default"]
    for ln in lines:
        assert (ln["width"], ln["height"], ln["frames"]) == (W, H, FRAMES) and ln["seconds"] > 0 and ln["compute_units"] > 0
        assert abs(ln["mpaths_per_s"] - W * H * FRAMES / ln["seconds"] / 1e6) < 1e-3 * ln["mpaths_per_s"]
    got = np.fromfile(out, np.float32).reshape(H, W, 4)
    np.testing.assert_allclose(got, _expected(scene, seeds), rtol=1e-5, atol=1e-5)
    return None


def test_harness_builds_and_reports_a_missing_cpu_device_or_runs_on_one(bdpt, tmp_path):
    if not os.path.exists(EXE):
        pytest.skip("tools/cl_baseline is not built (no OpenCL loader to link against?)")
    why = _run_probe(bdpt, tmp_path, "cpu")
    if why is not None:
        assert why.startswith("pocl baseline unavailable: ") and len(why) > 40
    # without the kernel file the answer is a reason, not a failure
    path, _, _ = _input(bdpt, tmp_path)
    r = subprocess.run([EXE, "--input", path, "--kernel", str(tmp_path / "no_such_kernel.cl")], capture_output=True, text=True, timeout=60)
    assert r.returncode == 0 and r.stdout.startswith("pocl baseline unavailable: ")
    # a file that is not an input of this tool, and bad arguments, are errors
    assert subprocess.run([EXE, "--input", PROBE], capture_output=True).returncode == 1
    assert subprocess.run([EXE], capture_output=True).returncode == 1
    assert subprocess.run([EXE, "--input", path, "--device-type", "tpu"], capture_output=True).returncode == 1


@pytest.mark.gpu
def test_harness_drives_a_stand_in_kernel_on_a_real_opencl_device(bdpt, tmp_path):
    if not os.path.exists(EXE):
        pytest.skip("tools/cl_baseline is not built")
    why = _run_probe(bdpt, tmp_path, "gpu")
    if why is not None:
        pytest.skip(why)

"""The C-ABI library loads and exports every entry point include/bdpt.h declares;
struct layouts match the reference's host structs.  No compute call is made."""
import ctypes as C
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared(header, prefix):
    text = open(os.path.join(ROOT, header)).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(%s\w+)\s*\(" % prefix, text)))


def test_every_declared_entry_point_is_exported(bdpt):
    names = _declared("include/bdpt.h", "bdpt_")
    assert len(names) >= 25
    for n in names:
        assert hasattr(bdpt.lib, n), "libbdpt_amd.so does not export %s" % n


def test_host_library_exports(bdpt):
    text = open(os.path.join(ROOT, "opencl-path-tracer_amd/host/bdpt_host.h")).read()
    names = sorted(set(re.findall(r"\b(bdpt_host_\w+)\s*\(", text)))
    assert len(names) >= 10
    for n in names:
        assert hasattr(bdpt.hostlib, n), "libbdpt_host.so does not export %s" % n


def test_struct_layouts_match_the_reference_host_structs(bdpt):
    # src/include/geometry/primitives.h:6-64, src/include/renderer/camera.h:37-41
    assert bdpt.sphere_dtype.itemsize == 32 and bdpt.sphere_dtype.fields["radius"][1] == 16
    assert bdpt.sphere_dtype.fields["surface"][1] == 20
    assert bdpt.triangle_dtype.itemsize == 64 and bdpt.triangle_dtype.fields["surface"][1] == 48
    assert bdpt.plane_dtype.itemsize == 48 and bdpt.plane_dtype.fields["surface"][1] == 32
    assert bdpt.surface_dtype.itemsize == 32 and bdpt.camera_dtype.itemsize == 48
    assert C.sizeof(bdpt.Params) == 68


def test_default_params_are_the_reference_literals(bdpt):
    p = bdpt.default_params()
    assert p.max_light_bounces == 5                       # kernel0.cl:683
    assert abs(p.p_continue - 0.8) < 1e-7                  # kernel0.cl:687
    assert p.blur_factor == 2.0                            # kernel0.cl:13
    assert p.indirect_only == 0 and p.light_index == 0
    assert p.rng_mode == bdpt.RNG_STREAM and p.accum_mode == bdpt.ACCUM_MEAN
    assert p.connection_falloff == bdpt.FALLOFF_INV_SQ     # kernel0.cl:719
    assert bdpt.version().startswith("bdpt-amd")


def test_argument_errors_need_no_gpu(bdpt):
    h = C.c_void_p()
    assert bdpt.lib.bdpt_create(0, 0, 16, C.byref(h)) == -1
    assert bdpt.lib.bdpt_create(0, 16, 16, None) == -1
    assert b"bdpt_create" in bdpt.lib.bdpt_last_error(None)
    assert bdpt.lib.bdpt_num_samples(None) == -1
    bdpt.lib.bdpt_destroy(None)


def test_without_a_gpu_create_fails_loudly(bdpt):
    """The product has no CPU path: on a machine without a GPU, creating a context is an error."""
    try:
        ctx = bdpt.Context(16, 16)
    except bdpt.BdptError as e:
        assert e.code == -4 and "no HIP device" in str(e)
    else:
        ctx.close()
        pytest.skip("a GPU is present")


def test_sliced_launch_hand_off_compiles_to_the_instructions_it_depends_on():
    """The run-to-run hand-off of sliced launches (csrc/bdpt_slice.h) is correct because of which store and load
    instructions it compiles to; tools/check_slice_isa.py compiles the probe for gfx950 (no GPU needed) and reads
    them.  The checker itself is checked on doctored assembly: a state store whose sc1 bit became nt, a plain store,
    a missing invalidate."""
    import importlib.util
    import shutil
    import subprocess
    import tempfile

    if not shutil.which("hipcc"):
        pytest.skip("hipcc is not on PATH")
    spec = importlib.util.spec_from_file_location("check_slice_isa", os.path.join(ROOT, "tools", "check_slice_isa.py"))
    chk = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(chk)
    with tempfile.TemporaryDirectory() as tmp:
        out = os.path.join(tmp, "probe.s")
        subprocess.run(["hipcc"] + chk.FLAGS + ["--cuda-device-only", "-S", "-o", out, chk.PROBE], check=True, capture_output=True)
        asm = open(out).read()
    assert chk.check(asm) == []
    assert chk.check(re.sub(r"(global_store_dwordx2 [^\n]*) sc1", r"\1 nt", asm, count=1))
    assert chk.check(re.sub(r"(global_store_dword [^\n]*) sc1", r"\1", asm, count=1))
    assert chk.check(asm.replace("buffer_inv sc1", "s_nop 0"))
    assert chk.check(asm.replace("s_waitcnt vmcnt(0)", "s_nop 0"))

"""ctypes binding of the CPU oracle (oracle/bdpt_oracle.c).

TEST INFRASTRUCTURE ONLY: imported by tests/, by __graft_entry__.smoke() and by
the cpu_baseline leg of bench.py -- never by the product package.
PARITY UNPINNED: the reference holds no fixtures to check this oracle against
(see the header of bdpt_oracle.c).
"""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_vp = C.c_void_p


def _cpu_has_fma():
    try:
        with open("/proc/cpuinfo") as f:
            for line in f:
                if line.startswith("flags"):
                    return " fma " in line + " "
    except OSError:
        pass
    return False


def _load():
    name = "libbdpt_oracle.so" if _cpu_has_fma() else "libbdpt_oracle_nofma.so"
    path = os.path.join(_HERE, name)
    if not os.path.exists(path):
        raise ImportError("%s is not built; run `make` at the repository root" % path)
    return C.CDLL(path)


class Job(C.Structure):
    _fields_ = [
        ("width", C.c_int), ("height", C.c_int),
        ("camera", _vp),
        ("spheres", _vp), ("n_spheres", C.c_int),
        ("planes", _vp), ("n_planes", C.c_int),
        ("triangles", _vp), ("n_triangles", C.c_int),
        ("surfaces", _vp), ("n_surfaces", C.c_int),
        ("params", _vp),
        ("x0", C.c_int), ("y0", C.c_int), ("w", C.c_int), ("h", C.c_int),
        ("num_samples", C.c_int), ("spp", C.c_int), ("sample_offset", C.c_int),
        ("seeds", _vp), ("accum", _vp),
        ("aov_eye_hit", _vp), ("aov_hit_hash", _vp), ("aov_radiance", _vp),
        ("aov_rng_draws", _vp), ("aov_rays", _vp),
        ("counters", C.c_uint64 * 3),
        ("n_threads", C.c_int),
    ]


lib = _load()
lib.orc_render.argtypes = [C.POINTER(Job)]
lib.orc_ray_intersect.argtypes = [_vp, C.c_int, _vp, C.c_int, _vp, C.c_int, _vp, _vp, _vp, C.c_int, C.c_int, _vp, _vp, _vp]
for _n in ("sphere", "plane", "triangle"):
    getattr(lib, "orc_ray_" + _n).argtypes = [_vp, _vp, _vp, _vp]
lib.orc_rng_float.argtypes = [C.POINTER(C.c_uint32)]
lib.orc_rng_float.restype = C.c_float
lib.orc_sample_cone.argtypes = [C.POINTER(C.c_uint32), _vp, C.c_float, _vp]
lib.orc_sample_cone.restype = None
lib.orc_sample_sphere.argtypes = [C.POINTER(C.c_uint32), _vp, _vp, _vp]
lib.orc_sample_sphere.restype = None
lib.orc_shoot_ray.argtypes = [_vp, C.c_float, C.c_float, _vp]
lib.orc_shoot_ray.restype = None
lib.orc_sincos.argtypes = [C.c_float, C.POINTER(C.c_float), C.POINTER(C.c_float)]
lib.orc_sincos.restype = None
lib.orc_acosf.argtypes = [C.c_float]
lib.orc_acosf.restype = C.c_float
lib.orc_acos_f64.argtypes = [C.c_double]
lib.orc_acos_f64.restype = C.c_double


def _ptr(a):
    return a.ctypes.data_as(_vp) if a is not None and a.size else None


class State:
    """Per-pixel state the oracle carries between calls (what the device keeps in HBM)."""

    def __init__(self, width, height, seeds):
        self.width, self.height = width, height
        self.seeds = np.array(seeds, dtype=np.uint32, copy=True)
        assert self.seeds.size == width * height * 4
        self.accum = np.zeros((height, width, 4), np.float32)
        self.num_samples = 0
        self.reset_aovs()

    def reset_aovs(self):
        h, w = self.height, self.width
        self.eye_hit = np.full((h, w, 2), -1, np.int32)
        self.hit_hash = np.zeros((h, w), np.uint32)
        self.radiance = np.zeros((h, w, 4), np.float32)
        self.rng_draws = np.zeros((h, w), np.uint32)
        self.rays = np.zeros((h, w), np.uint32)


def render(state, scene, camera, params, spp, region=None, sample_offset=0, n_threads=1):
    """Runs `spp` samples on `state` in place, advancing state.num_samples. Returns the work counters."""
    x0, y0, w, h = region if region is not None else (0, 0, state.width, state.height)
    cam = np.ascontiguousarray(camera)
    j = Job()
    j.width, j.height = state.width, state.height
    j.camera = cam.ctypes.data_as(_vp)
    j.spheres, j.n_spheres = _ptr(scene.spheres), len(scene.spheres)
    j.planes, j.n_planes = _ptr(scene.planes), len(scene.planes)
    j.triangles, j.n_triangles = _ptr(scene.triangles), len(scene.triangles)
    j.surfaces, j.n_surfaces = _ptr(scene.surfaces), len(scene.surfaces)
    j.params = C.cast(C.pointer(params), _vp)
    j.x0, j.y0, j.w, j.h = x0, y0, w, h
    j.num_samples, j.spp, j.sample_offset = state.num_samples, spp, sample_offset
    j.seeds, j.accum = _ptr(state.seeds), _ptr(state.accum)
    j.aov_eye_hit, j.aov_hit_hash, j.aov_radiance = _ptr(state.eye_hit), _ptr(state.hit_hash), _ptr(state.radiance)
    j.aov_rng_draws, j.aov_rays = _ptr(state.rng_draws), _ptr(state.rays)
    j.n_threads = n_threads
    rc = lib.orc_render(C.byref(j))
    if rc != 0:
        raise RuntimeError("orc_render failed (%d)" % rc)
    state.num_samples += spp
    return tuple(int(c) for c in j.counters)


def ray_intersect(scene, origin, direction, origin_id=(4, 0)):
    o = np.asarray(origin, np.float32)
    d = np.asarray(direction, np.float32)
    ids = np.zeros(2, np.int32)
    pos = np.zeros(3, np.float32)
    nrm = np.zeros(3, np.float32)
    hit = lib.orc_ray_intersect(_ptr(scene.spheres), len(scene.spheres), _ptr(scene.planes), len(scene.planes),
                                _ptr(scene.triangles), len(scene.triangles), _ptr(scene.surfaces), _ptr(o), _ptr(d),
                                int(origin_id[0]), int(origin_id[1]), _ptr(ids), _ptr(pos), _ptr(nrm))
    return hit, ids, pos, nrm


def ray_primitive(kind, prim, origin, direction):
    """kind in {'sphere','plane','triangle'}; prim a 1-element structured array."""
    o = np.asarray(origin, np.float32)
    d = np.asarray(direction, np.float32)
    pos = np.zeros(3, np.float32)
    p = np.ascontiguousarray(prim)
    hit = getattr(lib, "orc_ray_" + kind)(p.ctypes.data_as(_vp), _ptr(o), _ptr(d), _ptr(pos))
    return hit, pos


def rng_sequence(seed, n):
    w = C.c_uint32(seed)
    out = np.empty(n, np.float32)
    words = np.empty(n, np.uint32)
    for i in range(n):
        out[i] = lib.orc_rng_float(C.byref(w))
        words[i] = w.value
    return out, words


def sample_cone(seed, normal, half_angle):
    w = C.c_uint32(seed)
    n = np.asarray(normal, np.float32)
    out = np.zeros(3, np.float32)
    lib.orc_sample_cone(C.byref(w), _ptr(n), half_angle, _ptr(out))
    return out, w.value


def sample_sphere(seed, sphere):
    w = C.c_uint32(seed)
    s = np.ascontiguousarray(sphere)
    pos = np.zeros(3, np.float32)
    nrm = np.zeros(3, np.float32)
    lib.orc_sample_sphere(C.byref(w), s.ctypes.data_as(_vp), _ptr(pos), _ptr(nrm))
    return pos, nrm, w.value


def shoot_ray(camera, sx, sy):
    cam = np.ascontiguousarray(camera)
    out = np.zeros(3, np.float32)
    lib.orc_shoot_ray(cam.ctypes.data_as(_vp), sx, sy, _ptr(out))
    return out


def sincos(x):
    s, c = C.c_float(), C.c_float()
    lib.orc_sincos(float(x), C.byref(s), C.byref(c))
    return s.value, c.value


def acosf(x):
    return lib.orc_acosf(float(x))


def acos_f64(x):
    return lib.orc_acos_f64(float(x))

"""ctypes binding of the C-ABI in include/bdpt.h and of the host-side helpers.

The product is the two shared libraries beside this file (libbdpt_amd.so: HIP
kernels + C-ABI; libbdpt_host.so: the C++ renderer surface).  This module only
marshals numpy arrays across that boundary; it contains no rendering code and
no CPU fallback -- if the libraries are missing, importing it raises.

Array conventions: images are (height, width, C) float32/uint32/int32 arrays,
row-major; seeds are a flat uint32 array of width*height*4 words in the
reference's layout (index x*height + y, /root/reference/src/kernel/kernel0.cl:590).
"""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))

# ---- struct layouts (include/bdpt.h; the reference's primitives.h:6-64, camera.h:37-41) ----
sphere_dtype = np.dtype([("center", "<f4", 4), ("radius", "<f4"), ("surface", "<u4"), ("reserved", "<u4", 2)])
triangle_dtype = np.dtype([("p0", "<f4", 4), ("p1", "<f4", 4), ("p2", "<f4", 4), ("surface", "<u4"), ("reserved", "<u4", 3)])
plane_dtype = np.dtype([("normal", "<f4", 4), ("point", "<f4", 4), ("surface", "<u4"), ("reserved", "<u4", 3)])
surface_dtype = np.dtype([("color", "<f4", 4), ("emissive", "<f4", 4)])
camera_dtype = np.dtype([("pos", "<f4", 4), ("look_at", "<f4", 4), ("up", "<f4", 4)])
assert sphere_dtype.itemsize == 32 and triangle_dtype.itemsize == 64
assert plane_dtype.itemsize == 48 and surface_dtype.itemsize == 32 and camera_dtype.itemsize == 48

RNG_STREAM, RNG_STRIDED = 0, 1
LIGHT_ORIGIN_SELF, LIGHT_ORIGIN_NONE = 0, 1
FALLOFF_INV_SQ, FALLOFF_INV_SQ_PLUS_C = 0, 1
ACCUM_MEAN, ACCUM_SUM = 0, 1
KERNEL_AUTO, KERNEL_DIRECT, KERNEL_LDS, KERNEL_TILED, KERNEL_LDS2, KERNEL_BVH = 0, 1, 3, 4, 5, 6
PATHS_AUTO, PATHS_PER_LANE, PATHS_POOLED, PATHS_BY_CLASS = 0, 1, 2, 3
ORDER_AUTO, ORDER_INDEX, ORDER_ROULETTE_CLASS, ORDER_ADAPTIVE = 0, 1, 2, 3
SAMPLER_LAMBERT_CONE, SAMPLER_HEMISPHERE, SAMPLER_REFLECTIVE, SAMPLER_LAMBERT_IMP = 0, 1, 2, 3
INTEGRATOR_BIDIRECTIONAL, INTEGRATOR_PATH_TRACE = 0, 1
AOV_EYE_HIT, AOV_HIT_HASH, AOV_RADIANCE, AOV_RNG_DRAWS, AOV_RAYS = 0, 1, 2, 3, 4
COMBINE_AUTO, COMBINE_RCCL, COMBINE_COPY = 0, 1, 2
QUERY_NOALLOC_MAX_RAYS = 65536   # BDPT_QUERY_NOALLOC_MAX_RAYS
SCENE_CORNELL, SCENE_PLANES, SCENE_OUTDOOR, SCENE_SOUP = 0, 1, 2, 3


class Params(C.Structure):
    _fields_ = [
        ("max_light_bounces", C.c_int32),
        ("p_continue", C.c_float),
        ("blur_factor", C.c_float),
        ("indirect_only", C.c_int32),
        ("light_index", C.c_int32),
        ("rng_mode", C.c_int32),
        ("light_origin", C.c_int32),
        ("connection_falloff", C.c_int32),
        ("falloff_c", C.c_float),
        ("accum_mode", C.c_int32),
        ("aov_enable", C.c_int32),
        ("kernel_variant", C.c_int32),
        ("sample_order", C.c_int32),
        ("sampler", C.c_int32),
        ("imp_light_weight", C.c_float),
        ("imp_half_angle", C.c_float),
        ("integrator", C.c_int32),
    ]


class BdptError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__("bdpt error %d: %s" % (code, msg))
        self.code = code


def _load(name):
    # BDPT_LIB_DIR: directory of an alternative build of the libraries (A/B timing of two builds in one GPU session)
    path = os.path.join(os.environ.get("BDPT_LIB_DIR", _HERE), name)
    if not os.path.exists(path):
        raise ImportError("%s is not built; run `make` (or __graft_entry__.build()) at the repository root" % path)
    return C.CDLL(path, mode=C.RTLD_GLOBAL)


lib = _load("libbdpt_amd.so")
hostlib = _load("libbdpt_host.so")

_vp = C.c_void_p
lib.bdpt_version.restype = C.c_char_p
lib.bdpt_last_error.restype = C.c_char_p
lib.bdpt_last_error.argtypes = [_vp]
lib.bdpt_create.argtypes = [C.c_int, C.c_int, C.c_int, C.POINTER(_vp)]
lib.bdpt_destroy.argtypes = [_vp]
lib.bdpt_destroy.restype = None
lib.bdpt_default_params.argtypes = [C.POINTER(Params)]
lib.bdpt_default_params.restype = None
lib.bdpt_set_params.argtypes = [_vp, C.POINTER(Params)]
lib.bdpt_get_params.argtypes = [_vp, C.POINTER(Params)]
lib.bdpt_upload_scene.argtypes = [_vp, _vp, C.c_int, _vp, C.c_int, _vp, C.c_int, _vp, C.c_int]
lib.bdpt_upload_seeds.argtypes = [_vp, _vp]
lib.bdpt_read_seeds.argtypes = [_vp, _vp]
lib.bdpt_set_camera.argtypes = [_vp, _vp]
lib.bdpt_render.argtypes = [_vp, C.c_int]
lib.bdpt_render_async.argtypes = [_vp, C.c_int]
lib.bdpt_sync.argtypes = [_vp]
lib.bdpt_num_samples.argtypes = [_vp]
lib.bdpt_set_num_samples.argtypes = [_vp, C.c_int]
lib.bdpt_read_accum.argtypes = [_vp, _vp]
lib.bdpt_write_accum.argtypes = [_vp, _vp]
lib.bdpt_last_launch_variant.argtypes = [_vp]
lib.bdpt_hierarchy_info.argtypes = [_vp, C.POINTER(C.c_int)]
lib.bdpt_set_region.argtypes = [_vp, C.c_int, C.c_int, C.c_int, C.c_int]
lib.bdpt_set_sample_offset.argtypes = [_vp, C.c_int]
lib.bdpt_set_tile_interleave.argtypes = [_vp, C.c_int, C.c_int, C.c_int, C.c_int]
lib.bdpt_set_launch_chunks.argtypes = [_vp, C.c_int]
lib.bdpt_last_launch_chunks.argtypes = [_vp]
lib.bdpt_set_path_scheduling.argtypes = [_vp, C.c_int]
lib.bdpt_last_launch_path_scheduling.argtypes = [_vp]
lib.bdpt_two_stage_record_counts.argtypes = [_vp, C.POINTER(C.c_int)]
lib.bdpt_set_order_scratch_limit.argtypes = [_vp, C.c_size_t]
lib.bdpt_set_slice_wait_limit.argtypes = [_vp, C.c_double]
lib.bdpt_last_launch_order_table.argtypes = [_vp]
lib.bdpt_set_stream.argtypes = [_vp, _vp]
lib.bdpt_use_default_stream.argtypes = [_vp]
lib.bdpt_bind_accum.argtypes = [_vp, _vp]
lib.bdpt_accum_device_ptr.argtypes = [_vp]
lib.bdpt_accum_device_ptr.restype = _vp
lib.bdpt_read_aov.argtypes = [_vp, C.c_int, _vp]
lib.bdpt_enable_counters.argtypes = [_vp, C.c_int]
lib.bdpt_read_counters.argtypes = [_vp, C.POINTER(C.c_uint64)]
lib.bdpt_query_rays.argtypes = [_vp, _vp, _vp, C.c_int, _vp, _vp, _vp]
lib.bdpt_query_rays_noalloc.argtypes = [_vp, _vp, _vp, C.c_int, _vp, _vp, _vp]
lib.bdpt_clmath_eval.argtypes = [_vp, C.c_int, _vp, _vp, C.c_int]
lib.bdpt_last_launch_ms.argtypes = [_vp, C.POINTER(C.c_float)]

lib.bdpt_group_create.argtypes = [C.POINTER(C.c_int), C.c_int, C.c_int, C.c_int, C.c_int, C.POINTER(_vp)]
lib.bdpt_group_destroy.argtypes = [_vp]
lib.bdpt_group_destroy.restype = None
lib.bdpt_group_last_error.argtypes = [_vp]
lib.bdpt_group_last_error.restype = C.c_char_p
lib.bdpt_group_size.argtypes = [_vp]
lib.bdpt_group_member.argtypes = [_vp, C.c_int]
lib.bdpt_group_member.restype = _vp
lib.bdpt_group_combine.argtypes = [_vp]
lib.bdpt_group_set_params.argtypes = [_vp, C.POINTER(Params)]
lib.bdpt_group_upload_scene.argtypes = [_vp, _vp, C.c_int, _vp, C.c_int, _vp, C.c_int, _vp, C.c_int]
lib.bdpt_group_upload_seeds.argtypes = [_vp, _vp]
lib.bdpt_group_set_camera.argtypes = [_vp, _vp]
lib.bdpt_group_render.argtypes = [_vp, C.c_int]
lib.bdpt_group_num_samples.argtypes = [_vp]
lib.bdpt_group_read_image.argtypes = [_vp, _vp]

hostlib.bdpt_host_scene_create.restype = _vp
hostlib.bdpt_host_scene_create.argtypes = [C.c_int, C.c_int, C.c_uint32]
hostlib.bdpt_host_scene_destroy.argtypes = [_vp]
hostlib.bdpt_host_scene_destroy.restype = None
hostlib.bdpt_host_scene_counts.argtypes = [_vp, C.POINTER(C.c_int)]
for _n in ("spheres", "planes", "triangles", "surfaces"):
    getattr(hostlib, "bdpt_host_scene_" + _n).restype = _vp
    getattr(hostlib, "bdpt_host_scene_" + _n).argtypes = [_vp]
hostlib.bdpt_host_scene_default_camera.argtypes = [_vp, _vp]
hostlib.bdpt_host_scene_default_camera.restype = None
hostlib.bdpt_host_camera_reset.argtypes = [C.c_float, _vp, _vp, _vp]
hostlib.bdpt_host_camera_reset.restype = None
hostlib.bdpt_host_camera_orbit.argtypes = [C.c_float, _vp, _vp, C.c_float, C.c_float, C.c_double, C.c_double, C.c_int, _vp, _vp]
hostlib.bdpt_host_camera_orbit.restype = None
hostlib.bdpt_host_camera_zoom.argtypes = [C.c_float, _vp, _vp, C.c_float, C.c_double, C.c_int, _vp]
hostlib.bdpt_host_camera_zoom.restype = None
hostlib.bdpt_host_make_seeds.argtypes = [C.c_int, C.c_int, C.c_uint32, _vp]
hostlib.bdpt_host_make_seeds.restype = None
hostlib.bdpt_host_write_pfm.argtypes = [C.c_char_p, _vp, C.c_int, C.c_int]
hostlib.bdpt_host_scene_from_arrays.restype = _vp
hostlib.bdpt_host_scene_from_arrays.argtypes = [_vp, C.c_int, _vp, C.c_int, _vp, C.c_int, _vp, C.c_int, _vp]
hostlib.bdpt_host_scene_load.restype = _vp
hostlib.bdpt_host_scene_load.argtypes = [C.c_char_p, C.c_char_p, C.c_int]
hostlib.bdpt_host_scene_save.argtypes = [_vp, C.c_char_p, C.c_int]
hostlib.bdpt_host_scene_add_obj.argtypes = [_vp, C.c_char_p, C.c_uint32, C.c_float, _vp, C.c_char_p, C.c_int]
hostlib.bdpt_host_scene_view.argtypes = [_vp, _vp]
hostlib.bdpt_host_scene_view.restype = None
hostlib.bdpt_host_write_image.argtypes = [C.c_char_p, _vp, C.c_int, C.c_int, C.c_float, C.c_float]


def version():
    return lib.bdpt_version().decode()


def default_params(**overrides):
    p = Params()
    lib.bdpt_default_params(C.byref(p))
    for k, v in overrides.items():
        if not hasattr(p, k):
            raise AttributeError("bdpt_params has no field %r" % k)
        setattr(p, k, v)
    return p


def _ptr(a):
    return a.ctypes.data_as(_vp) if a is not None and a.size else None


def _as(a, dtype):
    a = np.ascontiguousarray(a, dtype=dtype)
    return a


class Scene:
    """Primitive arrays + default camera, built by the C++ host library."""

    def __init__(self, spheres, planes, triangles, surfaces, camera):
        self.spheres = _as(spheres, sphere_dtype)
        self.planes = _as(planes, plane_dtype)
        self.triangles = _as(triangles, triangle_dtype)
        self.surfaces = _as(surfaces, surface_dtype)
        self.camera = _as(camera, camera_dtype).reshape(())

    @staticmethod
    def build(kind, n_soup=0, seed=0xC0FFEE):
        h = hostlib.bdpt_host_scene_create(kind, n_soup, seed)
        if not h:
            raise ValueError("unknown scene kind %r" % (kind,))
        return Scene._take(h)

    @staticmethod
    def load(path, objs=()):
        """Scene file (text or binary, see host/scene_io.cpp), then OBJ files appended as triangles:
        objs = [(path, surface, scale, (tx, ty, tz)), ...]."""
        err = C.create_string_buffer(512)
        h = hostlib.bdpt_host_scene_load(os.fsencode(path), err, len(err))
        if not h:
            raise ValueError(err.value.decode())
        for obj_path, surface, scale, translate in objs:
            tr = np.asarray(translate, np.float32)
            if hostlib.bdpt_host_scene_add_obj(h, os.fsencode(obj_path), surface, scale, _ptr(tr), err, len(err)) < 0:
                hostlib.bdpt_host_scene_destroy(h)
                raise ValueError(err.value.decode())
        return Scene._take(h)

    def save(self, path, binary=False):
        """Writes the scene through the C++ host, as text or in the binary form (round-trips through Scene.load)."""
        view = np.asarray(getattr(self, "view", (1.0, 0.1, -0.1, 8.0)), np.float32)
        h = hostlib.bdpt_host_scene_from_arrays(_ptr(self.spheres), len(self.spheres), _ptr(self.planes), len(self.planes),
                                                _ptr(self.triangles), len(self.triangles), _ptr(self.surfaces),
                                                len(self.surfaces), _ptr(view))
        if not h:
            raise ValueError("bad scene arrays")
        try:
            if hostlib.bdpt_host_scene_save(h, os.fsencode(path), 1 if binary else 0) != 0:
                raise OSError("cannot write %s" % path)
        finally:
            hostlib.bdpt_host_scene_destroy(h)

    @staticmethod
    def _take(h):
        try:
            cnt = (C.c_int * 4)()
            hostlib.bdpt_host_scene_counts(h, cnt)

            def grab(name, dtype, n):
                if n == 0:
                    return np.zeros(0, dtype)
                p = getattr(hostlib, "bdpt_host_scene_" + name)(h)
                buf = (C.c_char * (n * dtype.itemsize)).from_address(p)
                return np.frombuffer(buf, dtype=dtype, count=n).copy()

            cam = np.zeros((), camera_dtype)
            hostlib.bdpt_host_scene_default_camera(h, cam.ctypes.data_as(_vp))
            view = (C.c_float * 4)()
            hostlib.bdpt_host_scene_view(h, view)
            sc = Scene(grab("spheres", sphere_dtype, cnt[0]), grab("planes", plane_dtype, cnt[1]),
                       grab("triangles", triangle_dtype, cnt[2]), grab("surfaces", surface_dtype, cnt[3]), cam)
            sc.view = tuple(view)
            return sc
        finally:
            hostlib.bdpt_host_scene_destroy(h)

    @staticmethod
    def cornell():
        return Scene.build(SCENE_CORNELL)

    @staticmethod
    def cornell_planes():
        return Scene.build(SCENE_PLANES)

    @staticmethod
    def outdoor():
        return Scene.build(SCENE_OUTDOOR)

    @staticmethod
    def soup(n_soup, seed=0xC0FFEE):
        return Scene.build(SCENE_SOUP, n_soup, seed)


def camera_reset(radius, orientation, direction):
    cam = np.zeros((), camera_dtype)
    o = np.asarray(orientation, np.float32)
    d = np.asarray(direction, np.float32)
    hostlib.bdpt_host_camera_reset(radius, _ptr(o), _ptr(d), cam.ctypes.data_as(_vp))
    return cam


def camera_orbit(radius, orientation, direction, speed_x, speed_y, dx, dy, steps):
    """Poses of ModelCamera after reset + `steps` cursor drags; returns (cams[steps+1], changed[steps])."""
    cams = np.zeros(steps + 1, camera_dtype)
    changed = np.zeros(max(steps, 1), np.int32)
    o = np.asarray(orientation, np.float32)
    d = np.asarray(direction, np.float32)
    hostlib.bdpt_host_camera_orbit(radius, _ptr(o), _ptr(d), speed_x, speed_y, dx, dy, steps, _ptr(cams), _ptr(changed))
    return cams, changed[:steps]


def camera_zoom(radius, orientation, direction, scroll_speed, dy, steps):
    """Poses of ModelCamera after reset + `steps` middle-button drags of dy; returns cams[steps+1]."""
    cams = np.zeros(steps + 1, camera_dtype)
    o = np.asarray(orientation, np.float32)
    d = np.asarray(direction, np.float32)
    hostlib.bdpt_host_camera_zoom(radius, _ptr(o), _ptr(d), scroll_speed, dy, steps, _ptr(cams))
    return cams


def make_seeds(width, height, global_seed=0x5EED):
    out = np.empty(width * height * 4, np.uint32)
    hostlib.bdpt_host_make_seeds(width, height, global_seed, _ptr(out))
    return out


def write_pfm(path, rgba):
    rgba = _as(rgba, np.float32)
    h, w = rgba.shape[:2]
    if hostlib.bdpt_host_write_pfm(os.fsencode(path), _ptr(rgba), w, h) != 0:
        raise OSError("cannot write %s" % path)


def write_image(path, rgba, scale=1.0, gamma=1.0):
    """.pfm (float), .ppm or .png (8-bit: clamp(v*scale, 0, 1) ** (1/gamma))."""
    rgba = _as(rgba, np.float32)
    h, w = rgba.shape[:2]
    if hostlib.bdpt_host_write_image(os.fsencode(path), _ptr(rgba), w, h, scale, gamma) != 0:
        raise OSError("cannot write %s" % path)


class Context:
    """One bdpt_ctx: one GPU, one image size."""

    def __init__(self, width, height, device=0):
        self._h = _vp()
        self.width, self.height = width, height
        rc = lib.bdpt_create(device, width, height, C.byref(self._h))
        if rc != 0:
            raise BdptError(rc, lib.bdpt_last_error(None).decode())

    def close(self):
        if self._h:
            lib.bdpt_destroy(self._h)
            self._h = _vp()

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _check(self, rc):
        if rc != 0:
            raise BdptError(rc, lib.bdpt_last_error(self._h).decode())

    def set_params(self, params=None, **overrides):
        p = params if params is not None else self.get_params()
        for k, v in overrides.items():
            if not hasattr(p, k):
                raise AttributeError("bdpt_params has no field %r" % k)
            setattr(p, k, v)
        self._check(lib.bdpt_set_params(self._h, C.byref(p)))

    def get_params(self):
        p = Params()
        self._check(lib.bdpt_get_params(self._h, C.byref(p)))
        return p

    def upload_scene(self, scene):
        self._check(lib.bdpt_upload_scene(self._h, _ptr(scene.spheres), len(scene.spheres), _ptr(scene.planes),
                                          len(scene.planes), _ptr(scene.triangles), len(scene.triangles),
                                          _ptr(scene.surfaces), len(scene.surfaces)))

    def upload_seeds(self, seeds):
        seeds = _as(seeds, np.uint32)
        if seeds.size != self.width * self.height * 4:
            raise ValueError("seeds must hold width*height*4 words")
        self._check(lib.bdpt_upload_seeds(self._h, _ptr(seeds)))

    def last_launch_variant(self):
        return lib.bdpt_last_launch_variant(self._h)

    def hierarchy_info(self):
        """bdpt_hierarchy_info: dict(nodes, depth, stack_entries, outside) of the hierarchy over the scene's triangles, or None"""
        info = (C.c_int * 4)()
        if not lib.bdpt_hierarchy_info(self._h, info):
            return None
        return dict(nodes=info[0], depth=info[1], stack_entries=info[2], outside=info[3])

    def write_accum(self, rgba):
        rgba = _as(rgba, np.float32)
        if rgba.shape != (self.height, self.width, 4):
            raise ValueError("accumulator must be height x width x 4")
        self._check(lib.bdpt_write_accum(self._h, _ptr(rgba)))

    def read_seeds(self):
        out = np.empty(self.width * self.height * 4, np.uint32)
        self._check(lib.bdpt_read_seeds(self._h, _ptr(out)))
        return out

    def set_camera(self, camera):
        cam = _as(camera, camera_dtype)
        self._check(lib.bdpt_set_camera(self._h, cam.ctypes.data_as(_vp)))

    def render(self, spp=1):
        self._check(lib.bdpt_render(self._h, spp))

    def render_async(self, spp=1):
        self._check(lib.bdpt_render_async(self._h, spp))

    def sync(self):
        self._check(lib.bdpt_sync(self._h))

    @property
    def num_samples(self):
        return lib.bdpt_num_samples(self._h)

    @num_samples.setter
    def num_samples(self, n):
        self._check(lib.bdpt_set_num_samples(self._h, n))

    def read_accum(self):
        out = np.empty((self.height, self.width, 4), np.float32)
        self._check(lib.bdpt_read_accum(self._h, _ptr(out)))
        return out

    def set_region(self, x0, y0, w, h):
        self._check(lib.bdpt_set_region(self._h, x0, y0, w, h))

    def set_tile_interleave(self, tile_w, tile_h, stride, phase):
        self._check(lib.bdpt_set_tile_interleave(self._h, tile_w, tile_h, stride, phase))

    def set_launch_chunks(self, chunks):
        """Runs per tile of a render call (0 = chosen per launch, 1 = unsliced); never changes a result."""
        self._check(lib.bdpt_set_launch_chunks(self._h, chunks))

    def set_path_scheduling(self, mode):
        """PATHS_PER_LANE, PATHS_POOLED (a wavefront's paths sorted by phase), PATHS_BY_CLASS (a wavefront's pixels sorted by
        the length of their next sample) or PATHS_AUTO (the library chooses); never changes a result."""
        self._check(lib.bdpt_set_path_scheduling(self._h, mode))

    def last_launch_path_scheduling(self):
        return lib.bdpt_last_launch_path_scheduling(self._h)

    def set_order_scratch_limit(self, nbytes):
        """Device memory the context may take for the table that keeps a launch's class order (0: none); never changes a result."""
        self._check(lib.bdpt_set_order_scratch_limit(self._h, nbytes))

    def two_stage_record_counts(self):
        """(parallelogram pairs, other coplanar pairs, single triangles, pairs of facing parallelograms) of the
        two-stage kernel's conservative test."""
        out = (C.c_int * 4)()
        self._check(lib.bdpt_two_stage_record_counts(self._h, out))
        return tuple(out)

    def last_launch_chunks(self):
        return lib.bdpt_last_launch_chunks(self._h)

    def set_sample_offset(self, first_sample):
        self._check(lib.bdpt_set_sample_offset(self._h, first_sample))

    def set_stream(self, stream_ptr):
        """bdpt_set_stream, with its meaning of a null handle: render on the HIP stream with this handle, 0 / None = the
        context's own stream."""
        self._check(lib.bdpt_set_stream(self._h, stream_ptr or None))

    def use_default_stream(self):
        """bdpt_use_default_stream: render on HIP's default stream (handle 0)."""
        self._check(lib.bdpt_use_default_stream(self._h))

    def use_own_stream(self):
        self._check(lib.bdpt_set_stream(self._h, None))

    def follow_stream(self, stream_ptr):
        """Render on the stream a caller's other work runs on, given as the handle torch reports
        (torch.cuda.current_stream().cuda_stream): a non-zero handle is that stream, and 0 -- which is how torch names
        HIP's default stream, and which bdpt_set_stream would read as "the context's own" -- selects the default stream."""
        if stream_ptr:
            self.set_stream(stream_ptr)
        else:
            self.use_default_stream()

    def set_slice_wait_limit(self, seconds):
        """Longest wait of a workgroup of a sliced launch for the run before it (0: derived from the expected launch time)."""
        self._check(lib.bdpt_set_slice_wait_limit(self._h, seconds))

    def last_launch_order_table(self):
        """1 if the last launch took its class order from the table, 0 if it worked the order out as it went."""
        return lib.bdpt_last_launch_order_table(self._h)

    def clear_accum(self):
        self.write_accum(np.zeros((self.height, self.width, 4), np.float32))

    def bind_accum(self, device_ptr):
        self._check(lib.bdpt_bind_accum(self._h, device_ptr))

    def read_aov(self, kind):
        shape, dt = {
            AOV_EYE_HIT: ((self.height, self.width, 2), np.int32),
            AOV_HIT_HASH: ((self.height, self.width), np.uint32),
            AOV_RADIANCE: ((self.height, self.width, 4), np.float32),
            AOV_RNG_DRAWS: ((self.height, self.width), np.uint32),
            AOV_RAYS: ((self.height, self.width), np.uint32),
        }[kind]
        out = np.empty(shape, dt)
        self._check(lib.bdpt_read_aov(self._h, kind, _ptr(out)))
        return out

    def enable_counters(self, on=True):
        self._check(lib.bdpt_enable_counters(self._h, 1 if on else 0))

    def read_counters(self):
        """(rays traced, samples that reached the light loop, samples that ended at the eye ray, cheap cull tests,
        exact triangle tests that followed a cull test) of the last counted launch."""
        c = (C.c_uint64 * 5)()
        self._check(lib.bdpt_read_counters(self._h, c))
        return tuple(int(v) for v in c)

    def query_rays(self, rays, origin_ids, noalloc=False):
        """Closest hits of caller-supplied rays; noalloc=True goes through bdpt_query_rays_noalloc."""
        rays = _as(rays, np.float32).reshape(-1, 6)
        origin_ids = _as(origin_ids, np.int32).reshape(-1, 2)
        n = rays.shape[0]
        ids = np.empty((n, 2), np.int32)
        pos = np.empty((n, 3), np.float32)
        nrm = np.empty((n, 3), np.float32)
        fn = lib.bdpt_query_rays_noalloc if noalloc else lib.bdpt_query_rays
        self._check(fn(self._h, rays.ctypes.data_as(_vp), origin_ids.ctypes.data_as(_vp), n, ids.ctypes.data_as(_vp),
                       pos.ctypes.data_as(_vp), nrm.ctypes.data_as(_vp)))
        return ids, pos, nrm

    def clmath_eval(self, op, inputs):
        x = np.zeros((len(inputs), 3), np.float64)
        inputs = np.asarray(inputs, np.float64)
        x[:, : (inputs.shape[1] if inputs.ndim == 2 else 1)] = inputs.reshape(len(inputs), -1)
        out = np.empty(len(x), np.float64)
        self._check(lib.bdpt_clmath_eval(self._h, op, _ptr(x), _ptr(out), len(x)))
        return out

    def last_launch_ms(self):
        ms = C.c_float()
        self._check(lib.bdpt_last_launch_ms(self._h, C.byref(ms)))
        return ms.value


class Group:
    """One bdpt_group: several GPUs of this node (one context each, one process), samples dealt to them."""

    def __init__(self, devices, width, height, combine=COMBINE_AUTO):
        self._h = _vp()
        self.width, self.height = width, height
        arr = (C.c_int * len(devices))(*devices)
        rc = lib.bdpt_group_create(arr, len(devices), width, height, combine, C.byref(self._h))
        if rc != 0:
            raise BdptError(rc, lib.bdpt_group_last_error(None).decode())

    def close(self):
        if self._h:
            lib.bdpt_group_destroy(self._h)
            self._h = _vp()

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    def _check(self, rc):
        if rc != 0:
            raise BdptError(rc, lib.bdpt_group_last_error(self._h).decode())

    @property
    def size(self):
        return lib.bdpt_group_size(self._h)

    @property
    def combine(self):
        return lib.bdpt_group_combine(self._h)

    @property
    def num_samples(self):
        return lib.bdpt_group_num_samples(self._h)

    def member_num_samples(self, i):
        """Samples member i has rendered since the last reset."""
        m = lib.bdpt_group_member(self._h, i)
        if not m:
            raise IndexError(i)
        return lib.bdpt_num_samples(m)

    def set_params(self, params):
        self._check(lib.bdpt_group_set_params(self._h, C.byref(params)))

    def upload_scene(self, scene):
        self._check(lib.bdpt_group_upload_scene(self._h, _ptr(scene.spheres), len(scene.spheres), _ptr(scene.planes),
                                                len(scene.planes), _ptr(scene.triangles), len(scene.triangles),
                                                _ptr(scene.surfaces), len(scene.surfaces)))

    def upload_seeds(self, seeds):
        seeds = _as(seeds, np.uint32)
        if seeds.size != self.width * self.height * 4:
            raise ValueError("seeds must hold width*height*4 words")
        self._check(lib.bdpt_group_upload_seeds(self._h, _ptr(seeds)))

    def set_camera(self, camera):
        cam = _as(camera, camera_dtype)
        self._check(lib.bdpt_group_set_camera(self._h, cam.ctypes.data_as(_vp)))

    def render(self, spp):
        self._check(lib.bdpt_group_render(self._h, spp))

    def read_image(self):
        out = np.empty((self.height, self.width, 4), np.float32)
        self._check(lib.bdpt_group_read_image(self._h, _ptr(out)))
        return out

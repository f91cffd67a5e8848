"""ctypes binding of libspira_hip.so (C ABI: include/spira_hip.h).

This is the same boundary the Julia shim (julia-spira_amd/julia/SPIRA.jl) binds with `ccall`.
There is no CPU fallback: if the HIP library is missing or no MI355X is visible, every render
call raises SpiraError.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.normpath(os.path.join(_HERE, "..", "csrc"))
LIB_PATH = os.environ.get("SPIRA_HIP_LIB", os.path.join(CSRC, "libspira_hip.so"))   # same override as julia/SPIRA.jl

ABI_VERSION = 3      # SPIRA_ABI_VERSION of the include/spira_hip.h this binding was written against (struct layouts, flag values)

# ---- flags (include/spira_hip.h) ----
SEM_A, SEM_CPU, SEM_METAL, SEM_HYBRID = 0x0, 0x1, 0x2, 0x3
KERNEL_DEFAULT, KERNEL_MEGA, KERNEL_BOUNCE, KERNEL_WAVEFRONT = 0x00, 0x10, 0x20, 0x30
POST_ACES, POST_ACES_GAMMA, POST_CLAMP_GAMMA, POST_NONE = 0x000, 0x100, 0x200, 0x300
ROWS_BOTTOM_UP = 0x1000
FLAG_PROFILE = 0x10000
EXT_DIELECTRIC, EXT_SPECTRAL = 0x20000, 0x40000

EXPORTS = [
    "spira_abi_version", "spira_build_id", "spira_last_error", "spira_device_count", "spira_set_device", "spira_get_counters",
    "spira_shutdown", "spira_camera_lookat_f32", "spira_camera_lookat_f64", "spira_render_f32", "spira_render_f64",
    "spira_render_device_f32", "spira_render_device_f64", "spira_trace_paths_f32", "spira_trace_paths_f64",
    "spira_tonemap_f32", "spira_stripe_rows", "spira_accumulate_f32", "spira_accumulate_f64", "spira_accumulate_device_f32",
    "spira_accumulate_device_f64", "spira_scene_create_f32", "spira_scene_create_f64", "spira_scene_destroy",
    "spira_render_scene_f32", "spira_render_scene_f64", "spira_render_scene_device_f32", "spira_render_scene_device_f64",
    "spira_render_multi_f32", "spira_render_multi_f64", "spira_scene_create_multi_f32", "spira_scene_create_multi_f64",
    "spira_render_multi_scene_f32", "spira_render_multi_scene_f64",
]


class SpiraError(RuntimeError):
    pass


class Params(C.Structure):
    _fields_ = [("width", C.c_uint32), ("height", C.c_uint32), ("spp", C.c_uint32), ("max_depth", C.c_uint32),
                ("n_spheres", C.c_uint32), ("n_materials", C.c_uint32), ("n_triangles", C.c_uint32),
                ("flags", C.c_uint32), ("seed", C.c_uint64), ("row0", C.c_uint32), ("rows", C.c_uint32),
                ("stripe_h", C.c_uint32), ("stripe_count", C.c_uint32), ("stripe_rank", C.c_uint32),
                ("batch_rays", C.c_uint32)]


class Counters(C.Structure):
    _fields_ = [("samples", C.c_uint64), ("segments", C.c_uint64), ("rays_enqueued", C.c_uint64),
                ("radiance_rmw", C.c_uint64), ("radiance_stores", C.c_uint64), ("passes", C.c_uint64), ("launches", C.c_uint64),
                ("kernel_ms", C.c_double), ("bounce_kernel_ms", C.c_double), ("bounce_launches", C.c_uint64), ("redone_waves", C.c_uint64), ("rays_parked", C.c_uint64),
                ("mesh_wave_trips", C.c_uint64), ("mesh_lane_trips", C.c_uint64), ("walk_kernel_ms", C.c_double)]

    def as_dict(self):
        return {k: getattr(self, k) for k, _ in self._fields_}


_lib = None


def build_library():
    """hipcc --offload-arch=gfx950 build of csrc/ (cross-compiles without a GPU)."""
    subprocess.run(["make", "-C", CSRC, "-s"], check=True)


def lib():
    """Load libspira_hip.so; raises SpiraError when it has not been built."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise SpiraError("libspira_hip.so is not built (%s); run __graft_entry__.build() / make -C %s" % (LIB_PATH, CSRC))
        try:
            import torch  # noqa: F401  (load torch's HIP runtime first so both share one libamdhip64.so.7)
        except Exception:
            pass
        _lib = C.CDLL(LIB_PATH)
        _lib.spira_last_error.restype = C.c_char_p
        _lib.spira_build_id.restype = C.c_char_p
        _lib.spira_stripe_rows.restype = C.c_uint32
        _lib.spira_stripe_rows.argtypes = [C.c_uint32] * 4
        for name in EXPORTS:
            getattr(_lib, name)  # AttributeError if an ABI symbol is missing
        have = _lib.spira_abi_version()
        if have != ABI_VERSION:      # a stale library (SPIRA_HIP_LIB, an old build directory): other struct sizes and flag values
            _lib = None
            raise SpiraError("%s has ABI version %d, this binding was written for %d" % (LIB_PATH, have, ABI_VERSION))
    return _lib


def build_id():
    """Hash of the kernel sources the loaded library was built from (csrc/Makefile); bench.py attaches PMC figures of exactly these sources."""
    return lib().spira_build_id().decode()


def _check(rc):
    if rc != 0:
        raise SpiraError("libspira_hip error %d: %s" % (rc, lib().spira_last_error().decode()))


def _dt(prec):
    if prec == "f32":
        return np.float32, C.c_float
    if prec == "f64":
        return np.float64, C.c_double
    raise ValueError("prec must be 'f32' or 'f64'")


def _arr(a, dtype):
    if a is None:
        return None, None
    a = np.ascontiguousarray(a, dtype=dtype)
    return a, a.ctypes.data_as(C.c_void_p)


def make_params(width, height, spp, max_depth, n_spheres, n_materials, n_triangles=0, flags=0, seed=0, row0=0, rows=0,
                stripe_h=0, stripe_count=0, stripe_rank=0, batch_rays=0):
    return Params(width, height, spp, max_depth, n_spheres, n_materials, n_triangles, flags, seed, row0, rows,
                  stripe_h, stripe_count, stripe_rank, batch_rays)


def device_count():
    return lib().spira_device_count()


def set_device(d):
    _check(lib().spira_set_device(C.c_int(d)))


def counters():
    c = Counters()
    _check(lib().spira_get_counters(C.byref(c)))
    return c.as_dict()


def stripe_rows(height, stripe_h, stripe_count, stripe_rank):
    return lib().spira_stripe_rows(height, stripe_h, stripe_count, stripe_rank)


def camera_lookat(position, look_at, up, fov_deg, aspect_ratio, focus_dist=1.0, prec="f32"):
    npdt, cdt = _dt(prec)
    p, pp = _arr(position, npdt)
    l, lp = _arr(look_at, npdt)
    u, up_ = _arr(up, npdt)
    out = np.zeros(12, dtype=npdt)
    if prec == "f32":
        _check(lib().spira_camera_lookat_f32(pp, lp, up_, cdt(fov_deg), cdt(aspect_ratio), out.ctypes.data_as(C.c_void_p)))
    else:
        _check(lib().spira_camera_lookat_f64(pp, lp, up_, cdt(fov_deg), cdt(aspect_ratio), cdt(focus_dist),
                                             out.ctypes.data_as(C.c_void_p)))
    return out


def render(spheres5, materials8, triangles10, camera12, params, prec="f32", want_hdr=True, want_img=False):
    """Host-pointer render.  Returns (hdr, img): arrays [3, rows, width] or None."""
    npdt, _ = _dt(prec)
    s, sp = _arr(spheres5, npdt)
    m, mp = _arr(materials8, npdt)
    t, tp = _arr(triangles10, npdt)
    c, cp = _arr(camera12, npdt)
    rows = params.rows if params.rows else params.height
    hdr = np.empty((3, rows, params.width), dtype=npdt) if want_hdr else None
    img = np.empty((3, rows, params.width), dtype=npdt) if want_img else None
    fn = lib().spira_render_f32 if prec == "f32" else lib().spira_render_f64
    _check(fn(sp, mp, tp, cp, C.byref(params), hdr.ctypes.data_as(C.c_void_p) if want_hdr else None,
              img.ctypes.data_as(C.c_void_p) if want_img else None))
    return hdr, img


def render_device(spheres5, materials8, triangles10, camera12, params, d_hdr_ptr, d_img_ptr, stream_ptr, prec="f32"):
    """Asynchronous render into DEVICE buffers (integer addresses, e.g. torch.Tensor.data_ptr())."""
    npdt, _ = _dt(prec)
    s, sp = _arr(spheres5, npdt)
    m, mp = _arr(materials8, npdt)
    t, tp = _arr(triangles10, npdt)
    c, cp = _arr(camera12, npdt)
    fn = lib().spira_render_device_f32 if prec == "f32" else lib().spira_render_device_f64
    _check(fn(sp, mp, tp, cp, C.byref(params), C.c_void_p(d_hdr_ptr or None), C.c_void_p(d_img_ptr or None),
              C.c_void_p(stream_ptr or None)))


def render_multi(spheres5, materials8, triangles10, camera12, params, n_devices, prec="f32", want_hdr=True, want_img=False):
    """spira_render_multi_*: the frame on n_devices GPUs of this node (stripes + one RCCL gather to device 0), host outputs."""
    npdt, _ = _dt(prec)
    s, sp = _arr(spheres5, npdt)
    m, mp = _arr(materials8, npdt)
    t, tp = _arr(triangles10, npdt)
    c, cp = _arr(camera12, npdt)
    hdr = np.empty((3, params.height, params.width), dtype=npdt) if want_hdr else None
    img = np.empty((3, params.height, params.width), dtype=npdt) if want_img else None
    fn = lib().spira_render_multi_f32 if prec == "f32" else lib().spira_render_multi_f64
    _check(fn(sp, mp, tp, cp, C.byref(params), C.c_int(n_devices), hdr.ctypes.data_as(C.c_void_p) if want_hdr else None,
              img.ctypes.data_as(C.c_void_p) if want_img else None))
    return hdr, img


class Scene:
    """A scene resident on the current device (spira_scene_create_* / spira_scene_destroy): validated, its BVH built
    and everything uploaded once.  Use as a context manager or call destroy()."""

    def __init__(self, spheres5, materials8, triangles10=None, prec="f32", n_devices=0):
        """n_devices >= 1: spira_scene_create_multi_* — validated and built once, resident on devices 0 .. n_devices-1 (render_multi)."""
        npdt, _ = _dt(prec)
        s, sp = _arr(spheres5, npdt)
        m, mp = _arr(materials8, npdt)
        t, tp = _arr(triangles10, npdt)
        self.prec = prec
        self.n_devices = n_devices
        self.counts = (0 if s is None else len(s), len(m), 0 if t is None else len(t))
        self._h = C.c_void_p()
        if n_devices:
            fn = lib().spira_scene_create_multi_f32 if prec == "f32" else lib().spira_scene_create_multi_f64
            _check(fn(sp, mp, tp, C.c_uint32(self.counts[0]), C.c_uint32(self.counts[1]), C.c_uint32(self.counts[2]), C.c_int(n_devices), C.byref(self._h)))
        else:
            fn = lib().spira_scene_create_f32 if prec == "f32" else lib().spira_scene_create_f64
            _check(fn(sp, mp, tp, C.c_uint32(self.counts[0]), C.c_uint32(self.counts[1]), C.c_uint32(self.counts[2]), C.byref(self._h)))

    def render_multi(self, camera12, params, n_devices=None, want_hdr=True, want_img=False):
        """spira_render_multi_scene_*: the frame on n_devices GPUs from the resident copies of this scene, host outputs."""
        npdt, _ = _dt(self.prec)
        c, cp = _arr(camera12, npdt)
        hdr = np.empty((3, params.height, params.width), dtype=npdt) if want_hdr else None
        img = np.empty((3, params.height, params.width), dtype=npdt) if want_img else None
        fn = lib().spira_render_multi_scene_f32 if self.prec == "f32" else lib().spira_render_multi_scene_f64
        _check(fn(self._h, cp, C.byref(params), C.c_int(n_devices or self.n_devices), hdr.ctypes.data_as(C.c_void_p) if want_hdr else None,
                  img.ctypes.data_as(C.c_void_p) if want_img else None))
        return hdr, img

    def destroy(self):
        if self._h:
            _check(lib().spira_scene_destroy(self._h))
            self._h = C.c_void_p()

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.destroy()

    def params(self, width, height, spp, max_depth, **kw):
        return make_params(width, height, spp, max_depth, *self.counts, **kw)

    def render(self, camera12, params, want_hdr=True, want_img=False):
        npdt, _ = _dt(self.prec)
        c, cp = _arr(camera12, npdt)
        rows = params.rows if params.rows else params.height
        hdr = np.empty((3, rows, params.width), dtype=npdt) if want_hdr else None
        img = np.empty((3, rows, params.width), dtype=npdt) if want_img else None
        fn = lib().spira_render_scene_f32 if self.prec == "f32" else lib().spira_render_scene_f64
        _check(fn(self._h, cp, C.byref(params), hdr.ctypes.data_as(C.c_void_p) if want_hdr else None,
                  img.ctypes.data_as(C.c_void_p) if want_img else None))
        return hdr, img

    def render_device(self, camera12, params, d_hdr_ptr, d_img_ptr, stream_ptr):
        npdt, _ = _dt(self.prec)
        c, cp = _arr(camera12, npdt)
        fn = lib().spira_render_scene_device_f32 if self.prec == "f32" else lib().spira_render_scene_device_f64
        _check(fn(self._h, cp, C.byref(params), C.c_void_p(d_hdr_ptr or None), C.c_void_p(d_img_ptr or None), C.c_void_p(stream_ptr or None)))


def accumulate(spheres5, materials8, triangles10, camera12, params, sample0, sum_rgb, rng_states=None, prec="f32"):
    """Progressive accumulation: adds samples [sample0, sample0 + params.spp) to sum_rgb ([3, rows, W], in place)."""
    npdt, _ = _dt(prec)
    s, sp = _arr(spheres5, npdt)
    m, mp = _arr(materials8, npdt)
    t, tp = _arr(triangles10, npdt)
    c, cp = _arr(camera12, npdt)
    assert sum_rgb.dtype == npdt and sum_rgb.flags["C_CONTIGUOUS"]
    rp = None
    if rng_states is not None:
        assert rng_states.dtype == np.uint32 and rng_states.flags["C_CONTIGUOUS"]
        rp = rng_states.ctypes.data_as(C.c_void_p)
    fn = lib().spira_accumulate_f32 if prec == "f32" else lib().spira_accumulate_f64
    _check(fn(sp, mp, tp, cp, C.byref(params), C.c_uint32(sample0), sum_rgb.ctypes.data_as(C.c_void_p), rp))
    return sum_rgb


def trace_paths(spheres5, materials8, triangles10, camera12, params, ijs, prec="f32"):
    """Diagnostic: per-segment (prims, ts, dirs) and radiance of the paths ijs = [[i, j, sample], ...]."""
    npdt, _ = _dt(prec)
    s, sp = _arr(spheres5, npdt)
    m, mp = _arr(materials8, npdt)
    t, tp = _arr(triangles10, npdt)
    c, cp = _arr(camera12, npdt)
    ij = np.ascontiguousarray(ijs, dtype=np.uint32).reshape(-1, 3)
    n, d = ij.shape[0], params.max_depth
    prims = np.zeros((n, d), dtype=np.int32)
    ts = np.zeros((n, d), dtype=npdt)
    dirs = np.zeros((n, d, 3), dtype=npdt)
    rad = np.zeros((n, 3), dtype=npdt)
    fn = lib().spira_trace_paths_f32 if prec == "f32" else lib().spira_trace_paths_f64
    _check(fn(sp, mp, tp, cp, C.byref(params), C.c_uint32(n), ij.ctypes.data_as(C.c_void_p),
              prims.ctypes.data_as(C.c_void_p), ts.ctypes.data_as(C.c_void_p), dirs.ctypes.data_as(C.c_void_p),
              rad.ctypes.data_as(C.c_void_p)))
    return prims, ts, dirs, rad


def tonemap(values, post):
    v = np.ascontiguousarray(values, dtype=np.float32).copy()
    _check(lib().spira_tonemap_f32(v.ctypes.data_as(C.c_void_p), C.c_uint64(v.size), C.c_uint32(post)))
    return v

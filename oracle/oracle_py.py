"""ctypes loader for oracle/libspira_oracle.so — TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg import this module.
The product package (julia-spira_amd/) never does.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None


class SpiraParams(C.Structure):
    """Mirror of spira_params (include/spira_hip.h); the product binding has its own copy."""
    _fields_ = [("width", C.c_uint32), ("height", C.c_uint32), ("spp", C.c_uint32), ("max_depth", C.c_uint32),
                ("n_spheres", C.c_uint32), ("n_materials", C.c_uint32), ("n_triangles", C.c_uint32),
                ("flags", C.c_uint32), ("seed", C.c_uint64), ("row0", C.c_uint32), ("rows", C.c_uint32),
                ("stripe_h", C.c_uint32), ("stripe_count", C.c_uint32), ("stripe_rank", C.c_uint32),
                ("batch_rays", C.c_uint32)]


def build():
    subprocess.run(["make", "-C", _HERE, "-s"], check=True)


def lib():
    global _LIB
    if _LIB is None:
        path = os.path.join(_HERE, "libspira_oracle.so")
        if not os.path.exists(path):
            build()
        _LIB = C.CDLL(path)
        _LIB.oracle_lcg_next.restype = C.c_uint32
        _LIB.oracle_lcg_next.argtypes = [C.c_uint32]
        _LIB.oracle_lcg_uniform.restype = C.c_float
        _LIB.oracle_lcg_uniform.argtypes = [C.c_uint32]
        _LIB.oracle_xorshift32.restype = C.c_uint32
        _LIB.oracle_xorshift32.argtypes = [C.c_uint32]
        _LIB.oracle_xorshift_uniform.restype = C.c_float
        _LIB.oracle_xorshift_uniform.argtypes = [C.c_uint32]
        _LIB.oracle_mix32_export.restype = C.c_uint32
        _LIB.oracle_mix32_export.argtypes = [C.c_uint32]
        _LIB.oracle_post_f32.restype = C.c_float
        _LIB.oracle_post_f32.argtypes = [C.c_float, C.c_uint32]
        _LIB.oracle_post_f64.restype = C.c_double
        _LIB.oracle_post_f64.argtypes = [C.c_double, C.c_uint32]
    return _LIB


def _dt(prec):
    return (np.float64, C.c_double, "_f64") if prec == "f64" else (np.float32, C.c_float, "_f32")


def _arr(a, dtype):
    if a is None:
        return None, None
    a = np.ascontiguousarray(a, dtype=dtype)
    return a, a.ctypes.data_as(C.c_void_p)


def make_params(width, height, spp, max_depth, n_spheres, n_materials, n_triangles=0, flags=0, seed=0,
                row0=0, rows=0, stripe_h=0, stripe_count=0, stripe_rank=0):
    return SpiraParams(width, height, spp, max_depth, n_spheres, n_materials, n_triangles, flags, seed,
                       row0, rows, stripe_h, stripe_count, stripe_rank, 0)


def render(spheres5, materials8, triangles10, camera12, params, prec="f64", n_threads=0, want_img=False):
    """Returns (hdr[3,rows,W], img or None, segments)."""
    npdt, cdt, suf = _dt(prec)
    s, sp = _arr(spheres5, npdt)
    m, mp = _arr(materials8, npdt)
    t, tp = _arr(triangles10, npdt)
    c, cp = _arr(camera12, npdt)
    rows = params.rows if params.rows else params.height
    hdr = np.empty((3, rows, params.width), dtype=npdt)
    img = np.empty((3, rows, params.width), dtype=npdt) if want_img else None
    seg = C.c_uint64(0)
    fn = getattr(lib(), "oracle_render" + suf)
    fn.restype = C.c_int
    rc = fn(sp, mp, tp, cp, C.byref(params), hdr.ctypes.data_as(C.c_void_p),
            img.ctypes.data_as(C.c_void_p) if want_img else None, C.c_int(n_threads), C.byref(seg))
    if rc != 0:
        raise RuntimeError("oracle_render%s failed: %d" % (suf, rc))
    return hdr, img, seg.value


def trace_path(spheres5, materials8, triangles10, camera12, params, i, j, sample, prec="f32"):
    """One path: (n_segments, prims[max_depth], ts, dirs[max_depth,3], radiance[3]); i, j 1-based."""
    npdt, cdt, suf = _dt(prec)
    s, sp = _arr(spheres5, npdt)
    m, mp = _arr(materials8, npdt)
    t, tp = _arr(triangles10, npdt)
    c, cp = _arr(camera12, npdt)
    d = params.max_depth
    prims = np.zeros(d, dtype=np.int32)
    ts = np.zeros(d, dtype=npdt)
    dirs = np.zeros((d, 3), dtype=npdt)
    rad = np.zeros(3, dtype=npdt)
    fn = getattr(lib(), "oracle_trace_path" + suf)
    fn.restype = C.c_int
    n = fn(sp, mp, tp, cp, C.byref(params), C.c_uint32(i), C.c_uint32(j), C.c_uint32(sample),
           prims.ctypes.data_as(C.c_void_p), ts.ctypes.data_as(C.c_void_p), dirs.ctypes.data_as(C.c_void_p),
           rad.ctypes.data_as(C.c_void_p))
    return n, prims, ts, dirs, rad


def render_variant(spheres5, materials8, camera12, params, prec="f32", n_threads=0, want_img=False):
    """SPIRA_SEM_CPU / SPIRA_SEM_METAL restatements (params.flags selects).  Returns (hdr, img or None, segments)."""
    npdt, cdt, suf = _dt(prec)
    s, sp = _arr(spheres5, npdt)
    m, mp = _arr(materials8, npdt)
    c, cp = _arr(camera12, npdt)
    rows = params.rows if params.rows else params.height
    hdr = np.empty((3, rows, params.width), dtype=npdt)
    img = np.empty((3, rows, params.width), dtype=npdt) if want_img else None
    seg = C.c_uint64(0)
    fn = getattr(lib(), "oracle_render_variant" + suf)
    fn.restype = C.c_int
    rc = fn(sp, mp, cp, C.byref(params), hdr.ctypes.data_as(C.c_void_p), img.ctypes.data_as(C.c_void_p) if want_img else None,
            C.c_int(n_threads), C.byref(seg))
    if rc != 0:
        raise RuntimeError("oracle_render_variant%s failed: %d" % (suf, rc))
    return hdr, img, seg.value


def render_hybrid(spheres5, materials8, camera12, params, prec="f32", n_threads=0):
    """SPIRA_SEM_HYBRID: render_hybrid_gpu of src/spira-metal-optimized.jl:1228-1343 as written (whole images only).  Returns (image, segments);
    the image is the reference's: the mean of per-sample tone-mapped colours, rows in the order the flags ask for."""
    npdt, cdt, suf = _dt(prec)
    s, sp = _arr(spheres5, npdt)
    m, mp = _arr(materials8, npdt)
    c, cp = _arr(camera12, npdt)
    hdr = np.empty((3, params.height, params.width), dtype=npdt)
    seg = C.c_uint64(0)
    fn = getattr(lib(), "oracle_render_hybrid" + suf)
    fn.restype = C.c_int
    rc = fn(sp, mp, cp, C.byref(params), hdr.ctypes.data_as(C.c_void_p), None, C.c_int(n_threads), C.byref(seg))
    if rc != 0:
        raise RuntimeError("oracle_render_hybrid%s failed: %d" % (suf, rc))
    return hdr, seg.value


def trace_path_variant(spheres5, materials8, camera12, params, i, j, sample, prec="f32"):
    npdt, cdt, suf = _dt(prec)
    s, sp = _arr(spheres5, npdt)
    m, mp = _arr(materials8, npdt)
    c, cp = _arr(camera12, npdt)
    d = params.max_depth
    prims = np.zeros(d, dtype=np.int32)
    ts = np.zeros(d, dtype=npdt)
    dirs = np.zeros((d, 3), dtype=npdt)
    rad = np.zeros(3, dtype=npdt)
    fn = getattr(lib(), "oracle_trace_path_variant" + suf)
    fn.restype = C.c_int
    n = fn(sp, mp, cp, C.byref(params), C.c_uint32(i), C.c_uint32(j), C.c_uint32(sample), prims.ctypes.data_as(C.c_void_p),
           ts.ctypes.data_as(C.c_void_p), dirs.ctypes.data_as(C.c_void_p), rad.ctypes.data_as(C.c_void_p))
    return n, prims, ts, dirs, rad


def sincos_turn(r, prec="f64"):
    npdt, cdt, suf = _dt(prec)
    out = np.zeros(2, dtype=npdt)
    fn = getattr(lib(), "oracle_sincos_turn" + suf)
    fn.restype = None
    fn(cdt(r), out.ctypes.data_as(C.c_void_p))
    return out


def camera(position, look_at, up, fov_deg, aspect_ratio, focus_dist=1.0, prec="f64"):
    npdt, cdt, suf = _dt(prec)
    p, pp = _arr(position, npdt)
    l, lp = _arr(look_at, npdt)
    u, up_ = _arr(up, npdt)
    out = np.zeros(12, dtype=npdt)
    fn = getattr(lib(), "oracle_camera" + suf)
    fn.restype = None
    fn(pp, lp, up_, cdt(fov_deg), cdt(aspect_ratio), cdt(focus_dist), out.ctypes.data_as(C.c_void_p))
    return out


def hit_sphere(s5, o, d, t_min, t_max, prec="f64"):
    npdt, cdt, suf = _dt(prec)
    s, sp = _arr(s5, npdt)
    o_, op = _arr(o, npdt)
    d_, dp = _arr(d, npdt)
    t = cdt(0)
    n = np.zeros(3, dtype=npdt)
    fn = getattr(lib(), "oracle_hit_sphere" + suf)
    fn.restype = C.c_int
    hit = fn(sp, op, dp, cdt(t_min), cdt(t_max), C.byref(t), n.ctypes.data_as(C.c_void_p))
    return bool(hit), t.value, n


def hit_triangle(t10, o, d, t_min, t_max, prec="f64"):
    npdt, cdt, suf = _dt(prec)
    s, sp = _arr(t10, npdt)
    o_, op = _arr(o, npdt)
    d_, dp = _arr(d, npdt)
    t = cdt(0)
    n = np.zeros(3, dtype=npdt)
    fn = getattr(lib(), "oracle_hit_triangle" + suf)
    fn.restype = C.c_int
    hit = fn(sp, op, dp, cdt(t_min), cdt(t_max), C.byref(t), n.ctypes.data_as(C.c_void_p))
    return bool(hit), t.value, n


def sky(d, prec="f64"):
    npdt, cdt, suf = _dt(prec)
    d_, dp = _arr(d, npdt)
    out = np.zeros(3, dtype=npdt)
    fn = getattr(lib(), "oracle_sky" + suf)
    fn.restype = None
    fn(dp, out.ctypes.data_as(C.c_void_p))
    return out


def post(x, post_flag, prec="f64"):
    return getattr(lib(), "oracle_post_" + prec)(x, post_flag)


def rng_try(seed, pixel, sample, bounce, t, prec="f64"):
    npdt, cdt, suf = _dt(prec)
    out = np.zeros(3, dtype=npdt)
    fn = getattr(lib(), "oracle_rng_try" + suf)
    fn.restype = None
    fn(C.c_uint64(seed), C.c_uint32(pixel), C.c_uint32(sample), C.c_uint32(bounce), C.c_uint32(t),
       out.ctypes.data_as(C.c_void_p))
    return out


def max_threads():
    fn = lib().oracle_max_threads
    fn.restype = C.c_int
    return fn()

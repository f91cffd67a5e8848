"""Deterministic synthetic workloads (flat C-ABI arrays) — BASELINE.md §3 / SURVEY.md §8d.

S1  create_scene() of src/spira-metal-optimized.jl:429-510 + Camera((0,1,3),(0,0,0),(0,1,0),40,16/9)
S2  create_scene() of examples/julia-raytracer.jl:605-641 (5 spheres + 1 triangle, fov 45)
S3  S1 enclosed in a large diffuse sphere: no path can escape, every path runs max_depth segments
Each builder returns a dict(spheres5, materials8, triangles10, camera12) of float64 arrays whose
values are exactly what the reference's constructors hold (Float32 values for S1/S3, Float64
literals for S2); cast to the render precision at the call.
"""
import numpy as np

from . import _binding as B


def _f32(x):
    return np.asarray(x, dtype=np.float32).astype(np.float64)


def scene_s1(aspect=16.0 / 9.0):
    # materials: albedo rgb, emission rgb, metallic, roughness  (src/spira-metal-optimized.jl:430-470)
    materials8 = _f32([
        [0.7, 0.3, 0.3, 0, 0, 0, 0.0, 0.5],   # diffuse red
        [0.5, 0.5, 0.5, 0, 0, 0, 0.0, 0.9],   # ground
        [0.8, 0.8, 0.8, 0, 0, 0, 1.0, 0.0],   # metal
        [0.8, 0.8, 1.0, 0, 0, 0, 0.9, 0.0],   # "glass-like" (a tinted mirror)
        [1.0, 1.0, 1.0, 5, 5, 5, 0.0, 0.0],   # light
    ])
    # spheres: cx cy cz r material(1-based)  (:472-507)
    spheres5 = _f32([
        [0.0, 0.0, 0.0, 0.5, 1],
        [0.0, -100.5, 0.0, 100.0, 2],
        [1.0, 0.0, 0.0, 0.5, 3],
        [-1.0, 0.0, 0.0, 0.5, 4],
        [0.0, 5.0, 0.0, 1.0, 5],
    ])
    cam = B.camera_lookat([0, 1, 3], [0, 0, 0], [0, 1, 0], 40.0, np.float32(aspect), prec="f32").astype(np.float64)
    return dict(spheres5=spheres5, materials8=materials8, triangles10=None, camera12=cam)


def scene_s2():
    # examples/julia-raytracer.jl:605-641; Material defaults specular=0.0 roughness=1.0 (:60-61)
    materials8 = np.array([
        [0.8, 0.8, 0.2, 0, 0, 0, 0.0, 1.0],   # ground
        [0.8, 0.2, 0.2, 0, 0, 0, 0.0, 1.0],   # red
        [0.8, 0.6, 0.2, 0, 0, 0, 0.8, 0.3],   # gold
        [0.8, 0.8, 0.8, 0, 0, 0, 1.0, 0.0],   # mirror ("glass-like")
        [0.8, 0.8, 0.8, 4, 4, 4, 0.0, 1.0],   # light
        [0.2, 0.8, 0.2, 0, 0, 0, 0.0, 1.0],   # green triangle
    ], dtype=np.float64)
    spheres5 = np.array([
        [0, -100.5, -1, 100, 1],
        [0, 0, -1, 0.5, 2],
        [1, 0, -1, 0.5, 3],
        [-1, 0, -1, 0.5, 4],
        [0, 2, 0, 0.5, 5],
    ], dtype=np.float64)
    triangles10 = np.array([[-0.5, 0, -2, 0.5, 0, -2, 0, 1, -2, 6]], dtype=np.float64)
    cam = B.camera_lookat([0.0, 1.0, 3.0], [0.0, 0.0, -1.0], [0.0, 1.0, 0.0], 45.0, 16.0 / 9.0, 1.0, prec="f64")
    return dict(spheres5=spheres5, materials8=materials8, triangles10=triangles10, camera12=cam)


def _wall(center, a, b, material):
    """One oversized triangle covering a whole box wall; its geometric normal normalize(cross(e1,e2))
    (examples/julia-raytracer.jl:105-109) is a x b and must point INTO the box, because semantics A
    never flips normals: a diffuse bounce leaves along n + random_in_unit_sphere (:356)."""
    c, a, b = np.asarray(center, float), np.asarray(a, float), np.asarray(b, float)
    v0 = c - 30.0 * a - 30.0 * b
    return list(v0) + list(v0 + 120.0 * a) + list(v0 + 120.0 * b) + [material]


def scene_s3(aspect=16.0 / 9.0):
    """S1 inside a closed box x,z in [-8,8], y in [-1,9] made of 6 single-triangle walls (no seams inside
    the box, neighbouring walls overlap beyond the edges): no path can reach the sky."""
    s = scene_s1(aspect)
    s["materials8"] = np.vstack([s["materials8"], _f32([[0.8, 0.8, 0.8, 0, 0, 0, 0.0, 1.0]])])
    X, Y, Z = [1, 0, 0], [0, 1, 0], [0, 0, 1]
    s["triangles10"] = _f32([
        _wall([-8, 4, 0], Y, Z, 6),   # normal +x
        _wall([8, 4, 0], Z, Y, 6),    # normal -x
        _wall([0, -1, 0], Z, X, 6),   # normal +y (floor; mostly hidden under the ground sphere)
        _wall([0, 9, 0], X, Z, 6),    # normal -y
        _wall([0, 4, -8], X, Y, 6),   # normal +z
        _wall([0, 4, 8], Y, X, 6),    # normal -z
    ])
    return s


def seed_for(config_index):
    return 0x5EED0001 + config_index

"""Deterministic synthetic workloads (flat C-ABI arrays) — BASELINE.md §3 / SURVEY.md §8d.

S1  create_scene() of src/spira-metal-optimized.jl:429-510 + Camera((0,1,3),(0,0,0),(0,1,0),40,16/9)
S2  create_scene() of examples/julia-raytracer.jl:605-641 (5 spheres + 1 triangle, fov 45)
S3  S1 enclosed in a large diffuse sphere: no path can escape, every path runs max_depth segments
S4  create_scene_with_obj() of examples/julia-raytracer.jl:644-706 with a procedural 81 920-triangle mesh (configs[4])
S5  S4 seen from close up: the mesh fills 70 % of the frame — the stress scene of the mesh path (the S3 of meshes)
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


def scene_s2_glass():
    """S2 with the mirror sphere turned into glass (ior 1.5), the gold one into tinted glass (ior 1.33) and a glass triangle (ior 1.1): the scene of the
    SPIRA_EXT_DIELECTRIC / SPIRA_EXT_SPECTRAL extension (include/spira_hip.h: a material with NEGATIVE roughness is a dielectric of index -roughness).
    No such scene exists in the reference (README.md:10 only names the features): parity unpinned."""
    s = scene_s2()
    m = s["materials8"].copy()
    m[3] = [0.95, 0.95, 0.95, 0, 0, 0, 0.0, -1.5]
    m[2] = [0.9, 0.7, 0.3, 0, 0, 0, 0.0, -1.33]
    m[5] = [0.8, 1.0, 0.8, 0, 0, 0, 0.0, -1.1]
    s["materials8"] = m
    return s


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


# ---------------------------------------------------------------------------------------- meshes
def icosphere(level):
    """Unit icosphere: (vertices [n,3], faces [m,3] 0-based); level 6 -> 81 920 triangles.  The synthetic
    stand-in for the bunny of tests/bunny-test.jl:17-20 (a URL download, unavailable offline); same idea as
    the generator of examples/spira-metal-raytracer.jl:258-309."""
    t = (1.0 + 5.0 ** 0.5) / 2.0
    v = [(-1, t, 0), (1, t, 0), (-1, -t, 0), (1, -t, 0), (0, -1, t), (0, 1, t), (0, -1, -t), (0, 1, -t),
         (t, 0, -1), (t, 0, 1), (-t, 0, -1), (-t, 0, 1)]
    f = [(0, 11, 5), (0, 5, 1), (0, 1, 7), (0, 7, 10), (0, 10, 11), (1, 5, 9), (5, 11, 4), (11, 10, 2), (10, 7, 6), (7, 1, 8),
         (3, 9, 4), (3, 4, 2), (3, 2, 6), (3, 6, 8), (3, 8, 9), (4, 9, 5), (2, 4, 11), (6, 2, 10), (8, 6, 7), (9, 8, 1)]
    verts = [np.array(p, dtype=np.float64) / np.linalg.norm(p) for p in v]
    faces = list(f)
    for _ in range(level):
        cache, nf = {}, []

        def mid(a, b):
            key = (a, b) if a < b else (b, a)
            if key not in cache:
                m = verts[a] + verts[b]
                verts.append(m / np.linalg.norm(m))
                cache[key] = len(verts) - 1
            return cache[key]
        for a, b, c in faces:
            ab, bc, ca = mid(a, b), mid(b, c), mid(c, a)
            nf += [(a, ab, ca), (b, bc, ab), (c, ca, bc), (ab, bc, ca)]
        faces = nf
    return np.array(verts), np.array(faces, dtype=np.int64)


def bumpy_blob(level=6):
    """Icosphere displaced radially by a smooth lobed field: a non-convex closed mesh with concavities."""
    v, f = icosphere(level)
    r = 1.0 + 0.18 * np.sin(5.0 * v[:, 0]) * np.sin(4.0 * v[:, 1] + 1.0) * np.sin(3.0 * v[:, 2] + 2.0) + 0.1 * np.sin(9.0 * v[:, 1])
    return v * r[:, None], f


def transform_vertices(vertices, scale=(1.0, 1.0, 1.0), rotation=(0.0, 0.0, 0.0), translation=(0.0, 0.0, 0.0), center=True,
                       normalize_size=False):
    """The vertex pipeline of load_obj_mesh, examples/julia-raytracer.jl:510-591, in its order:
    bbox-centre -> normalise by the largest bbox dimension -> rotate X, Y, Z (degrees) -> scale -> translate."""
    v = np.array(vertices, dtype=np.float64)
    if center or normalize_size:                                           # :511
        mn, mx = v.min(axis=0), v.max(axis=0)
        center_point = (mn + mx) / 2.0                                     # :522
        max_dimension = float((mx - mn).max())                             # :523-524
        if center:
            v = v - center_point                                           # :527-531
        if normalize_size and max_dimension > 0:
            v = v * (1.0 / max_dimension)                                  # :534-539
    rx, ry, rz = rotation
    if rx != 0 or ry != 0 or rz != 0:                                      # :543
        if rx != 0:
            th = rx * (np.pi / 180.0)                                      # deg2rad
            y = v[:, 1] * np.cos(th) - v[:, 2] * np.sin(th)
            z = v[:, 1] * np.sin(th) + v[:, 2] * np.cos(th)
            v = np.stack([v[:, 0], y, z], axis=1)                          # :548-553
        if ry != 0:
            th = ry * (np.pi / 180.0)
            x = v[:, 0] * np.cos(th) + v[:, 2] * np.sin(th)
            z = -v[:, 0] * np.sin(th) + v[:, 2] * np.cos(th)
            v = np.stack([x, v[:, 1], z], axis=1)                          # :556-561
        if rz != 0:
            th = rz * (np.pi / 180.0)
            x = v[:, 0] * np.cos(th) - v[:, 1] * np.sin(th)
            y = v[:, 0] * np.sin(th) + v[:, 1] * np.cos(th)
            v = np.stack([x, y, v[:, 2]], axis=1)                          # :564-569
    if tuple(scale) != (1.0, 1.0, 1.0):
        v = v * np.asarray(scale, dtype=np.float64)                        # :576-584
    if tuple(translation) != (0.0, 0.0, 0.0):
        v = v + np.asarray(translation, dtype=np.float64)                  # :587-591
    return v


def mesh_triangles10(vertices, faces, material_index):
    v, f = np.asarray(vertices, dtype=np.float64), np.asarray(faces)
    return np.concatenate([v[f[:, 0]], v[f[:, 1]], v[f[:, 2]], np.full((len(f), 1), float(material_index))], axis=1)


def scene_s4(level=6):
    """create_scene_with_obj() of examples/julia-raytracer.jl:644-706 with the procedural blob in place of the
    bunny: ground sphere + light sphere + mesh (material (.7,.3,.2) specular .2 roughness .4), the mesh centred,
    normalised, rotated 90 deg about Y, scaled .5, moved to z = -1; camera of :697-703.  Level 6 = 81 920 triangles."""
    materials8 = np.array([
        [0.8, 0.8, 0.2, 0, 0, 0, 0.0, 1.0],   # ground (:649)
        [0.8, 0.8, 0.8, 4, 4, 4, 0.0, 1.0],   # light (:652)
        [0.7, 0.3, 0.2, 0, 0, 0, 0.2, 0.4],   # mesh_material (:655)
    ], dtype=np.float64)
    spheres5 = np.array([[0, -100.5, -1, 100, 1], [0, 2, 0, 0.5, 2]], dtype=np.float64)
    v, f = bumpy_blob(level)
    v = transform_vertices(v, scale=(0.5, 0.5, 0.5), rotation=(0.0, 90.0, 0.0), translation=(0.0, 0.0, -1.0), center=True,
                           normalize_size=True)                           # :671-679
    cam = B.camera_lookat([0.0, 1.0, 3.0], [0.0, 0.0, -1.0], [0.0, 1.0, 0.0], 45.0, 16.0 / 9.0, 1.0, prec="f64")
    return dict(spheres5=spheres5, materials8=materials8, triangles10=mesh_triangles10(v, f, 3), camera12=cam)


def scene_s5(level=6):
    """Mesh-dominant stress scene: the objects of S4, the camera 16 cm in front of the mesh, looking at its centre (vfov 45): the mesh
    covers 70 % of the frame (in S4 it covers 0.7 %), so nearly every camera ray walks the tree, and every bounce off the mesh starts
    inside the mesh's box and walks it again.  Not a reference scene: what S3 is to the sphere path, this is to G16."""
    s = scene_s4(level)
    s["camera12"] = B.camera_lookat([0.0, 0.1, -0.6], [0.0, 0.0, -1.0], [0.0, 1.0, 0.0], 45.0, 16.0 / 9.0, 1.0, prec="f64")
    return s

"""Host-side mirror of the surface of examples/julia-raytracer.jl (the parity variant "A").

Same names / keyword arguments / return values as the reference script, executed by the HIP
backend: `render(world, camera, width, height; samples_per_pixel, max_depth)` returns
`(img, hdr_data)` with row 0 = image top, like :387-421.  Float64 by default (the reference's
precision); `precision="f32"` selects the Float32 kernels.
"""
import numpy as np

from . import _binding as B


class Vec3:
    """struct Vec3 (Float64) examples/julia-raytracer.jl:11-15."""
    __slots__ = ("x", "y", "z")

    def __init__(self, x, y, z):
        self.x, self.y, self.z = float(x), float(y), float(z)

    def tolist(self):
        return [self.x, self.y, self.z]

    def __repr__(self):
        return "(%r, %r, %r)" % (self.x, self.y, self.z)   # show(), :29


class Material:
    """Material(; diffuse=Vec3(.8,.8,.8), emission=Vec3(0,0,0), specular=0.0, roughness=1.0) (:53-62)."""

    def __init__(self, diffuse=None, emission=None, specular=0.0, roughness=1.0):
        self.diffuse = diffuse if diffuse is not None else Vec3(0.8, 0.8, 0.8)
        self.emission = emission if emission is not None else Vec3(0.0, 0.0, 0.0)
        self.specular = float(specular)
        self.roughness = float(roughness)


class Sphere:
    """struct Sphere <: Hittable (center, radius, material) (:77-81)."""

    def __init__(self, center, radius, material):
        self.center, self.radius, self.material = center, float(radius), material


class Triangle:
    """struct Triangle <: Hittable (vertices::Vector{Vec3}, material) (:84-94)."""

    def __init__(self, vertices, material):
        assert len(vertices) == 3, "Triangle must have exactly 3 vertices"   # :90
        self.vertices, self.material = list(vertices), material


class BoundingVolumeHierarchy:
    """struct BoundingVolumeHierarchy <: Hittable: stores the list as-is (:231-239)."""

    def __init__(self, objects):
        self.objects = list(objects)


HittableList = BoundingVolumeHierarchy   # same linear scan (:190-210)


class Camera:
    """Camera(; position, look_at, up, fov=90.0, aspect_ratio=16/9, aperture=0.0, focus_dist=1.0) (:261-295)."""

    def __init__(self, position=None, look_at=None, up=None, fov=90.0, aspect_ratio=16.0 / 9.0, aperture=0.0, focus_dist=1.0):
        position = position if position is not None else Vec3(0, 0, 0)
        look_at = look_at if look_at is not None else Vec3(0, 0, -1)
        up = up if up is not None else Vec3(0, 1, 0)
        c = B.camera_lookat(position.tolist(), look_at.tolist(), up.tolist(), fov, aspect_ratio, focus_dist, prec="f64")
        self.position = position
        self.lower_left_corner, self.horizontal, self.vertical = Vec3(*c[3:6]), Vec3(*c[6:9]), Vec3(*c[9:12])
        self.lens_radius = aperture / 2   # stored, unused (:293, :299-300)
        self._flat = c

    def flat(self):
        return self._flat


def create_scene():
    """create_scene() (:605-641) -> (scene, camera)."""
    objects = [
        Sphere(Vec3(0, -100.5, -1), 100, Material(diffuse=Vec3(0.8, 0.8, 0.2))),
        Sphere(Vec3(0, 0, -1), 0.5, Material(diffuse=Vec3(0.8, 0.2, 0.2))),
        Sphere(Vec3(1, 0, -1), 0.5, Material(diffuse=Vec3(0.8, 0.6, 0.2), specular=0.8, roughness=0.3)),
        Sphere(Vec3(-1, 0, -1), 0.5, Material(diffuse=Vec3(0.8, 0.8, 0.8), specular=1.0, roughness=0.0)),
        Sphere(Vec3(0, 2, 0), 0.5, Material(diffuse=Vec3(0.8, 0.8, 0.8), emission=Vec3(4, 4, 4))),
        Triangle([Vec3(-0.5, 0, -2), Vec3(0.5, 0, -2), Vec3(0, 1, -2)], Material(diffuse=Vec3(0.2, 0.8, 0.2))),
    ]
    scene = BoundingVolumeHierarchy(objects)
    camera = Camera(position=Vec3(0.0, 1.0, 3.0), look_at=Vec3(0.0, 0.0, -1.0), up=Vec3(0.0, 1.0, 0.0), fov=45.0,
                    aspect_ratio=16.0 / 9.0)
    return scene, camera


def load_obj_mesh(filename, material, scale=None, rotation=None, translation=None, center=True, normalize_size=False):
    """load_obj_mesh(filename, material; scale, rotation, translation, center, normalize_size) (:466-602).

    Parses `v` and `f` records (first index of `a/b/c`, fan triangulation of n-gons :498-505), applies the
    transform pipeline in the reference's order (scenes.transform_vertices) and returns the Triangle list."""
    from .scenes import transform_vertices
    vertices, faces = [], []
    with open(filename, "r") as fh:
        for line in fh:
            if line.startswith("v "):                                   # :478-484
                parts = line.split()
                vertices.append((float(parts[1]), float(parts[2]), float(parts[3])))
            elif line.startswith("f "):                                 # :485-505
                idx = [int(tok.split("/")[0]) for tok in line.split()[1:]]
                if len(idx) == 3:
                    faces.append(idx)
                elif len(idx) > 3:
                    for i in range(2, len(idx)):
                        faces.append([idx[0], idx[i - 1], idx[i]])
    v = transform_vertices(np.array(vertices, dtype=np.float64).reshape(-1, 3),
                           scale=tuple(scale.tolist()) if scale is not None else (1.0, 1.0, 1.0),
                           rotation=tuple(rotation.tolist()) if rotation is not None else (0.0, 0.0, 0.0),
                           translation=tuple(translation.tolist()) if translation is not None else (0.0, 0.0, 0.0),
                           center=center, normalize_size=normalize_size)
    tris = []
    for a, b, c in faces:                                               # 1-based OBJ indices (:490, :597)
        tris.append(Triangle([Vec3(*v[a - 1]), Vec3(*v[b - 1]), Vec3(*v[c - 1])], material))
    return tris


def create_scene_with_obj(obj_file=None):
    """create_scene_with_obj() (:644-706): ground + light + the OBJ mesh, or a sphere when the file is missing (:687-691)."""
    import os
    objects = [
        Sphere(Vec3(0, -100.5, -1), 100, Material(diffuse=Vec3(0.8, 0.8, 0.2))),
        Sphere(Vec3(0, 2, 0), 0.5, Material(diffuse=Vec3(0.8, 0.8, 0.8), emission=Vec3(4, 4, 4))),
    ]
    mesh_material = Material(diffuse=Vec3(0.7, 0.3, 0.2), specular=0.2, roughness=0.4)
    obj_file = obj_file or os.path.expanduser("~/Downloads/bunny.obj")
    if os.path.isfile(obj_file):
        objects += load_obj_mesh(obj_file, mesh_material, center=True, normalize_size=True, scale=Vec3(0.5, 0.5, 0.5),
                                 rotation=Vec3(0.0, 90.0, 0.0), translation=Vec3(0.0, 0.0, -1.0))
    else:
        print("OBJ file not found, adding a sphere instead")
        objects.insert(2, Sphere(Vec3(0, 0, -1), 0.5, mesh_material))
    scene = BoundingVolumeHierarchy(objects)
    camera = Camera(position=Vec3(0.0, 1.0, 3.0), look_at=Vec3(0.0, 0.0, -1.0), up=Vec3(0.0, 1.0, 0.0), fov=45.0, aspect_ratio=16.0 / 9.0)
    return scene, camera


def flatten_world(world):
    """Hittable list -> the C-ABI flat arrays.  The ABI intersects spheres first, then triangles;
    a list that interleaves them differently is reordered only if that cannot change a result,
    i.e. never silently: mixed orders raise."""
    spheres, tris, mats, mat_index = [], [], [], {}
    seen_triangle = False
    for obj in world.objects:
        m = obj.material
        if id(m) not in mat_index:        # one table entry per distinct Material object (a mesh shares one, :598)
            mats.append(m.diffuse.tolist() + m.emission.tolist() + [m.specular, m.roughness])
            mat_index[id(m)] = len(mats)
        idx = mat_index[id(m)]
        if isinstance(obj, Sphere):
            if seen_triangle:
                raise ValueError("the C ABI scans spheres before triangles; list spheres first")
            spheres.append(obj.center.tolist() + [obj.radius, idx])
        elif isinstance(obj, Triangle):
            seen_triangle = True
            v = obj.vertices
            tris.append(v[0].tolist() + v[1].tolist() + v[2].tolist() + [idx])
        else:
            raise TypeError("unsupported Hittable: %r" % (obj,))
    return (np.array(spheres, dtype=np.float64).reshape(-1, 5), np.array(mats, dtype=np.float64).reshape(-1, 8),
            np.array(tris, dtype=np.float64).reshape(-1, 10) if tris else None)


def to_acescg(color):
    """to_acescg(color) (:370-384) on an (..., 3) array, via the library's host transform."""
    a = np.asarray(color, dtype=np.float32)
    return B.tonemap(a, B.POST_ACES).reshape(a.shape)


def render(world, camera, width, height, samples_per_pixel=50, max_depth=20, seed=0, precision="f64", kernel=B.KERNEL_WAVEFRONT):
    """render(world, camera, width, height; samples_per_pixel=50, max_depth=20) (:387-421).

    Returns (img, hdr_data): img (H, W, 3) Float32 after to_acescg, hdr_data (H, W, 3) linear means.
    """
    spheres5, materials8, triangles10 = flatten_world(world)
    p = B.make_params(width, height, samples_per_pixel, max_depth, len(spheres5), len(materials8),
                      0 if triangles10 is None else len(triangles10), flags=B.SEM_A | kernel | B.POST_ACES, seed=seed)
    hdr, img = B.render(spheres5, materials8, triangles10, camera.flat(), p, prec=precision, want_hdr=True, want_img=True)
    return (np.ascontiguousarray(np.moveaxis(img, 0, -1)).astype(np.float32), np.ascontiguousarray(np.moveaxis(hdr, 0, -1)))


def save_exr(hdr_data, filename):
    """save_exr(hdr_data, filename) (:424-463): 32-bit float EXR of the linear radiance; on failure the
    to_acescg PNG next to it and False, like the reference's fallback branch."""
    from . import exr, png
    try:
        exr.save_exr(filename, hdr_data)
        return True
    except (OSError, ValueError) as e:
        print("Error saving EXR file: %s\nSaving as PNG instead..." % (e,))
        png.save_png(filename.replace(".exr", ".png"), to_acescg(hdr_data))
        return False

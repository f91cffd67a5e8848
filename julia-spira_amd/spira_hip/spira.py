"""Host-side mirror of the SPIRA package surface (src/SPIRA.jl:11-13, src/spira-metal-optimized.jl).

Same names, argument meaning and defaults as the reference's Julia API, implemented over the
C ABI of libspira_hip.so (the executable twin of julia-spira_amd/julia/SPIRA.jl, whose `ccall`s
bind the same symbols).  `render` dispatches to the HIP backend where the reference's
`render` (:1453-1490) picks Metal/CUDA; there is no CPU fallback here — `render_with_cpu`
(:1346-1450) keeps its name and estimator but runs on the GPU too.
"""
import time

import numpy as np

from . import _binding as B
from .png import save_png

Float32 = np.float32


def Vec3(x, y, z):
    """const Vec3 = SVector{3, Float32} (:282); Point3 and Color are aliases (:283-284)."""
    return np.array([x, y, z], dtype=np.float32)


Point3 = Vec3
Color = Vec3

INF = np.float32(1e20)   # :287
EPS = np.float32(1e-6)   # :288
BLACK = Vec3(0.0, 0.0, 0.0)
WHITE = Vec3(1.0, 1.0, 1.0)


class Ray:
    """struct Ray (:293-298): the constructor normalises the direction."""

    def __init__(self, origin, direction):
        self.origin = np.asarray(origin, dtype=np.float32)
        d = np.asarray(direction, dtype=np.float32)
        self.direction = d / np.sqrt(np.dot(d, d), dtype=np.float32)


def at(ray, t):
    return ray.origin + np.float32(t) * ray.direction   # :300-302


class Sphere:
    """struct Sphere(center::Point3, radius::Float32, material::Int) (:305-311); material is 1-based."""

    def __init__(self, center, radius, material):
        self.center = np.asarray(center, dtype=np.float32)
        self.radius = np.float32(radius)
        self.material = int(material)


class Material:
    """Material(albedo; emission=BLACK, metallic=0f0, roughness=0.5f0) (:314-322)."""

    def __init__(self, albedo, emission=BLACK, metallic=0.0, roughness=0.5):
        self.albedo = np.asarray(albedo, dtype=np.float32)
        self.emission = np.asarray(emission, dtype=np.float32)
        self.metallic = np.float32(metallic)
        self.roughness = np.float32(roughness)


class Camera:
    """Camera(lookfrom, lookat, vup, vfov, aspect_ratio) (:325-348): origin, lower_left_corner,
    horizontal, vertical in Float32 (arithmetic done by spira_camera_lookat_f32)."""

    def __init__(self, lookfrom, lookat, vup, vfov, aspect_ratio):
        c = B.camera_lookat(lookfrom, lookat, vup, float(vfov), float(aspect_ratio), prec="f32")
        self.origin, self.lower_left_corner, self.horizontal, self.vertical = c[0:3], c[3:6], c[6:9], c[9:12]

    def flat(self):
        return np.concatenate([self.origin, self.lower_left_corner, self.horizontal, self.vertical]).astype(np.float32)


class Scene:
    """struct Scene(spheres, materials) (:351-354)."""

    def __init__(self, spheres, materials):
        self.spheres = list(spheres)
        self.materials = list(materials)


def create_scene():
    """create_scene() (:429-510): 5 materials, 5 spheres."""
    materials = [
        Material(Vec3(0.7, 0.3, 0.3), emission=BLACK, metallic=0.0, roughness=0.5),
        Material(Vec3(0.5, 0.5, 0.5), emission=BLACK, metallic=0.0, roughness=0.9),
        Material(Vec3(0.8, 0.8, 0.8), emission=BLACK, metallic=1.0, roughness=0.0),
        Material(Vec3(0.8, 0.8, 1.0), emission=BLACK, metallic=0.9, roughness=0.0),
        Material(Vec3(1.0, 1.0, 1.0), emission=Vec3(5.0, 5.0, 5.0), metallic=0.0, roughness=0.0),
    ]
    spheres = [
        Sphere(Point3(0.0, 0.0, 0.0), 0.5, 1),
        Sphere(Point3(0.0, -100.5, 0.0), 100.0, 2),
        Sphere(Point3(1.0, 0.0, 0.0), 0.5, 3),
        Sphere(Point3(-1.0, 0.0, 0.0), 0.5, 4),
        Sphere(Point3(0.0, 5.0, 0.0), 1.0, 5),
    ]
    return Scene(spheres, materials)


def prepare_scene_data(scene):
    """prepare_scene_data(scene) (:515-542): (sphere_data[5n], material_data[8m]) as flat Float32."""
    sphere_data = np.zeros(5 * len(scene.spheres), dtype=np.float32)
    for i, s in enumerate(scene.spheres):
        sphere_data[5 * i:5 * i + 3] = s.center
        sphere_data[5 * i + 3] = s.radius
        sphere_data[5 * i + 4] = np.float32(s.material)
    material_data = np.zeros(8 * len(scene.materials), dtype=np.float32)
    for i, m in enumerate(scene.materials):
        material_data[8 * i:8 * i + 3] = m.albedo
        material_data[8 * i + 3:8 * i + 6] = m.emission
        material_data[8 * i + 6] = m.metallic
        material_data[8 * i + 7] = m.roughness
    return sphere_data, material_data


SEMANTICS = {
    "A": (B.SEM_A, B.POST_ACES_GAMMA),        # ray_color of examples/julia-raytracer.jl, K7 display transform (:1128-1144)
    "cpu": (B.SEM_CPU, B.POST_CLAMP_GAMMA),   # trace_ray of render_with_cpu (:1346-1450): what render() runs on an AMD box today
    "metal": (B.SEM_METAL, B.POST_ACES_GAMMA),  # path_trace of src/spira_path_trace_kernel.metal:140-269
    "hybrid": (B.SEM_HYBRID, B.POST_NONE),    # the host loop of render_hybrid_gpu itself (:1274-1341) AS WRITTEN: last bounce only, tone map per sample
}


def render_hybrid_gpu(width, height, scene, camera, samples_per_pixel=16, max_depth=4, seed=0, post=None,
                      kernel=B.KERNEL_WAVEFRONT, semantics="A"):
    """render_hybrid_gpu(width, height, scene, camera; samples_per_pixel, max_depth) (:1228-1343).

    Returns an (H, W, 3) Float32 image, row 0 = image top (finalize_image_from_gpu_buffer :1157-1190),
    after a display transform.  `seed` is new: the reference never seeds.  `semantics` picks which of the
    reference's estimators runs: "A" (default, the parity oracle), "cpu", "metal", or "hybrid" — the host loop of :1274-1341
    itself, as written (it shades only the last bounce, tone-maps every sample and ends a sample when no ray of the image
    hits anything: what the reference shows on a Metal / CUDA machine; `post` is ignored for it).
    """
    sphere_data, material_data = prepare_scene_data(scene)
    sem, default_post = SEMANTICS[semantics]
    p = B.make_params(width, height, samples_per_pixel, max_depth, len(scene.spheres), len(scene.materials), 0,
                      flags=sem | kernel | (default_post if post is None else post), seed=seed)
    _, img = B.render(sphere_data, material_data, None, camera.flat(), p, prec="f32", want_hdr=False, want_img=True)
    return np.ascontiguousarray(np.moveaxis(img, 0, -1))


def render_with_cpu(width, height, scene, camera, samples_per_pixel=16, max_depth=4, seed=0):
    """render_with_cpu(width, height, scene, camera; samples_per_pixel, max_depth) (:1346-1450), exported by
    src/SPIRA.jl:13.  Same estimator (`trace_ray`, :1351-1412) and the same clamp + sqrt display transform
    (:1441-1442) — executed by the HIP kernels, not on the CPU: there is no CPU renderer in this package."""
    return render_hybrid_gpu(width, height, scene, camera, samples_per_pixel=samples_per_pixel, max_depth=max_depth, seed=seed,
                             semantics="cpu")


def render(scene, camera, width, height, samples_per_pixel=16, max_depth=4, output_path="metal_optimized_render.png",
           seed=0, semantics="A"):
    """render(scene, camera, width, height; samples_per_pixel=16, max_depth=4, output_path) (:1453-1490)."""
    start = time.time()
    print("Rendering with HIP GPU (MI355X, GPU-side accumulation)...")
    img = render_hybrid_gpu(width, height, scene, camera, samples_per_pixel=samples_per_pixel, max_depth=max_depth, seed=seed,
                            semantics=semantics)
    print("Render completed in %.2f seconds" % (time.time() - start))
    if output_path:
        save_png(output_path, img)
        print("Saved render to %s" % output_path)
    return img

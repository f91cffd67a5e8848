"""spira_hip — host-side mirror of SPIRA's render surface over libspira_hip.so (MI355X / gfx950).

Exports follow src/SPIRA.jl:11-13 (plus Point3/Vec3/Color, which README.md:50-53 and
examples/basic_render.jl:22-25 use but the reference forgets to export).
"""
from . import _binding, raytracer, scenes  # noqa: F401
from ._binding import SpiraError, build_library  # noqa: F401
from .spira import (BLACK, WHITE, Camera, Color, Material, Point3, Ray, Scene, Sphere, Vec3, create_scene,  # noqa: F401
                    prepare_scene_data, render, render_hybrid_gpu, render_with_cpu)

__all__ = ["Scene", "Camera", "Ray", "Sphere", "Material", "Point3", "Vec3", "Color", "render_hybrid_gpu", "render_with_cpu",
           "render", "create_scene", "prepare_scene_data", "SpiraError"]

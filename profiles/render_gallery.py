#!/usr/bin/env python3
"""Renders the synthetic scenes with the HIP path and writes PNGs (run on the GPU box; outputs under gpurun_out/renders,
copied into docs/renders/ for the record).  Display transform: to_acescg (examples/julia-raytracer.jl:370-384)."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "julia-spira_amd")]
import numpy as np  # noqa: E402
from spira_hip import _binding as B, scenes  # noqa: E402
from spira_hip.png import save_png  # noqa: E402

out = os.path.join(ROOT, "gpurun_out", "renders")
os.makedirs(out, exist_ok=True)
def glass_s2():
    """S2 with the mirror sphere turned into glass (ior 1.5), the gold one into amber glass (ior 1.33) — SPIRA_EXT_DIELECTRIC."""
    s = scenes.scene_s2()
    m = s["materials8"].copy()
    m[3] = [0.95, 0.95, 0.95, 0, 0, 0, 0.0, -1.5]
    m[2] = [0.9, 0.7, 0.3, 0, 0, 0, 0.0, -1.33]
    s["materials8"] = m
    return s


jobs = [("s2_ext_dielectric", glass_s2(), B.EXT_DIELECTRIC, 10), ("s2_ext_dielectric_spectral", glass_s2(), B.EXT_DIELECTRIC | B.EXT_SPECTRAL, 10),
        ("s1_semA", scenes.scene_s1(), 0, 8), ("s1_semCPU", scenes.scene_s1(), 1, 8), ("s1_semMETAL", scenes.scene_s1(), 2, 8),
        ("s2_semA", scenes.scene_s2(), 0, 8), ("s3_closed_box", scenes.scene_s3(), 0, 8), ("s4_mesh_bvh", scenes.scene_s4(6), 0, 12),
        ("s5_mesh_stress", scenes.scene_s5(6), 0, 12),                 # round 4: the mesh stress scene (the mesh fills 70 % of the frame)
        ("s1_semHYBRID", scenes.scene_s1(), 3, 4)]                     # round 4: render_hybrid_gpu's own estimator as written (last bounce only, tone map per sample)
if len(sys.argv) > 1:
    jobs = [j for j in jobs if j[0] in sys.argv[1:]]
for name, s, sem, depth in jobs:
    ns, nm = len(s["spheres5"]), len(s["materials8"])
    nt = 0 if s["triangles10"] is None else len(s["triangles10"])
    p = B.make_params(640, 360, 256, depth, ns, nm, nt, flags=sem | B.POST_ACES, seed=scenes.seed_for(3))
    t = time.time()
    _, img = B.render(s["spheres5"], s["materials8"], s["triangles10"], s["camera12"], p, "f32", want_hdr=False, want_img=True)
    c = B.counters()
    save_png(os.path.join(out, name + ".png"), np.moveaxis(img, 0, -1))
    print("%-16s 640x360 spp256 depth%-2d  %.1f ms kernels  %.0f Msamples/s  %.2f segments/sample" %
          (name, depth, c["kernel_ms"], c["samples"] / c["kernel_ms"] / 1e3, c["segments"] / c["samples"]), flush=True)

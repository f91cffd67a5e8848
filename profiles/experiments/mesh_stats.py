#!/usr/bin/env python3
"""Config 5 (mesh scene) with the in-kernel traversal counters of the `make stats` build (csrc/libspira_hip_stats.so; load it
with SPIRA_HIP_LIB=...): where a wave's time goes, how full the traversal sessions run.  usage: mesh_stats.py [f32|f64] [reps] [s4|s5]"""
import ctypes as C
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [os.path.join(ROOT, "julia-spira_amd")]
os.environ.setdefault("SPIRA_HIP_LIB", os.path.join(ROOT, "julia-spira_amd", "csrc", "libspira_hip_stats.so"))
import torch  # noqa: E402
from spira_hip import _binding as B, scenes  # noqa: E402

prec = sys.argv[1] if len(sys.argv) > 1 else "f32"
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 3
scene = sys.argv[3] if len(sys.argv) > 3 else "s4"        # s4 = BASELINE configs[4]; s5 = the same mesh filling 70 % of the frame
s = scenes.scene_s5() if scene == "s5" else scenes.scene_s4()
W, H, spp, depth = 1920, 1080, 64, 12
params = B.make_params(W, H, spp, depth, len(s["spheres5"]), len(s["materials8"]), len(s["triangles10"]), flags=B.KERNEL_WAVEFRONT | B.POST_NONE, seed=scenes.seed_for(5))
out = torch.empty((3, H, W), dtype=torch.float32 if prec == "f32" else torch.float64, device="cuda")
h = B.Scene(s["spheres5"], s["materials8"], s["triangles10"], prec)
st = torch.cuda.current_stream()
for _ in range(3):
    h.render_device(s["camera12"], params, out.data_ptr(), 0, st.cuda_stream)
torch.cuda.synchronize()
lib = B.lib()
buf = (C.c_ulonglong * 32)()
assert lib.spira_debug_mesh_stats(buf, 1) == 0
t0 = time.perf_counter()
for _ in range(reps):
    h.render_device(s["camera12"], params, out.data_ptr(), 0, st.cuda_stream)
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / reps
c = B.counters()
assert lib.spira_debug_mesh_stats(buf, 0) == 0
names = ["wave_cycles", "session_cycles", "sessions", "wave_steps", "lane_steps", "refill_blocks", "rays", "rounds", "trips_le8_lanes", "walk_cycles", "waves", "first_sess_wave_steps", "first_sess_lane_steps", "trips_9_24", "trips_25_48", "trips_49_64"]
print("%s %s: %.2f ms/frame, k_path %.2f ms; env %s" % (scene, prec, dt * 1e3, c["bounce_kernel_ms"], {k: v_ for k, v_ in os.environ.items() if k.startswith("SPIRA_") and k != "SPIRA_HIP_LIB"}))
for label, base in (("first / only launch", 0), ("second launch (fat waves)", 16)):
    v = [x / reps for x in buf[base:base + 16]]
    if not v[0]:
        continue
    print(" %s: " % label + ", ".join("%s %.4g" % (n, x) for n, x in zip(names, v)))
    print("   session share of wave cycles %.3f, walk share %.3f; lane utilisation in the walk %.3f; steps/ray %.2f; cycles per wave-step %.0f; rays/session %.1f; "
          "rounds/wave %.1f; mean wave lifetime %.0f cycles"
          % (v[1] / v[0], v[9] / v[0], v[4] / max(v[3], 1) / 64, v[4] / max(v[6], 1), v[9] / max(v[3], 1), v[6] / max(v[2], 1), v[7] / max(v[10], 1), v[0] / max(v[10], 1)))

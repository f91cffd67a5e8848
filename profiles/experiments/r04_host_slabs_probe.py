"""Where a host-output frame's time goes: slabs on / off, fresh / reused host buffers (GPU box).  usage: python profiles/experiments/r04_host_slabs_probe.py [f64|f32]"""
import ctypes as C
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..", "julia-spira_amd"))
from spira_hip import _binding as B, scenes  # noqa: E402

prec = sys.argv[1] if len(sys.argv) > 1 else "f64"
npdt = np.float64 if prec == "f64" else np.float32
s = scenes.scene_s1()
arrs = [np.ascontiguousarray(s[k], dtype=npdt) for k in ("spheres5", "materials8")]
cam = np.ascontiguousarray(s["camera12"], dtype=npdt)
W, H = 1920, 1080
p = B.make_params(W, H, 64, 8, len(arrs[0]), len(arrs[1]), 0, flags=B.POST_NONE, seed=3)
fn = B.lib().spira_render_f64 if prec == "f64" else B.lib().spira_render_f32
ptr = lambda a: a.ctypes.data_as(C.c_void_p)


def call(out):
    rc = fn(ptr(arrs[0]), ptr(arrs[1]), None, ptr(cam), C.byref(p), ptr(out), None)
    assert rc == 0, rc


for slabs, pf in (("0", "0"), ("0", "1"), ("4", "0"), ("4", "1"), ("2", "1")):
    os.environ["SPIRA_PREFAULT"] = pf
    os.environ["SPIRA_HOST_SLABS"] = slabs
    reused = np.empty((3, H, W), dtype=npdt)
    call(reused); call(reused)
    t = time.perf_counter()
    for _ in range(5):
        call(reused)
    t_reused = (time.perf_counter() - t) / 5
    dev = B.counters()["kernel_ms"]
    t = time.perf_counter()
    for _ in range(5):
        call(np.empty((3, H, W), dtype=npdt))
    t_fresh = (time.perf_counter() - t) / 5
    print("slabs", slabs, "prefault", pf, prec, "reused buffer %.2f ms, fresh buffer %.2f ms, device span %.2f ms" % (t_reused * 1e3, t_fresh * 1e3, dev), flush=True)
os.environ["SPIRA_LOG_TIMING"] = "1"
for slabs in ("0", "4"):
    os.environ["SPIRA_HOST_SLABS"] = slabs
    call(reused)

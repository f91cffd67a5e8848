"""configs[4] through host arrays (Float64 HDR out): whole frame vs row slabs.  Needs a library whose SPIRA_HOST_SLABS also splits mesh scenes."""
import ctypes as C, os, sys, time
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..", "julia-spira_amd"))
from spira_hip import _binding as B, scenes  # noqa: E402
m = scenes.scene_s4()
a = [np.ascontiguousarray(m[k], dtype=np.float64) for k in ("spheres5", "materials8", "triangles10", "camera12")]
p = B.make_params(1920, 1080, 64, 12, len(a[0]), len(a[1]), len(a[2]), flags=B.POST_NONE, seed=5)
ptr = lambda x: x.ctypes.data_as(C.c_void_p)
out = np.empty((3, 1080, 1920))
for slabs in ("0", "2", "3", "4", "0", "2"):
    os.environ["SPIRA_HOST_SLABS"] = slabs
    for _ in range(2):
        assert B.lib().spira_render_f64(ptr(a[0]), ptr(a[1]), ptr(a[2]), ptr(a[3]), C.byref(p), ptr(out), None) == 0
    t = time.perf_counter()
    for _ in range(5):
        assert B.lib().spira_render_f64(ptr(a[0]), ptr(a[1]), ptr(a[2]), ptr(a[3]), C.byref(p), ptr(out), None) == 0
    print("slabs", slabs, "%.2f ms per call, device span %.2f ms" % ((time.perf_counter() - t) / 5 * 1e3, B.counters()["kernel_ms"]), flush=True)

"""Can a memory-bound stream (what k_resolve is) run UNDER the persistent path kernel when it sits on a high-priority stream?
(resolve_overlap_probe.py, round 2: on an ordinary second stream it cannot.)  Prints ms per frame: renders alone, renders with a same-sized
read on a second stream of default / high priority issued right after each render call, and the read alone."""
import sys, time
sys.path[:0] = ["julia-spira_amd"]
import torch
from spira_hip import _binding as B, scenes
s = scenes.scene_s1()
for prec, tdt in (("f64", torch.float64), ("f32", torch.float32)):
    out = torch.empty((3, 1080, 1920), dtype=tdt, device="cuda")
    p = B.make_params(1920, 1080, 64, 8, 5, 5, 0, flags=B.POST_NONE, seed=1)
    h = B.Scene(s["spheres5"], s["materials8"], None, prec)
    nbytes = 1920 * 1080 * 64 * 3 * (8 if prec == "f64" else 4)
    src = torch.empty(nbytes // 4, dtype=torch.float32, device="cuda").normal_()
    def run(n, side):
        torch.cuda.synchronize(); t = time.perf_counter()
        for i in range(n):
            h.render_device(s["camera12"], p, out.data_ptr(), 0, torch.cuda.current_stream().cuda_stream)
            if side is not None:
                with torch.cuda.stream(side):
                    src.sum()
        torch.cuda.synchronize(); return (time.perf_counter() - t) / n * 1e3
    run(3, None)
    a = run(20, None)
    b = run(20, torch.cuda.Stream())
    c = run(20, torch.cuda.Stream(priority=-1))
    torch.cuda.synchronize(); t = time.perf_counter()
    for i in range(20): src.sum()
    torch.cuda.synchronize(); d = (time.perf_counter() - t) / 20 * 1e3
    print(prec, "render alone %.3f ms | + %.2f GB read, default-priority stream %.3f | high-priority stream %.3f | the read alone %.3f" % (a, nbytes / 1e9, b, c, d))
    h.destroy()

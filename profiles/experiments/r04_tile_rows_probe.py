"""Does a tile's cost per segment depend on its row count?  Contiguous slabs through the spheres (S1, Float64, spp 512), k_path time per million segments."""
import sys, time
sys.path.insert(0, "julia-spira_amd")
import torch
from spira_hip import _binding as B, scenes
s = scenes.scene_s1()
H, W = 1080, 1920
sc = B.Scene(s["spheres5"], s["materials8"], None, "f64")
st = torch.cuda.current_stream().cuda_stream
for spp in (512, 64):
    for rows in (120, 128, 130, 132, 134, 135, 136, 137, 138, 140, 144, 150, 160):
        p = B.make_params(W, H, spp, 8, 5, 5, 0, flags=B.KERNEL_WAVEFRONT | B.POST_NONE, seed=3, row0=470, rows=rows)
        out = torch.empty((3, rows, W), dtype=torch.float64, device="cuda")
        for _ in range(2):
            sc.render_device(s["camera12"], p, out.data_ptr(), 0, st)
        torch.cuda.synchronize()
        ks = []
        for _ in range(4):
            sc.render_device(s["camera12"], p, out.data_ptr(), 0, st)
            torch.cuda.synchronize()
            ks.append(B.counters()["bounce_kernel_ms"])
        c = B.counters()
        k = min(ks)
        print("spp %3d rows %3d: k_path %.3f ms (4 runs: %s), %.1f M segments, %.3f us per k-segment, passes %d" %
              (spp, rows, k, " ".join("%.3f" % x for x in ks), c["segments"] / 1e6, k * 1e3 / (c["segments"] / 1e3), c["passes"]), flush=True)

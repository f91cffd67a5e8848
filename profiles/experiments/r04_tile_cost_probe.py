"""Why a rank's tile at world 8 (spp 512 on 128-136 rows) costs more per sample than the whole frame at spp 64: variants on one GPU (S1, Float64)."""
import sys, time
sys.path.insert(0, "julia-spira_amd")
import torch
from spira_hip import _binding as B, scenes, distributed as D
s = scenes.scene_s1()
H, W = 1080, 1920
sc = B.Scene(s["spheres5"], s["materials8"], None, "f64")
st = torch.cuda.current_stream().cuda_stream
cases = [("frame spp 64", 64, {}), ("frame spp 512 (8 passes)", 512, {}), ("rows 0..135 contiguous, spp 512", 512, dict(row0=0, rows=136)),
         ("rows 472..607 contiguous, spp 512", 512, dict(row0=472, rows=136)), ("striped tile rank 0 of 8, spp 512", 512, D.tile_params(H, 8, 0)),
         ("striped tile rank 0 of 8, spp 64", 64, D.tile_params(H, 8, 0)), ("striped tile rank 3 of 8, stripes of 1 row, spp 512", 512, D.tile_params(H, 8, 3, 1)),
         ("rows 0..539 contiguous, spp 128", 128, dict(row0=0, rows=540))]
cases += [("striped tile rank 0 of 8, spp 512, 64 slots per pass", 512, dict(D.tile_params(H, 8, 0), batch_rays=64 * 136 * W)),
          ("striped tile rank 0 of 8, spp 512, 128 slots per pass", 512, dict(D.tile_params(H, 8, 0), batch_rays=128 * 136 * W)),
          ("striped tile rank 0 of 8, spp 512, 256 slots per pass", 512, dict(D.tile_params(H, 8, 0), batch_rays=256 * 136 * W)),
          ("frame spp 64, 16 slots per pass", 64, dict(batch_rays=16 * H * W)),
          ("striped tile rank 0 of 2, spp 128", 128, D.tile_params(H, 2, 0)), ("striped tile rank 0 of 4, spp 256", 256, D.tile_params(H, 4, 0))]
for name, spp, tile in cases:
    rows = tile.get("rows") or H
    p = B.make_params(W, H, spp, 8, 5, 5, 0, flags=B.KERNEL_WAVEFRONT | B.POST_NONE, seed=3, **tile)
    out = torch.empty((3, rows, W), dtype=torch.float64, device="cuda")
    for _ in range(3):
        sc.render_device(s["camera12"], p, out.data_ptr(), 0, st)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(5):
        sc.render_device(s["camera12"], p, out.data_ptr(), 0, st)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 5
    c = B.counters()
    n = rows * W * spp
    print("%-52s %7.3f ms  k_path %7.3f ms  passes %d  %6.0f Msamples/s  k_path ns per ksample %.2f  segments/sample %.3f" %
          (name, dt * 1e3, c["bounce_kernel_ms"], c["passes"], n / dt / 1e6, c["bounce_kernel_ms"] * 1e6 / (n / 1e3), c["segments"] / c["samples"]), flush=True)

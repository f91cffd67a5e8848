import sys, time
sys.path.insert(0, "julia-spira_amd")
import numpy as np, torch
from spira_hip import _binding as B, scenes, distributed as D
s = scenes.scene_s1()
H, W = 1080, 1920
for world, rank in ((1, 0), (8, 0), (8, 7), (4, 1), (2, 1)):
    tile = D.tile_params(H, world, rank)
    rows = tile["rows"] or H
    p = B.make_params(W, H, 64 * world, 8, 5, 5, 0, flags=B.KERNEL_WAVEFRONT | B.POST_NONE, seed=3, **tile)
    out = torch.empty((3, rows, W), dtype=torch.float64, device="cuda")
    sc = B.Scene(s["spheres5"], s["materials8"], None, "f64")
    for _ in range(4):
        sc.render_device(s["camera12"], p, out.data_ptr(), 0, torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(6):
        sc.render_device(s["camera12"], p, out.data_ptr(), 0, torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 6
    c = B.counters()
    print("world %d rank %d: rows %d spp %d  %.3f ms/step  kernel %.3f ms (k_path %.3f)  %.0f Msamples/s per GPU  redone %d" % (world, rank, rows, 64 * world, dt * 1e3, c["kernel_ms"], c["bounce_kernel_ms"], rows * W * 64 * world / dt / 1e6, c["redone_waves"]), flush=True)

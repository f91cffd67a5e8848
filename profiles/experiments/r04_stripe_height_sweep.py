"""World 8 on one GPU: every rank's tile (S1, Float64, spp 512 = weak scaling of configs[2]) for several stripe heights — the step of an 8-GPU run
costs what its slowest rank's render costs.  usage: r04_stripe_height_sweep.py [world] [scene s1|s3|s4] [stripe heights, comma separated]"""
import sys, time
sys.path.insert(0, "julia-spira_amd")
import torch
from spira_hip import _binding as B, scenes, distributed as D
world = int(sys.argv[1]) if len(sys.argv) > 1 else 8
name = sys.argv[2] if len(sys.argv) > 2 else "s1"
s = {"s1": scenes.scene_s1, "s3": scenes.scene_s3, "s4": scenes.scene_s4}[name]()
nt = 0 if s["triangles10"] is None else len(s["triangles10"])
H, W = 1080, 1920
sc = B.Scene(s["spheres5"], s["materials8"], s["triangles10"], "f64")
st = torch.cuda.current_stream().cuda_stream
spp = 256 if name == "s3" else 64 * world
depth = 12 if name == "s4" else 8
def run(tile):
    rows = tile.get("rows") or H
    p = B.make_params(W, H, spp, depth, len(s["spheres5"]), len(s["materials8"]), nt, flags=B.KERNEL_WAVEFRONT | B.POST_NONE, seed=3, **tile)
    out = torch.empty((3, rows, W), dtype=torch.float64, device="cuda")
    for _ in range(2):
        sc.render_device(s["camera12"], p, out.data_ptr(), 0, st)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(4):
        sc.render_device(s["camera12"], p, out.data_ptr(), 0, st)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / 4 * 1e3, rows
full, _ = run({}) if name == "s1" else (0.0, 0)
if name == "s1":
    p1 = B.make_params(W, H, 64, 8, 5, 5, 0, flags=B.KERNEL_WAVEFRONT | B.POST_NONE, seed=3)
    out = torch.empty((3, H, W), dtype=torch.float64, device="cuda")
    sc.render_device(s["camera12"], p1, out.data_ptr(), 0, st); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(4):
        sc.render_device(s["camera12"], p1, out.data_ptr(), 0, st)
    torch.cuda.synchronize()
    print("one GPU, whole frame spp 64: %.3f ms" % ((time.perf_counter() - t0) / 4 * 1e3), flush=True)
for sh in ((1, 2, 4, 8, 16, 27, 45, 135) if len(sys.argv) < 4 else tuple(int(x) for x in sys.argv[3].split(","))):
    ts = [run(D.tile_params(H, world, r, sh)) for r in range(world)]
    ms = [t for t, _ in ts]
    print("stripe_h %3d: rows %s  render ms min %.3f max %.3f mean %.3f" % (sh, sorted(set(r for _, r in ts)), min(ms), max(ms), sum(ms) / len(ms)), flush=True)

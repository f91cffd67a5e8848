import sys, time
sys.path.insert(0, "julia-spira_amd")
import torch
from spira_hip import distributed as D
H, W = 1080, 1920
for world in (2, 4, 8):
    mr = D.max_rows(H, world)
    sr, sl = D.source_of_rows(H, world)
    stacked = torch.randn((world, 3, mr, W), dtype=torch.float64, device="cuda")
    a = torch.as_tensor(sr, device="cuda"); b = torch.as_tensor(sl, device="cuda")
    for _ in range(3):
        img = stacked[a, :, b].permute(1, 0, 2).contiguous()
    torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10):
        img = stacked[a, :, b].permute(1, 0, 2).contiguous()
    e1.record(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(10):
        img = stacked[a, :, b].permute(1, 0, 2).contiguous()
    host = (time.perf_counter() - t0) / 10
    torch.cuda.synchronize()
    print("world %d: assemble %.3f ms on the device per frame, %.3f ms of host time to enqueue" % (world, e0.elapsed_time(e1) / 10, host * 1e3))

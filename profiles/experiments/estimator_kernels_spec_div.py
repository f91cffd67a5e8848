import os, sys, time
sys.path.insert(0, "julia-spira_amd")
import numpy as np, torch
from spira_hip import _binding as B, scenes
s = scenes.scene_s1()
for SEM, name in ((B.SEM_METAL | B.KERNEL_WAVEFRONT, "METAL wavefront"),):
  for prec, tdt in (("f32", torch.float32), ("f64", torch.float64)):
      for mode in (0, 1):
          os.environ["SPIRA_SPEC_DIV"] = str(mode)
          p = B.make_params(1920, 1080, 64, 8, 5, 5, 0, flags=SEM | B.POST_NONE, seed=3)
          out = torch.empty((3, 1080, 1920), dtype=tdt, device="cuda")
          sc = B.Scene(s["spheres5"], s["materials8"], None, prec)
          st = torch.cuda.current_stream().cuda_stream
          for _ in range(3): sc.render_device(s["camera12"], p, out.data_ptr(), 0, st)
          torch.cuda.synchronize(); t0 = time.perf_counter()
          for _ in range(6): sc.render_device(s["camera12"], p, out.data_ptr(), 0, st)
          torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 6
          c = B.counters()
          print(name + " %s SPEC_DIV=%d: %.3f ms/step  %.0f Msamples/s  segments %d redone %d" % (prec, mode, dt * 1e3, 1920 * 1080 * 64 / dt / 1e6, c["segments"], c["redone_waves"]), flush=True)

import sys, time
sys.path[:0]=["julia-spira_amd"]
import torch, numpy as np
from spira_hip import _binding as B, scenes
s=scenes.scene_s4()
for prec,tdt in (("f32",torch.float32),("f64",torch.float64)):
    out=torch.empty((3,1080,1920),dtype=tdt,device="cuda")
    for name,tri in (("with the 81 920-triangle mesh",s["triangles10"]),("mesh removed",None)):
        nt=0 if tri is None else len(tri)
        p=B.make_params(1920,1080,64,12,2,3,nt,flags=B.POST_NONE,seed=5)
        h=B.Scene(s["spheres5"],s["materials8"],tri,prec)
        for i in range(8):
            if i==3: torch.cuda.synchronize(); t=time.perf_counter()
            h.render_device(s["camera12"],p,out.data_ptr(),0,0)
        torch.cuda.synchronize(); dt=(time.perf_counter()-t)/5
        c=B.counters(); h.destroy()
        print(prec,name,"%.2f ms  %.0f Msamples/s  seg/sample %.3f  kernel %.2f ms"%(dt*1e3,1920*1080*64/dt/1e6,c["segments"]/c["samples"],c["bounce_kernel_ms"]))

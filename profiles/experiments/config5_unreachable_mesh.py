import sys, time
sys.path[:0]=["julia-spira_amd"]
import torch, numpy as np
from spira_hip import _binding as B, scenes
s=scenes.scene_s4()
def moved(dy, dz=0.0):
    t=s["triangles10"].copy(); t[:,[1,4,7]] += dy; t[:,[2,5,8]] += dz; return t
out=torch.empty((3,1080,1920),dtype=torch.float32,device="cuda")
for name,tri in (("mesh in place",s["triangles10"]),("mesh 30 up (out of reach, well conditioned)",moved(30.0)),("mesh behind the camera (z+10)",moved(0.0,10.0)),("mesh removed",None)):
    nt=0 if tri is None else len(tri)
    p=B.make_params(1920,1080,64,12,2,3,nt,flags=B.POST_NONE,seed=5)
    h=B.Scene(s["spheres5"],s["materials8"],tri,"f32")
    for i in range(8):
        if i==3: torch.cuda.synchronize(); t=time.perf_counter()
        h.render_device(s["camera12"],p,out.data_ptr(),0,0)
    torch.cuda.synchronize(); dt=(time.perf_counter()-t)/5
    c=B.counters(); h.destroy()
    print(name,"%.2f ms  seg/sample %.3f  kernel %.2f ms"%(dt*1e3,c["segments"]/c["samples"],c["bounce_kernel_ms"]))

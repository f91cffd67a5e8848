import sys, time
sys.path[:0]=["julia-spira_amd"]
import torch
from spira_hip import _binding as B, scenes
s=scenes.scene_s1()
for prec,tdt in (("f64",torch.float64),("f32",torch.float32)):
    out=torch.empty((3,1080,1920),dtype=tdt,device="cuda")
    p=B.make_params(1920,1080,64,8,5,5,0,flags=B.POST_NONE,seed=1)
    h=B.Scene(s["spheres5"],s["materials8"],None,prec)
    side=torch.cuda.Stream()
    nbytes = 1920*1080*64*3*(8 if prec=="f64" else 4)          # what one resolve reads
    src=torch.empty(nbytes//4,dtype=torch.float32,device="cuda").normal_()
    def run(n, with_side):
        torch.cuda.synchronize(); t=time.perf_counter()
        for i in range(n):
            h.render_device(s["camera12"],p,out.data_ptr(),0,torch.cuda.current_stream().cuda_stream)
            if with_side:
                with torch.cuda.stream(side):
                    src.sum()        # a memory-bound read of the same size as the resolve's, concurrent with the next render
        torch.cuda.synchronize(); return (time.perf_counter()-t)/n*1e3
    run(3,False)
    a=run(20,False); b=run(20,True)
    torch.cuda.synchronize(); t=time.perf_counter()
    for i in range(20): src.sum()
    torch.cuda.synchronize(); c=(time.perf_counter()-t)/20*1e3
    print(prec,"render alone %.3f ms | render + concurrent %.2f GB read on a 2nd stream %.3f ms | that read alone %.3f ms"%(a,nbytes/1e9,b,c))
    h.destroy()

import os, sys, time
ROOT="/root/repo"; sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT,"video-analysis_amd"))
import torch
from bench import synth_batch
from video.engine import FrameEngine
dev=torch.device("cuda",0); w,h,n=1920,1080,256
frames=synth_batch(torch,dev,w,h,n,40,0.002,seed=3)
labels=torch.empty((n,h,w),dtype=torch.int32,device=dev); counts=torch.zeros((n,),dtype=torch.int32,device=dev)
eng=FrameEngine(size=(w,h),max_batch=n,background="mean",sigma=5.0,thresh=20,morphology=(("dilate","rect",5),("erode","rect",5)),connectivity=4)
st=torch.cuda.current_stream(dev).cuda_stream
def run(k, prof):
    eng.profile(prof)
    torch.cuda.synchronize(); t0=time.perf_counter()
    for i in range(k): eng.run_device(frames.data_ptr(),n,None,None,labels.data_ptr(),counts.data_ptr(),None,st)
    torch.cuda.synchronize(); dt=time.perf_counter()-t0
    if prof: eng.stage_times()
    eng.profile(False)
    return k*n/dt
run(5,False)
for rep in range(4):
    print("profile off %.0f  on %.0f frames/s" % (run(30,False), run(30,True)))

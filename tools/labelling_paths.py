#!/usr/bin/env python3
"""Whole-chain time with the chip-wide labelling passes against the per-frame LDS kernel over frame
sizes and batch sizes: the measurements behind the cost model in ccl_frame_kernel_used (va_ccl.hip).
Run on an MI355X."""
import os, sys, time
ROOT=os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT,"video-analysis_amd"))
import numpy as np, torch
import bench
from video import _hip
from video.engine import FrameEngine
dev=torch.device("cuda:0")
for (w,h) in ((1920,1080),(640,480),(3840,2160)):
    for n in (16,32,64,96,128,256):
        if w==3840 and n>128: continue
        frames=bench.synth_batch(torch, dev, w, h, n, 40, 0.002, 3)
        labels=torch.empty((n,h,w),dtype=torch.int32,device=dev); counts=torch.empty((n,),dtype=torch.int32,device=dev)
        res=[]
        for path in (1,2):
            _hip.check(_hip.lib().va_test_hook_labelling(path,0))
            eng=FrameEngine(size=(w,h),max_batch=n,background="mean",sigma=5.0,thresh=20,morphology=(("dilate","rect",5),("erode","rect",5)),connectivity=4,device=0)
            st=torch.cuda.current_stream(dev).cuda_stream
            for _ in range(3): eng.run_device(frames.data_ptr(),n,None,None,labels.data_ptr(),counts.data_ptr(),None,st)
            torch.cuda.synchronize(); t0=time.perf_counter()
            for _ in range(10): eng.run_device(frames.data_ptr(),n,None,None,labels.data_ptr(),counts.data_ptr(),None,st)
            torch.cuda.synchronize(); res.append((time.perf_counter()-t0)/10*1e3); eng.close()
        _hip.check(_hip.lib().va_test_hook_labelling(0,0))
        print("%dx%d n=%3d  chip-wide %.3f ms  per-frame %.3f ms  -> %s" % (w,h,n,res[0],res[1],"frame" if res[1]<res[0] else "chip"))
        del frames, labels

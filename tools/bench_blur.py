#!/usr/bin/env python3
"""FilterBlur alone (va_gaussian_u8: u8 in, u8 out, 256 x 1080p resident in HBM) through its three
implementations: matrix-core kernel, dot4/dot2 kernel, generic two-pass.  Run on an MI355X."""
import os, sys, time, ctypes as C
import numpy as np
ROOT=os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT,"video-analysis_amd"))
import torch
from video import _hip
L=_hip.lib()
n,h,w=256,1080,1920
src=torch.randint(0,256,(n,h,w),dtype=torch.uint8,device="cuda")
dst=torch.empty_like(src)
st=torch.cuda.current_stream().cuda_stream
for name,fn in (("default",L.va_gaussian_u8),("valu",L.va_gaussian_u8_valu),("generic",L.va_gaussian_u8_generic)):
    for sigma in (5.0, 2.0, 8.0, 10.0):     # (radius <= 16: "default" = matrix cores; 17 ... 32: the wide dot4/dot2 kernel)
        for _ in range(2): _hip.check(fn(src.data_ptr(),dst.data_ptr(),n,h,w,1,C.c_double(sigma),st))
        torch.cuda.synchronize(); t0=time.perf_counter()
        for _ in range(5): _hip.check(fn(src.data_ptr(),dst.data_ptr(),n,h,w,1,C.c_double(sigma),st))
        torch.cuda.synchronize(); dt=(time.perf_counter()-t0)/5
        print("%-8s sigma=%.0f: %.3f ms per 256 frames -> %.0f frames/s, %.2f TB/s" % (name,sigma,dt*1e3,n/dt,2*n*h*w/dt/1e12))

import os, sys, ctypes as C
ROOT = "/root/repo"
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "video-analysis_amd"))
from video import _hip
_hip.LIB_PATH = os.path.join(ROOT, "video-analysis_amd/csrc/build/dbg/libva_dbg.so")
import torch
from bench import synth_batch
from video.engine import FrameEngine
dev = torch.device("cuda", 0)
w, h, n = 1920, 1080, 256
frames = synth_batch(torch, dev, w, h, n, 40, 0.002, seed=3)
labels = torch.empty((n, h, w), dtype=torch.int32, device=dev)
counts = torch.zeros((n,), dtype=torch.int32, device=dev)
eng = FrameEngine(size=(w, h), max_batch=n, background="mean", sigma=5.0, thresh=20,
                  morphology=(("dilate", "rect", 5), ("erode", "rect", 5)), connectivity=4)
st = torch.cuda.current_stream(dev).cuda_stream
L = _hip.lib()
buf = (C.c_longlong * 16)()
xb = (C.c_int * 32)()
for rep in range(12):
    eng.profile(True)
    eng.run_device(frames.data_ptr(), n, None, None, labels.data_ptr(), counts.data_ptr(), None, st)
    torch.cuda.synchronize()
    stt = eng.stage_times(); eng.profile(False)
    L.va_debug_ccl_xcc.argtypes = [C.c_void_p]; L.va_debug_ccl_xcc(xb)
    print("frame WGs 0..7 on XCC", list(xb[0:8]), " paint blocks 0..7 on XCC", list(xb[16:24]),
          " paint %.4f ms" % (stt["ccl_paint"][0] / stt["ccl_paint"][1]))
    L.va_debug_ccl_stamps.argtypes = [C.c_void_p]
    print(L.va_debug_ccl_stamps(buf), [ (buf[i+1]-buf[i])/100.0 for i in range(6)], "| link:", [buf[i]/100.0 for i in range(8,12)], int(counts.sum()))

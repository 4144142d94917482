#!/usr/bin/env python3
"""Probe: do two independent full chains on two HIP streams overlap (HBM-bound label painting of
one batch beside the VALU/latency-bound kernels of the other)?  Prints frames/s for one engine on
one stream and for two engines on two streams.  Run on an MI355X: python tools/two_stream_probe.py"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "video-analysis_amd"))
import torch
from bench import synth_batch
from video.engine import FrameEngine

dev = torch.device("cuda", 0)
w, h, n = 1920, 1080, 256
frames = [synth_batch(torch, dev, w, h, n, 40, 0.002, seed=3 + i) for i in range(2)]
labels = [torch.empty((n, h, w), dtype=torch.int32, device=dev) for _ in range(2)]
counts = [torch.zeros((n,), dtype=torch.int32, device=dev) for _ in range(2)]
engs = [FrameEngine(size=(w, h), max_batch=n, background="mean", sigma=5.0, thresh=20,
                    morphology=(("dilate", "rect", 5), ("erode", "rect", 5)), connectivity=4) for _ in range(2)]
streams = [torch.cuda.Stream(dev) for _ in range(2)]

def run(k, nstreams, steps):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for s in range(steps):
        for i in range(nstreams):
            engs[i].run_device(frames[i].data_ptr(), n, None, None, labels[i].data_ptr(), counts[i].data_ptr(), None,
                               streams[i].cuda_stream)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    return nstreams * steps * n / dt

run(0, 2, 3)
print("one stream : %.0f frames/s" % run(0, 1, 20))
print("two streams: %.0f frames/s" % run(0, 2, 20))
print("one stream : %.0f frames/s" % run(0, 1, 20))

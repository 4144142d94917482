#!/usr/bin/env python3
"""A/B on ONE box (boxes differ by several percent in HBM write rate): the headline chain with the run-table
hand-off between labelling and painting against the sparse-word convention (test hook path 3)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "video-analysis_amd"))
import torch
from bench import synth_batch
from video import _hip
from video.engine import FrameEngine
dev = torch.device("cuda", 0)
w, h, n = 1920, 1080, 256
frames = synth_batch(torch, dev, w, h, n, 40, 0.002, seed=3)
labels = torch.empty((n, h, w), dtype=torch.int32, device=dev)
counts = torch.zeros((n,), dtype=torch.int32, device=dev)
eng = FrameEngine(size=(w, h), max_batch=n, background="mean", sigma=5.0, thresh=20,
                  morphology=(("dilate", "rect", 5), ("erode", "rect", 5)), connectivity=4)
st = torch.cuda.current_stream(dev).cuda_stream
def run(steps=20):
    for _ in range(3):
        eng.run_device(frames.data_ptr(), n, None, None, labels.data_ptr(), counts.data_ptr(), None, st)
    torch.cuda.synchronize(); eng.profile(True); t0 = time.perf_counter()
    for _ in range(steps):
        eng.run_device(frames.data_ptr(), n, None, None, labels.data_ptr(), counts.data_ptr(), None, st)
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    stt = eng.stage_times(); eng.profile(False)
    return n * steps / dt, {k: round(v[0] / v[1], 4) for k, v in stt.items()}
for rep in range(2):
    for path, name in ((0, "run tables "), (3, "sparse words")):
        _hip.check(_hip.lib().va_test_hook_labelling(path, 0))
        fps, stages = run()
        print("%s: %.0f frames/s  %s" % (name, fps, stages))
_hip.check(_hip.lib().va_test_hook_labelling(0, 0))

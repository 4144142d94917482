#!/usr/bin/env python3
"""Device time of the stand-alone entry points of SURVEY.md 8(a) (the calls the Filter* classes make when a chain is
NOT contracted into the fused pipeline), HIP events around each call, 64 x 1080p resident in HBM.  One line per op:
ms per call, frames/s, algorithmic GB/s (inputs read once + outputs written once).  Run on an MI355X."""
import ctypes as C
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "video-analysis_amd"))
import torch
from video import _hip

L = _hip.lib()
dev = torch.device("cuda", 0)
st = torch.cuda.current_stream(dev)
S = st.cuda_stream
n, h, w = 64, 1080, 1920
px = h * w
g = torch.Generator(device=dev); g.manual_seed(11)
img = torch.randint(0, 256, (n, h, w), dtype=torch.uint8, device=dev, generator=g)
img2 = torch.randint(0, 256, (n, h, w), dtype=torch.uint8, device=dev, generator=g)
col = torch.randint(0, 256, (n, h, w, 3), dtype=torch.uint8, device=dev, generator=g)
yy = torch.arange(h, device=dev).view(1, h, 1); xx = torch.arange(w, device=dev).view(1, 1, w)
mask = torch.zeros((n, h, w), dtype=torch.uint8, device=dev)
for k in range(40):
    cx, cy, r = (97 * k * 7) % w, (61 * k * 5) % h, 10 + 2 * k
    mask |= (((xx - cx) ** 2 + (yy - cy) ** 2) <= r * r).to(torch.uint8)
mask *= 255
out8 = torch.empty_like(img)
colout = torch.empty_like(col)
out16 = torch.empty((n, h, w), dtype=torch.int16, device=dev)
labels = torch.empty((n, h, w), dtype=torch.int32, device=dev)
counts = torch.empty((n,), dtype=torch.int32, device=dev)
state = torch.zeros((h, w), dtype=torch.float64, device=dev)
m2 = torch.zeros((h, w), dtype=torch.float64, device=dev)
res = []


def timed(name, fn, alg_bytes, reps=5):
    fn(); torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record(st)
    for _ in range(reps):
        fn()
    b.record(st); b.synchronize()
    ms = a.elapsed_time(b) / reps
    row = {"op": name, "ms": round(ms, 4), "frames_per_s": round(n / ms * 1e3), "alg_GBs": round(alg_bytes / ms / 1e6, 1)}
    res.append(row)
    print(json.dumps(row), flush=True)


ck = _hip.check
timed("A1 va_gaussian_u8 sigma=3", lambda: ck(L.va_gaussian_u8(img.data_ptr(), out8.data_ptr(), n, h, w, 1, 3.0, S)), 2 * n * px)
timed("A1 va_gaussian_u8 sigma=3, 3 channels", lambda: ck(L.va_gaussian_u8(col.data_ptr(), colout.data_ptr(), n, h, w, 3, 3.0, S)), 6 * n * px)
timed("A2 va_bg_update running mean (+ difference)", lambda: ck(L.va_bg_update(1, 0, img.data_ptr(), out8.data_ptr(), state.data_ptr(), 0, 0.0, n, px, S)), 2 * n * px)
timed("A2 va_welford_u8", lambda: ck(L.va_welford_u8(img.data_ptr(), state.data_ptr(), m2.data_ptr(), 0, n, px, S)), n * px)
timed("A3 va_time_difference_u8", lambda: ck(L.va_time_difference_u8(img.data_ptr(), img2.data_ptr(), out16.data_ptr(), n * px, S)), 4 * n * px)
timed("A4 va_threshold_u8", lambda: ck(L.va_threshold_u8(img.data_ptr(), out8.data_ptr(), n * px, 100, 255, S)), 2 * n * px)
timed("A5 va_mono_mean_u8", lambda: ck(L.va_mono_mean_u8(col.data_ptr(), out8.data_ptr(), n * px, S)), 4 * n * px)
timed("A5 va_normalize_u8", lambda: ck(L.va_normalize_u8(img.data_ptr(), out8.data_ptr(), n * px, 20.0, 220.0, 255 / 200.0, 0.0, S)), 2 * n * px)
for nm, op, shape, ks in (("dilate rect 5", 1, 0, 5), ("erode cross 3", 0, 1, 3), ("dilate ellipse 9", 1, 2, 9)):
    timed("A6 va_morph_u8 %s" % nm, lambda: ck(L.va_morph_u8(mask.data_ptr(), out8.data_ptr(), n, h, w, op, shape, ks, S)), 2 * n * px)
wsb = L.va_label_workspace_bytes(n, h, w)
ws = torch.empty((wsb,), dtype=torch.uint8, device=dev)
timed("A7 va_label_i32 4-connectivity", lambda: ck(L.va_label_i32(mask.data_ptr(), labels.data_ptr(), counts.data_ptr(), n, h, w, 4, ws.data_ptr(), wsb, S)), 5 * n * px)
maxl = 64
stats = torch.empty((n, maxl, 16), dtype=torch.int64, device=dev)
timed("A9 va_moments_i64 (<= 64 labels)", lambda: ck(L.va_moments_i64(labels.data_ptr(), n, h, w, maxl, stats.data_ptr(), S)), 4 * n * px)
largest = torch.empty((n,), dtype=torch.int32, device=dev)
larea = torch.empty((n,), dtype=torch.int64, device=dev)
timed("A7 va_largest_region (+ mask)", lambda: ck(L.va_largest_region(labels.data_ptr(), counts.data_ptr(), stats.data_ptr(), n, h, w, maxl, largest.data_ptr(), larea.data_ptr(), out8.data_ptr(), S)), 5 * n * px)
rot = torch.empty((n, w, h), dtype=torch.uint8, device=dev)
timed("N4 va_rot90 k=1", lambda: ck(L.va_rot90(img.data_ptr(), rot.data_ptr(), n, h, w, 1, 1, S)), 2 * n * px)
print(json.dumps({"ops": res}))

#!/usr/bin/env python3
"""Device time of the next-tier entry points, HIP events around each call, inputs
resident in HBM, 1080p.  One line per op: ms per call, frames/s, algorithmic GB/s (every input byte
read once + every output byte written once).  Run on an MI355X:  python tools/bench_next_tier.py"""
import ctypes as C
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "video-analysis_amd"))
import numpy as np
import torch
from video import _hip

L = _hip.lib()
dev = torch.device("cuda", 0)
st = torch.cuda.current_stream(dev)
S = st.cuda_stream
n, h, w = 64, 1080, 1920
g = torch.Generator(device=dev); g.manual_seed(7)
img = torch.randint(0, 256, (n, h, w), dtype=torch.uint8, device=dev, generator=g)
yy = torch.arange(h, device=dev).view(1, h, 1); xx = torch.arange(w, device=dev).view(1, 1, w)
mask = torch.zeros((n, h, w), dtype=torch.uint8, device=dev)
for k in range(30):
    cx, cy, r = (97 * k * 7) % w, (61 * k * 5) % h, 15 + 3 * k
    mask |= (((xx - cx) ** 2 + (yy - cy) ** 2) <= r * r).to(torch.uint8)
mask *= 255
out_u8 = torch.empty_like(img)
res = []


def timed(name, fn, frames, alg_bytes, reps=5):
    fn(); torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record(st)
    for _ in range(reps):
        fn()
    b.record(st); b.synchronize()
    ms = a.elapsed_time(b) / reps
    row = {"op": name, "ms": round(ms, 4), "frames_per_s": round(frames / ms * 1e3), "alg_GBs": round(alg_bytes / ms / 1e6, 1)}
    res.append(row)
    print(json.dumps(row), flush=True)


px = h * w
# FilterResize (cv2.resize restatement), half size and 1.5x, every mode
for mode, name in ((0, "nearest"), (1, "linear"), (2, "cubic"), (3, "area")):
    for (dh, dw) in ((540, 960), (1620, 2880)):
        dst = torch.empty((n, dh, dw), dtype=torch.uint8, device=dev)
        timed("va_resize_u8 %s %dx%d->%dx%d" % (name, w, h, dw, dh),
              lambda: _hip.check(L.va_resize_u8(img.data_ptr(), dst.data_ptr(), n, h, w, 1, dh, dw, mode, S)),
              n, n * (px + dh * dw))
# get_image_statistics: local mean + variance, 11x11 box (float64 outputs)
mean = torch.empty((8, h, w), dtype=torch.float64, device=dev); var = torch.empty_like(mean)
timed("va_image_statistics_u8 box ksize=5 (8 frames)",
      lambda: _hip.check(L.va_image_statistics_u8(img.data_ptr(), mean.data_ptr(), var.data_ptr(), 8, h, w, 0, 5, 0.0, 0, S)),
      8, 8 * px * (1 + 16))
timed("va_detect_peaks_u8", lambda: _hip.check(L.va_detect_peaks_u8(img.data_ptr(), out_u8.data_ptr(), n, h, w, 1, S)),
      n, n * px * 2)
# contour of the largest region (8-connected labelling without paint + border following) and its moments
maxp = 8192
wsb = L.va_contour_workspace_bytes(n, h, w)
ws = torch.empty((wsb,), dtype=torch.uint8, device=dev)
pts = torch.empty((n, maxp, 2), dtype=torch.int32, device=dev)
npts = torch.empty((n,), dtype=torch.int32, device=dev)
area = torch.empty((n,), dtype=torch.float64, device=dev); ncomp = torch.empty((n,), dtype=torch.int32, device=dev)
timed("va_largest_contour", lambda: _hip.check(L.va_largest_contour(mask.data_ptr(), n, h, w, pts.data_ptr(), maxp, npts.data_ptr(),
                                                                      area.data_ptr(), ncomp.data_ptr(), ws.data_ptr(), wsb, S)),
      n, n * px)
mom = torch.empty((n, 10), dtype=torch.float64, device=dev)
timed("va_contour_moments (%d contours, <= %d points)" % (n, int(npts.max().item())),
      lambda: _hip.check(L.va_contour_moments(pts.data_ptr(), npts.data_ptr(), n, maxp, 0, mom.data_ptr(), S)),
      n, int(npts.sum().item()) * 8 + n * 80)
# mask_thinning (iterates until the mask is empty; one frame)
one = mask[0].clone(); scratch = torch.empty_like(one); skel = torch.empty_like(one); it = C.c_int()
def thin():
    one.copy_(mask[0])
    _hip.check(L.va_mask_thinning_u8(one.data_ptr(), scratch.data_ptr(), skel.data_ptr(), h, w, C.byref(it), S))
timed("va_mask_thinning_u8 (1 frame)", thin, 1, px * 3, reps=2)
res[-1]["iterations"] = it.value
# FilterBlur alone, both tap rules
for rule, nm in ((0, "cv4"), (1, "cv3")):
    timed("va_gaussian_u8_rule sigma=5 %s" % nm,
          lambda: _hip.check(L.va_gaussian_u8_rule(img.data_ptr(), out_u8.data_ptr(), n, h, w, 1, 5.0, rule, S)), n, n * px * 2)
# A5: crop -> monochrome (float64 mean of three channels) -> normalise, one pass (va_prepare_u8), 64 colour frames
col = torch.randint(0, 256, (n, h, w, 3), dtype=torch.uint8, device=dev, generator=g)
timed("va_prepare_u8 mono-mean + normalize 1920x1080x3 -> 1920x1080",
      lambda: _hip.check(L.va_prepare_u8(col.data_ptr(), out_u8.data_ptr(), n, h, w, 3, 0, 0, w, h, 3, 1, 20.0, 220.0,
                                         255 / 200.0, 0.0, S)), n, n * px * 4)
timed("va_prepare_u8 crop 1600x900 + channel pick",
      lambda: _hip.check(L.va_prepare_u8(col.data_ptr(), out_u8.data_ptr(), n, h, w, 3, 160, 90, 1600, 900, 1, 0, 0.0, 0.0,
                                         0.0, 0.0, S)), n, n * 1600 * 900 * 4)
timed("va_prepare_u8 crop 1599x900 + channel pick (rows of no whole dwords: one sample per thread)",
      lambda: _hip.check(L.va_prepare_u8(col.data_ptr(), out_u8.data_ptr(), n, h, w, 3, 160, 90, 1599, 900, 1, 0, 0.0, 0.0,
                                         0.0, 0.0, S)), n, n * 1599 * 900 * 4)
print(json.dumps({"next_tier": res}))

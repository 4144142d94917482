#!/usr/bin/env python3
"""Randomised bit-pattern sweep of the float32 Gaussian (packed kernels, older kernels for large
radii, generic path) against the CPU oracle; run on an MI355X: python tools/stress_f32.py"""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "video-analysis_amd"))
from oracle import oracle as O
from video import ops
O.build()
rng = np.random.default_rng(5)
t_end = time.time() + 150
case = 0
while time.time() < t_end:
    case += 1
    c = int(rng.choice([1, 3]))
    h = int(rng.integers(1, 90)); w = int(rng.integers(1, 400))
    if rng.random() < 0.1: w = int(rng.choice([1920, 1280, 640, 2000]))
    n = int(rng.integers(1, 4))
    sigma = float(rng.choice([0.5, 1.0, 2.0, 3.3, 5.0, 9.0, 11.0, 12.5, 15.0]))
    shape = (n, h, w, c) if c == 3 else (n, h, w)
    f = (rng.random(shape, dtype=np.float32) * 3 - 1).astype(np.float32)
    f.flat[::53] = 0.0; f.flat[7::131] = -0.0
    ref = O.gaussian_f32(f, sigma)
    got = ops.gaussian_blur(f, sigma, color=(c == 3))
    ok = np.array_equal(got.view(np.uint32), ref.view(np.uint32))
    if not ok:
        print("MISMATCH", shape, sigma, int((got.view(np.uint32) != ref.view(np.uint32)).sum())); sys.exit(1)
print("all %d f32 cases ok" % case)

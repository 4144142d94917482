#!/usr/bin/env python3
"""Randomised bit-pattern sweep aimed at the compile-time-radius float32 kernels (row_is_f32_kernel,
col_sym_f32_kernel): integer sigmas, frame heights around and above the column kernels' 120/128-row steps,
row lengths that are multiples of 4 samples (the fused path) and others (the generic path), plain blur and the
EMA pipeline with split batches -- all against the CPU oracle.  Run on an MI355X:
    python tools/stress_f32_kernels.py [seconds]"""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "video-analysis_amd"))
from oracle import oracle as O          # (checker only)
from video import ops
from video.engine import FrameEngine
O.build()
rng = np.random.default_rng(int(os.environ.get("SEED", "21")))
t_end = time.time() + (float(sys.argv[1]) if len(sys.argv) > 1 else 150.0)
case = 0
while time.time() < t_end:
    case += 1
    c = int(rng.choice([1, 3]))
    h = int(rng.choice([int(rng.integers(1, 60)), int(rng.integers(100, 140)), int(rng.integers(230, 270)),
                        int(rng.integers(300, 700))]))
    w = int(rng.choice([4 * int(rng.integers(10, 160)), int(rng.integers(40, 500)), 640, 1280, 1920]))
    sigma = float(rng.integers(1, 10))
    n = int(rng.integers(1, 4))
    shape = (n, h, w, c) if c == 3 else (n, h, w)
    if rng.random() < 0.5:
        f = (rng.random(shape, dtype=np.float32) * 3 - 1).astype(np.float32)
        f.flat[::53] = 0.0; f.flat[7::131] = -0.0
        ref = O.gaussian_f32(f, sigma)
        got = ops.gaussian_blur(f, sigma, color=(c == 3))
        ok, what = np.array_equal(got.view(np.uint32), ref.view(np.uint32)), "blur"
    else:
        n = int(rng.integers(1, 7)); shape = (n,) + shape[1:]
        rate = float(rng.choice([0.02, 0.3, 1.0]))
        clip = (rng.random(shape, dtype=np.float32) * 2 - 0.5).astype(np.float32)
        eng = FrameEngine(size=(w, h), channels=c, dtype=np.float32, max_batch=n, background="ema", bg_rate=rate, sigma=sigma)
        k = int(rng.integers(0, n + 1))
        parts = [eng.run(p, want=("filtered",))["filtered"] for p in (clip[:k], clip[k:]) if len(p)]
        got = np.concatenate(parts)
        state, seen = eng.get_background(); eng.close()
        diff, bg = O.bg_ema_f32(clip.reshape(n, -1), rate=np.float32(rate))
        ref = O.gaussian_f32(diff.reshape(shape), sigma)
        ok = np.array_equal(got.view(np.uint32), ref.view(np.uint32)) and seen == n and \
            np.array_equal(state.ravel().view(np.uint32), bg.ravel().view(np.uint32))
        what = "ema rate=%g split=%d" % (rate, k)
    if not ok:
        bad = np.argwhere(got.view(np.uint32) != ref.view(np.uint32))
        print("MISMATCH case %d" % case, what, shape, sigma, "differing samples:", len(bad), bad[:4].tolist(), flush=True)
        out = os.path.join(ROOT, "gpurun_out", "stress_f32_fail_%d.npz" % case)
        os.makedirs(os.path.dirname(out), exist_ok=True)
        np.savez_compressed(out, data=(f if what == "blur" else clip), got=got, ref=ref, sigma=sigma, what=what)
        nfail = globals().get("nfail", 0) + 1
        if nfail >= 3:
            sys.exit(1)
    if case % 50 == 0:
        print("%d cases ok" % case, flush=True)
print("all %d cases ok" % case)

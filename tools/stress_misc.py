#!/usr/bin/env python3
"""Randomised parity sweeps of the round-2 paths against the CPU oracle (run on an MI355X):
float32 EMA + blur through the pipeline (fused row kernel with the background state in registers,
marching column kernel; split batches), cv2.resize in every mode, contour moments of traced contours.
    python tools/stress_misc.py [seconds] [seed]"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "video-analysis_amd"))
from oracle import oracle as O          # noqa: E402  (checker only)
from video import ops                   # noqa: E402
from video.analysis import image, regions   # noqa: E402
from video.engine import FrameEngine    # noqa: E402


def case_f32_pipeline(rng):
    c = int(rng.choice([1, 3]))
    h = int(rng.integers(2, 150))
    w = int(rng.choice([int(rng.integers(8, 400)), 292, 640, 1280, 1920, 2000]))
    n = int(rng.integers(1, 9))
    sigma = float(rng.choice([1.0, 2.0, 3.3, 5.0, 7.5, 9.0]))
    rate = float(rng.choice([0.02, 0.3, 1.0]))
    shape = (n, h, w, c) if c == 3 else (n, h, w)
    clip = (rng.random(shape, dtype=np.float32) * 2 - 0.5).astype(np.float32)
    eng = FrameEngine(size=(w, h), channels=c, dtype=np.float32, max_batch=n, background="ema", bg_rate=rate, sigma=sigma)
    k = int(rng.integers(0, n + 1))
    got = np.concatenate([eng.run(clip[:k], want=("filtered",))["filtered"] if k else np.empty((0,) + shape[1:], np.float32),
                          eng.run(clip[k:], want=("filtered",))["filtered"] if k < n else np.empty((0,) + shape[1:], np.float32)])
    state, seen = eng.get_background()
    desc = eng.description
    eng.close()
    diff, bg = O.bg_ema_f32(clip.reshape(n, -1), rate=np.float32(rate))
    ref = O.gaussian_f32(diff.reshape(shape), sigma) if c == 1 else O.gaussian_f32(diff.reshape(shape), sigma)
    ok = np.array_equal(got.view(np.uint32), ref.view(np.uint32)) and seen == n and \
        np.array_equal(state.ravel().view(np.uint32), bg.ravel().view(np.uint32))
    return ok, "f32 pipeline %r sigma=%g rate=%g split=%d [%s]" % (shape, sigma, rate, k, desc)


def case_resize(rng):
    c = int(rng.choice([1, 3]))
    h, w = int(rng.integers(1, 200)), int(rng.integers(1, 300))
    dh, dw = int(rng.integers(1, 260)), int(rng.integers(1, 400))
    if rng.random() < 0.3:                       # integer factors
        f = int(rng.integers(2, 5))
        h, w = dh * f, dw * f
    mode = str(rng.choice(["nearest", "linear", "cubic", "area"]))
    img = rng.integers(0, 256, (2, h, w, c) if c == 3 else (2, h, w), dtype=np.uint8)
    got = ops.resize(img, (dw, dh), mode, color=c == 3)
    ref = O.resize_u8(img, (dw, dh), mode)
    return np.array_equal(got, ref), "resize %s %r -> %dx%d" % (mode, img.shape, dw, dh)


def case_contour_moments(rng):
    h, w = int(rng.integers(20, 300)), int(rng.integers(20, 500))
    yy, xx = np.mgrid[:h, :w]
    m = np.zeros((h, w), np.uint8)
    for _ in range(int(rng.integers(1, 6))):
        cx, cy, a, b = rng.uniform(0, w), rng.uniform(0, h), rng.uniform(2, w / 3), rng.uniform(2, h / 3)
        m[((xx - cx) / a) ** 2 + ((yy - cy) / b) ** 2 <= 1] = 1
    if not m.any():
        return True, "contour moments (empty)"
    contour = regions.get_contour_from_largest_region(m)
    ref_c = O.get_contour_from_largest_region(m)
    ok = np.array_equal(contour, ref_c)
    got = image.contour_moments(np.asarray(contour, np.int32).reshape(-1, 2))
    ref = O.contour_moments(np.asarray(ref_c, np.int32).reshape(-1, 2))
    ok = ok and all(np.float64(got[k]).tobytes() == np.float64(v).tobytes() for k, v in ref.items())
    return ok, "contour moments %dx%d, %d points" % (w, h, len(np.atleast_2d(contour)))


def main():
    budget = float(sys.argv[1]) if len(sys.argv) > 1 else 120.0
    rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
    O.build()
    t_end = time.time() + budget
    counts = {}
    cases = (case_f32_pipeline, case_resize, case_contour_moments)
    while time.time() < t_end:
        fn = cases[int(rng.integers(0, len(cases)))]
        ok, desc = fn(rng)
        counts[fn.__name__] = counts.get(fn.__name__, 0) + 1
        if not ok:
            print("MISMATCH:", desc, flush=True)
            sys.exit(1)
        if sum(counts.values()) % 50 == 0:
            print(sum(counts.values()), "cases ok; last:", desc, flush=True)
    print("all cases ok:", counts)


if __name__ == "__main__":
    main()

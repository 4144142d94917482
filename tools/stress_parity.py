#!/usr/bin/env python3
"""Randomised parity sweep of the fused chain against the CPU oracle (run on an MI355X):
    python tools/stress_parity.py [seconds] [seed]
Random frame sizes (incl. widths the matrix-core Gaussian / per-frame labelling kernels do and do
not take), sigmas, thresholds, morphology chains, connectivities and batch sizes (>= 96 frames so
that the per-frame labelling kernel runs). Prints one line per case, exits non-zero on a mismatch."""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "video-analysis_amd"))
from oracle import oracle as O          # noqa: E402  (checker only)
from video.engine import FrameEngine    # noqa: E402


def run_case(rng, case=0, allow_big=True):
    """one random configuration of the fused chain against the oracle; returns (ok, description)"""
    codes = {"erode": O.ERODE, "dilate": O.DILATE}
    big = allow_big and rng.random() < 0.15            # full-width frames now and then
    h = int(rng.integers(32, 200)) if not big else int(rng.integers(100, 420))
    w = int(rng.choice([64, 80, 96, 112, 128, 160, 208, 256, 272, 320, int(rng.integers(33, 300))]))
    if big:
        w = int(rng.choice([640, 1280, 1920, 1936, 2048, 2064, 2560, 3840, 4000]))
    n = int(rng.choice([1, 3, 7, 96, 100, 130])) if not big else int(rng.choice([2, 97]))
    sigma = float(rng.choice([0.6, 0.8, 1.0, 1.7, 2.0, 3.0, 4.2, 5.0, 5.3, 6.0, 7.5, 8.4, 10.0, 10.7, 11.0]))
    thresh = int(rng.integers(5, 60))
    conn = int(rng.choice([4, 8]))
    bg = str(rng.choice(["mean", "none"]))
    nm = int(rng.integers(0, 3))
    morph = tuple((str(rng.choice(["erode", "dilate"])), "rect", int(rng.choice([3, 5, 7]))) for _ in range(nm))
    base = rng.normal(100, 10, (h, w))
    clip = np.empty((n, h, w), np.uint8)
    yy, xx = np.mgrid[:h, :w]
    nb = int(rng.integers(1, 8))
    cx, cy = rng.uniform(0, w, nb), rng.uniform(0, h, nb)
    vx, vy = rng.uniform(-2, 2, nb), rng.uniform(-2, 2, nb)
    rad = rng.uniform(3, 25, nb)
    for t in range(n):
        f = base + rng.normal(0, 4, (h, w))
        for k in range(nb):
            f[(xx - cx[k] - vx[k] * t) ** 2 + (yy - cy[k] - vy[k] * t) ** 2 <= rad[k] ** 2] += 60
        f[rng.random((h, w)) < 0.003] = 255
        clip[t] = np.clip(f, 0, 255).astype(np.uint8)
    with_stats = rng.random() < 0.3              # per-label statistics come from the paint pass
    eng = FrameEngine(size=(w, h), max_batch=n, background=None if bg == "none" else bg, sigma=sigma,
                      thresh=thresh, morphology=morph, connectivity=conn,
                      max_labels=48 if with_stats else 0)
    want = ("filtered", "mask", "labels", "counts") + (("stats",) if with_stats else ())
    out = eng.run(clip, want=want)
    eng.close()
    cur = clip
    if bg == "mean":
        cur, _ = O.bg_mean_u8(clip)
    blur = O.gaussian_u8(cur, sigma)
    m = O.threshold_u8(blur, thresh)
    for op, _, k in morph:
        m = O.morph_u8(m, codes[op], O.RECT, k)
    rl, rc = O.label_batch(m, conn)
    ok = (np.array_equal(out["filtered"], blur) and np.array_equal(out["mask"], m)
          and np.array_equal(out["labels"], rl) and np.array_equal(out["counts"], rc))
    if ok and with_stats:
        for f in range(0, n, max(1, n // 5)):
            c = min(int(rc[f]), 48)
            ok = ok and np.array_equal(out["stats"][f, :c, :14], O.region_stats(rl[f], int(rc[f]))[:c, :14])
    if ok and case % 7 == 0 and m[0].any():       # contour of the largest region (8-conn roots + tracing)
        from video.analysis import regions
        ref, ref_area = O.get_contour_from_largest_region(m[0], ret_area=True)
        got, area = regions.get_contour_from_largest_region(m[0], ret_area=True)
        ok = np.array_equal(got, ref) and area == ref_area
    desc = ("case %d: n=%d %dx%d sigma=%.1f t=%d conn=%d bg=%s morph=%s"
            % (case, n, w, h, sigma, thresh, conn, bg, morph))
    if not ok:
        for nmk, a, b in (("filtered", out["filtered"], blur), ("mask", out["mask"], m), ("labels", out["labels"], rl)):
            desc += "\n    %s differs at %d elements" % (nmk, int((a != b).sum()))
    return ok, desc


def main():
    budget = float(sys.argv[1]) if len(sys.argv) > 1 else 120.0
    rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
    O.build()
    t_end = time.time() + budget
    case = 0
    while time.time() < t_end:
        case += 1
        ok, desc = run_case(rng, case)
        print(desc, "->", "ok" if ok else "MISMATCH", flush=True)
        if not ok:
            sys.exit(1)
    print("all %d cases ok" % case)


if __name__ == "__main__":
    main()

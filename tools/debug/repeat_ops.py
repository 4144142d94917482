"""Repeated identical calls through the NumPy layer (video.ops: pooled device buffers, pageable uploads / downloads):
every result must equal the first bit for bit.  On an MI355X:  python tools/debug/repeat_ops.py"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "video-analysis_amd"))
import numpy as np  # noqa: E402
from video import ops  # noqa: E402

rng = np.random.default_rng(99)
h, w = 1080, 1920
bad_runs = 0
for n, c, sigma, dtype in ((64, 1, 5.0, np.float32), (8, 1, 9.0, np.float32), (24, 3, 5.0, np.float32), (64, 1, 5.0, np.float32),
                           (128, 1, 5.0, np.uint8), (32, 3, 3.0, np.uint8)):
    shape = (n, h, w, 3) if c == 3 else (n, h, w)
    g = (rng.random(shape, dtype=np.float32) * 3 - 1).astype(np.float32) if dtype == np.float32 else rng.integers(0, 256, shape, dtype=np.uint8)
    first = ops.gaussian_blur(g, sigma, color=(c == 3))
    diffs = []
    for _ in range(8):
        o = ops.gaussian_blur(g, sigma, color=(c == 3))
        diffs.append(int((o.view(np.uint8) != first.view(np.uint8)).sum()))
    print(shape, np.dtype(dtype).name, sigma, "bytes differing from the first result:", diffs, flush=True)
    bad_runs += sum(1 for d in diffs if d)
print("runs with differences:", bad_runs)
sys.exit(1 if bad_runs else 0)

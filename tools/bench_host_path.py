#!/usr/bin/env python3
"""PCIe-inclusive throughput of the chain (NOT the headline `value`): frames start in pageable
host memory, are staged through pinned buffers and uploaded every batch, results come back to
the host.  Uses video.streaming.StreamedEngine (3 HIP streams, 3 slots).

    python tools/bench_host_path.py [--frames 512] [--batch 64] [--want counts,stats | labels]
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "video-analysis_amd")):
    sys.path.insert(0, p)
import numpy as np  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--frames", type=int, default=512)
    ap.add_argument("--batch", type=int, default=64)
    ap.add_argument("--want", default="counts,stats")
    ap.add_argument("--size", default="1920x1080")
    ap.add_argument("--zero-copy", action="store_true",
                    help="frames are produced in the pinned input buffers, results read in place")
    args = ap.parse_args()
    from video.engine import FrameEngine
    from video.streaming import StreamedEngine
    w, h = (int(v) for v in args.size.split("x"))
    rng = np.random.default_rng(3)
    base = np.clip(rng.normal(100, 10, (h, w)), 0, 255)
    clip = np.clip(base + rng.normal(0, 4, (args.batch, h, w)), 0, 255).astype(np.uint8)
    yy, xx = np.mgrid[:h, :w]
    for t in range(args.batch):
        for k in range(10):
            cx, cy, r = (97 * k + 11 * t) % w, (61 * k + 7 * t) % h, 10 + 5 * k
            clip[t][(xx - cx) ** 2 + (yy - cy) ** 2 <= r * r] = 200
    want = tuple(args.want.split(","))
    eng = FrameEngine(size=(w, h), max_batch=args.batch, background="mean", sigma=5.0, thresh=20,
                      morphology=(("dilate", "rect", 5), ("erode", "rect", 5)), connectivity=4,
                      max_labels=128)
    nb = args.frames // args.batch
    with StreamedEngine(eng, want=want, slots=3, copy_results=not args.zero_copy) as s:
        if args.zero_copy:            # a producer that writes into pinned memory: fill each slot once
            for _ in range(3):
                buf, _ = s.input_buffer()
                buf[:args.batch] = clip
                s.submit(None, n=args.batch)
        else:
            for _ in range(2):
                s.submit(clip)
        s.drain()
        t0 = time.perf_counter()
        got = 0
        for _ in range(nb):
            if args.zero_copy:
                _, done = s.input_buffer()
                got += len(done) + len(s.submit(None, n=args.batch))
            else:
                got += len(s.submit(clip))
        got += len(s.drain())
        dt = time.perf_counter() - t0
    assert got == nb
    fps = nb * args.batch / dt
    in_b = w * h
    out_b = sum({"counts": 4, "stats": 128 * 16 * 8, "labels": 4 * w * h, "mask": w * h,
                 "filtered": w * h}[k] for k in want)
    print(json.dumps({"metric": "frames/s incl. PCIe (host frames in, %s out)" % "+".join(want),
                      "value": round(fps, 1), "zero_copy": bool(args.zero_copy), "frames": nb * args.batch, "batch": args.batch,
                      "h2d_GBs": round(fps * in_b / 1e9, 2), "d2h_GBs": round(fps * out_b / 1e9, 2)}))
    eng.close()


if __name__ == "__main__":
    main()

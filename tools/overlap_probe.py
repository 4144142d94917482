#!/usr/bin/env python3
"""Probe: the store-bound label-image write of batch k beside the VALU/latency-bound stages of batch
k + 1 (va_pipeline_overlap), on cfg#3 (256 x 1080p).  Three modes of the same pipeline, same process,
alternating: serial / overlapped with one label buffer (the labelling stage waits for the previous
write) / overlapped with two alternating label buffers.  Checks that all modes produce identical
labels and counts, prints frames/s and the per-stage event times of each mode.
Run on an MI355X:  python tools/overlap_probe.py [steps]"""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "video-analysis_amd"))
import torch
from bench import synth_batch
from video.engine import FrameEngine

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 20
dev = torch.device("cuda", 0)
w, h, n = 1920, 1080, 256
clips = [synth_batch(torch, dev, w, h, n, 40, 0.002, seed=3 + i) for i in range(2)]     # two different batches
labels = [torch.empty((n, h, w), dtype=torch.int32, device=dev) for _ in range(2)]
counts = [torch.zeros((n,), dtype=torch.int32, device=dev) for _ in range(2)]
eng = FrameEngine(size=(w, h), max_batch=n, background="mean", sigma=5.0, thresh=20,
                  morphology=(("dilate", "rect", 5), ("erode", "rect", 5)), connectivity=4)
st = torch.cuda.current_stream(dev).cuda_stream


def run(mode, k, profile=False):
    """k steps; batch i comes from clips[i % 2]; returns (frames/s, stage times, final labels, counts)"""
    eng.overlap(mode != "serial")
    eng.set_background(None, 0)
    torch.cuda.synchronize()
    if profile:
        eng.profile(True)
    t0 = time.perf_counter()
    for i in range(k):
        j = i % 2 if mode == "overlap2" else 0
        eng.run_device(clips[i % 2].data_ptr(), n, None, None, labels[j].data_ptr(), counts[i % 2].data_ptr(), None, st)
    eng.fence(st)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    stage = eng.stage_times() if profile else {}
    if profile:
        eng.profile(False)
    last = labels[(k - 1) % 2 if mode == "overlap2" else 0]
    return k * n / dt, stage, last.clone(), counts[(k - 1) % 2].clone()


ref = None
for mode in ("serial", "overlap1", "overlap2"):
    run(mode, 3)
res = {}
for rep in range(3):
    for mode in ("serial", "overlap1", "overlap2"):
        fps, stage, lab, cnt = run(mode, steps, profile=True)
        if ref is None:
            ref = (lab, cnt)
        same = bool(torch.equal(lab, ref[0]) and torch.equal(cnt, ref[1]))
        res.setdefault(mode, []).append(round(fps))
        print(mode, "%.0f frames/s" % fps, "identical to serial:", same,
              {k: round(v[0] / max(v[1], 1), 4) for k, v in sorted(stage.items())}, flush=True)
        if not same:
            raise SystemExit("MISMATCH in mode " + mode)
print(json.dumps(res))

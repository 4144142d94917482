#!/usr/bin/env python3
"""bench.py -- frames/s of the per-frame hot path on MI355X (BASELINE.json metric).

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

A "step" is one pass of the full chain over one batch of synthetic frames that already sit in
HBM: FilterBackground(mean, float64) -> FilterBlur(sigma=5) -> FilterThreshold(20) ->
5x5 dilate + erode -> 4-connected labelling with object counts (BASELINE.json configs[2],
the configuration the metric "bg-sub+blur+thresh+CCL" is quoted on: 1920x1080 uint8,
batch 256).  Every rank owns one batch (frames shard across GPUs, no data-path collective);
the only exchange is the RCCL all-gather of the per-frame object counts at the end of each
step.  torch is used for device memory, streams and torch.distributed only; all compute goes
through the C ABI of libvideoanalysis_hip.so.

Prints ONE JSON line (rank 0).  `roofline` is for the dominant kernel of the chain, from HIP
events recorded on the launch stream inside the timed region; `cpu_baseline` is the oracle's
C restatement of the same chain timed on this box's host cores on a bounded sample.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
PKG = os.path.join(ROOT, "video-analysis_amd")
for _p in (ROOT, PKG):
    if _p not in sys.path:
        sys.path.insert(0, _p)

HBM_PEAK_GBS = 8000.0            # MI355X HBM3E spec peak (MI355X_MICROARCH.md)

WORKLOADS = {
    # name: (width, height, batch, sigma, thresh, morph ksize, blobs per frame, salt)
    "cfg3_1080p_full_chain": (1920, 1080, 256, 5.0, 20, 5, 40, 0.002),
    "cfg2_1080p_blur_thresh": (1920, 1080, 256, 5.0, 20, 0, 40, 0.0),
    "cfg4_4k_full_chain": (3840, 2160, 128, 5.0, 20, 5, 160, 0.002),
    "cfg1_480p_small": (640, 480, 64, 2.0, 20, 0, 3, 0.0),
}
# BASELINE.json configs[4]: float32 3-channel frames, adaptive (EMA) background + sigma=9 blur
F32_WORKLOADS = {"cfg5_1080p_f32x3_sigma9": (1920, 1080, 3, 256, 9.0, 0.02)}

# algorithmic (compulsory) bytes per frame-pixel of each stage = every input byte of the stage
# read once + every output byte written once (DESIGN.md "Kernels"); bits = 1/8 B per pixel
STAGE_BYTES_PER_PX = {
    "bg": 1 + 1,                 # u8 frame in, u8 difference out (f64 state amortised per batch)
    "gauss_fused": 1 + 1 / 8,    # u8 in, thresholded bit mask out
    "gauss_mfma": 1 + 1 / 8,     # the same contract on the matrix cores
    "gauss_generic": 1 + 1,      # u8 in, u8 out (the u16 scratch round trip is NOT compulsory)
    "gauss_planes": 1 + 1,       # u8 in, u8 out through reflected-padded planes
    "threshold_pack": 1 + 1 / 8,
    "morph_fused": 2 / 8,        # bit mask in, bit mask out (+ forest seeds, sparse)
    "morph_dilate": 2 / 8,
    "morph_erode": 2 / 8,
    "mask_unpack": 1 / 8 + 1,
    "ccl_init": 1 / 8,
    "ccl_link": 1 / 8,
    "ccl_flatten": 1 / 8,
    "ccl_rowscan": 0,
    "ccl_rank": 1 / 8,
    "ccl_frame": 1 / 8,          # per-frame labelling kernel: bit mask in, one sparse word per run out
    "ccl_paint": 1 / 8 + 4,      # bit mask in, int32 labels out
    "stats_init": 0,
}


def synth_batch(torch, device, w, h, n, blobs, salt, seed):
    """static Gaussian background + per-frame noise + moving discs (+ salt)"""
    g = torch.Generator(device=device)
    g.manual_seed(seed)
    bg = torch.normal(100.0, 10.0, (h, w), generator=g, device=device).clamp_(0, 255)
    frames = torch.empty((n, h, w), dtype=torch.uint8, device=device)
    yy = torch.arange(h, device=device, dtype=torch.float32).view(1, h, 1)
    xx = torch.arange(w, device=device, dtype=torch.float32).view(1, 1, w)
    pos = torch.rand((blobs, 2), generator=g, device=device) * torch.tensor([w, h], device=device)
    vel = (torch.rand((blobs, 2), generator=g, device=device) - 0.5) * 6
    rad = 8 + torch.rand((blobs,), generator=g, device=device) * 52
    t = torch.arange(n, device=device, dtype=torch.float32).view(n, 1, 1)
    chunk = 32
    for a in range(0, n, chunk):
        b = min(n, a + chunk)
        f = bg.unsqueeze(0) + torch.normal(0.0, 4.0, (b - a, h, w), generator=g, device=device)
        tt = t[a:b]
        for k in range(blobs):
            cx = pos[k, 0] + vel[k, 0] * tt
            cy = pos[k, 1] + vel[k, 1] * tt
            f += 60.0 * (((xx - cx) ** 2 + (yy - cy) ** 2) <= rad[k] ** 2)
        if salt > 0:
            f = torch.where(torch.rand((b - a, h, w), generator=g, device=device) < salt,
                            torch.full_like(f, 255.0), f)
        frames[a:b] = f.clamp_(0, 255).to(torch.uint8)
    return frames


def cpu_baseline(frames_host, sigma, thresh, morph, sample_frames):
    """the oracle's C restatement of the same chain, single thread, bounded sample"""
    from oracle import oracle as O
    O.build()
    sample = frames_host[:sample_frames]
    t0 = time.perf_counter()
    O.chain_u8(sample, sigma, thresh, morph_ksize=morph, connectivity=4, want_mask=False,
               want_labels=True)
    dt = time.perf_counter() - t0
    return {"value": round(len(sample) / dt, 3), "unit": "frames/s", "cores": 1, "kind": "port",
            "sample": "%d of the batch's %dx%d frames, full chain incl. labelling, oracle/va_oracle.c "
                      "gcc -O3 -march=native, 1 thread of %d host cores, %.1f s"
                      % (len(sample), frames_host.shape[2], frames_host.shape[1],
                         os.cpu_count() or 0, dt)}


def bench_f32(args, torch, device, dev_index, rank, world):
    """side workload (not the headline metric): cfg#5 float32 x 3 channels"""
    import numpy as np
    from video.engine import FrameEngine
    w, h, c, batch, sigma, rate = F32_WORKLOADS[args.workload]
    if args.batch > 0:
        batch = args.batch
    g = torch.Generator(device=device)
    g.manual_seed(5 + rank)
    frames = torch.rand((batch, h, w, c), generator=g, device=device, dtype=torch.float32)
    out = torch.empty_like(frames)
    eng = FrameEngine(size=(w, h), channels=c, dtype=np.float32, max_batch=batch, background="ema",
                      bg_rate=rate, sigma=sigma, device=dev_index)
    stream = torch.cuda.current_stream(device)
    for _ in range(args.warmup):
        eng.run_device(frames.data_ptr(), batch, out.data_ptr(), None, None, None, None, stream.cuda_stream)
    torch.cuda.synchronize(device)
    eng.profile(True)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        eng.run_device(frames.data_ptr(), batch, out.data_ptr(), None, None, None, None, stream.cuda_stream)
    torch.cuda.synchronize(device)
    dt = time.perf_counter() - t0
    stage = eng.stage_times()
    fps = batch * args.steps / dt
    alg = w * h * c * 4 * 2
    print(json.dumps({"metric": "frames/sec (EMA bg + sigma=9 blur) 1080p float32x3", "value": round(fps, 2),
                      "unit": "frames/s", "n_gpus": 1, "steps": args.steps, "warmup": args.warmup,
                      "ms_per_step": round(dt / args.steps * 1e3, 3), "dtype": "f32", "data": "synthetic",
                      "config": {"workload": args.workload, "engine": eng.description, "batch": batch},
                      "chain": {"alg_bytes_per_frame": alg, "achieved_GBs": round(alg * fps / 1e9, 1),
                                "stage_avg_ms": {k: round(v[0] / max(v[1], 1), 3) for k, v in stage.items()}}}),
          flush=True)
    eng.close()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--workload", default="cfg3_1080p_full_chain",
                    choices=sorted(WORKLOADS) + sorted(F32_WORKLOADS))
    ap.add_argument("--batch", type=int, default=0, help="override frames per step per GPU")
    ap.add_argument("--cpu-frames", type=int, default=160, help="frames in the CPU baseline sample")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"],
                    help="collective backend; gloo (CPU tensors) only to rehearse the multi-rank "
                         "control flow on a box with fewer GPUs than ranks")
    args = ap.parse_args()

    import numpy as np
    import torch
    import torch.distributed as dist

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus and world > 1:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d" % (args.gpus, world))
    if args.gpus > 1 and world == 1:
        raise SystemExit("launch with torch.distributed.run for --gpus > 1")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X (no CPU fallback)")
    dev_index = local_rank % torch.cuda.device_count() if args.backend == "gloo" else local_rank
    torch.cuda.set_device(dev_index)
    device = torch.device("cuda", dev_index)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if args.backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=device)
        else:
            dist.init_process_group("gloo", rank=rank, world_size=world)

    from video.engine import FrameEngine
    from video.sharding import gather_counts

    if args.workload in F32_WORKLOADS:
        return bench_f32(args, torch, device, dev_index, rank, world)
    w, h, batch, sigma, thresh, morph, blobs, salt = WORKLOADS[args.workload]
    if args.batch > 0:
        batch = args.batch
    frames = synth_batch(torch, device, w, h, batch, blobs, salt, seed=3 + rank)
    labels = torch.empty((batch, h, w), dtype=torch.int32, device=device)
    counts = torch.zeros((batch,), dtype=torch.int32, device=device)
    steps_morph = (("dilate", "rect", morph), ("erode", "rect", morph)) if morph else ()
    eng = FrameEngine(size=(w, h), max_batch=batch, background="mean", sigma=sigma, thresh=thresh,
                      morphology=steps_morph, connectivity=4, device=dev_index)
    stream = torch.cuda.current_stream(device)

    def step():
        eng.run_device(frames.data_ptr(), batch, None, None, labels.data_ptr(), counts.data_ptr(),
                       None, stream.cuda_stream)
        if world > 1:       # the path's only exchange: object counts of every shard, RCCL over xGMI
            return gather_counts(counts if args.backend == "nccl" else counts.cpu(), world * batch)
        return counts

    def fence():
        torch.cuda.synchronize(device)
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize(device)

    if world > 1:                 # communicator set-up is not a step: do it before any timing
        gather_counts(counts if args.backend == "nccl" else counts.cpu(), world * batch)
    fence()
    for _ in range(args.warmup):
        step()
    fence()
    eng.profile(True)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    fence()
    dt = time.perf_counter() - t0
    stage = eng.stage_times()
    eng.profile(False)
    if world > 1:
        tmax = torch.tensor([dt], dtype=torch.float64, device=device if args.backend == "nccl" else "cpu")
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dt = float(tmax.item())

    total_frames = world * batch * args.steps
    fps = total_frames / dt
    if rank == 0:
        px = w * h
        # dominant kernel by device time inside the timed region
        name, (ms_total, launches) = max(stage.items(), key=lambda kv: kv[1][0])
        avg_ms = ms_total / max(launches, 1)
        alg_bytes = STAGE_BYTES_PER_PX.get(name, 0) * px * batch
        achieved = alg_bytes / (avg_ms * 1e-3) / 1e9 if avg_ms > 0 else 0.0
        traffic = None
        tpath = os.path.join(ROOT, "profiles", "traffic.json")
        if os.path.exists(tpath):
            try:
                traffic = json.load(open(tpath)).get(args.workload, {}).get(name)
            except Exception:
                traffic = None
        chain_bytes_per_frame = px * 1 + px * 4 + 4          # SURVEY.md 8(d) cfg#3: 10 368 004 B @1080p
        stage_ms = {k: round(v[0] / max(v[1], 1), 4) for k, v in sorted(stage.items())}
        res = {
            "metric": "frames/sec (bg-sub+blur+thresh+CCL) 1080p uint8",
            "value": round(fps, 2),
            "unit": "frames/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": round(dt / args.steps * 1e3, 4),
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "u8",
            "data": "synthetic",
            "config": {"workload": "%s: %dx%d uint8, batch %d/GPU, running-mean bg-sub (f64 state) + "
                                   "sigma=%g Gaussian (q8.8) + threshold %d + %s + 4-connected "
                                   "labelling + object counts; inputs resident in HBM"
                                   % (args.workload, w, h, batch, sigma, thresh,
                                      "%dx%d dilate/erode" % (morph, morph) if morph else "no morphology"),
                       "frames_per_step_per_gpu": batch, "engine": eng.description},
            "roofline": {"bound": "hbm", "kernel": name, "achieved": round(achieved, 2),
                         "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": round(achieved / HBM_PEAK_GBS, 5), "traffic": traffic,
                         "avg_launch_ms": round(avg_ms, 4), "alg_bytes_per_launch": int(alg_bytes)},
            "chain": {"alg_bytes_per_frame": chain_bytes_per_frame,
                      "achieved_GBs": round(chain_bytes_per_frame * fps / world / 1e9, 2),
                      "frac_of_hbm_peak_per_gpu": round(chain_bytes_per_frame * fps / world / 1e9 / HBM_PEAK_GBS, 5),
                      "stage_avg_ms": stage_ms},
        }
        if world == 1 and not args.no_cpu_baseline:
            res["cpu_baseline"] = cpu_baseline(frames[:max(args.cpu_frames, 1)].cpu().numpy(), sigma,
                                               thresh, morph, args.cpu_frames)
        print(json.dumps(res), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    eng.close()


if __name__ == "__main__":
    main()

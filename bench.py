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
    # name: (width, height, batch, sigma, thresh, morph ksize (0: no morphology AND no labelling:
    #        the chain ends with the uint8 mask), blobs per frame, salt)
    "cfg3_1080p_full_chain": (1920, 1080, 256, 5.0, 20, 5, 40, 0.002),
    "cfg2_1080p_blur_thresh": (1920, 1080, 256, 5.0, 20, 0, 40, 0.0),
    "cfg4_4k_full_chain": (3840, 2160, 128, 5.0, 20, 5, 160, 0.002),
    "cfg1_480p_small": (640, 480, 64, 2.0, 20, 0, 3, 0.0),
}
# BASELINE.json configs[4]: float32 3-channel frames, adaptive (EMA) background + sigma=9 blur
F32_WORKLOADS = {"cfg5_1080p_f32x3_sigma9": (1920, 1080, 3, 256, 9.0, 0.02),
                 # probe, not a BASELINE config: mono frames with cfg#5's samples per row and per frame -- what
                 # the row pass costs without the channel interleave
                 "probe_5760x1080_f32x1_sigma9": (5760, 1080, 1, 256, 9.0, 0.02)}

# algorithmic (compulsory) bytes per frame-pixel of each stage = every input byte of the stage
# read once + every output byte written once (DESIGN.md "Kernels"); bits = 1/8 B per pixel
STAGE_BYTES_PER_PX = {
    "bg": 1 + 1,                 # u8 frame in, u8 difference out (f64 state amortised per batch)
    "gauss_fused": 1 + 1 / 8,    # u8 in, thresholded bit mask out
    "gauss_mfma": 1 + 1 / 8,     # the same contract on the matrix cores
    "gauss_mfma_mask8": 1 + 1,   # u8 in, thresholded uint8 mask out (the chain that ends at the mask)
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


# float32 side workload: bytes per float element of each stage
F32_STAGE_BYTES_PER_ELEM = {
    "bg": 4 + 4,                 # frame in, |frame - bg| out (state amortised over the batch)
    "gauss_f32": 4 + 4,          # difference in, blurred out (the row-pass intermediate is NOT compulsory)
    "gauss_generic": 4 + 4,
    "ema_row_f32": 4 + 4,        # fused EMA + row pass: frame in, row-filtered difference out
    "col_f32": 4 + 4,            # column pass: row-filtered in, blurred out
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


def _cpu_count():
    try:
        return len(os.sched_getaffinity(0))
    except AttributeError:
        return os.cpu_count() or 1


def _cpu_chain_frames(args):
    """worker of the all-core leg: blur -> threshold -> close -> label for frames [a, b) of the
    background-subtracted sample (memory-mapped), returns the object counts"""
    path, shape, a, b, sigma, thresh, morph = args
    import numpy as np
    from oracle import oracle as O
    diff = np.load(path, mmap_mode="r")
    counts = []
    for f in range(a, b):
        m = O.threshold_u8(O.gaussian_u8(np.ascontiguousarray(diff[f]), sigma), thresh)
        if morph:
            m = O.morph_u8(O.morph_u8(m, O.DILATE, O.RECT, morph), O.ERODE, O.RECT, morph)
        counts.append(O.label(m, 4)[1])
    return counts


def _cpu_bg_strip(args):
    """worker of the all-core leg: the running mean is sequential in time but pixel-local, so
    the background pass shards over image rows"""
    src, dst, shape, ya, yb = args
    import numpy as np
    from oracle import oracle as O
    frames = np.load(src, mmap_mode="r")
    out = np.lib.format.open_memmap(dst, mode="r+")
    d, _ = O.bg_mean_u8(np.ascontiguousarray(frames[:, ya:yb]))
    out[:, ya:yb] = d
    out.flush()
    return yb - ya


def cpu_baseline_main(argv):
    """`bench.py --cpu-baseline-child sample.npy sigma thresh morph`: runs in a FRESH process that
    never touches the GPU (so forking worker processes is safe) and prints one JSON object:
      value/cores=1 : the oracle's C chain (oracle/va_oracle.c), one thread -- kind "port"
      all_cores     : the same chain on every host core this process may use: background pass
                      sharded over image rows (temporal recurrence, pixel-local), then frames
                      sharded over processes
      literal_recipe: the reference's own NumPy/SciPy lines where they exist without cv2:
                      running mean (video/analysis/video.py:33) + ndimage.label + the per-label
                      np.sum loop (video/analysis/regions.py:162-169), a few frames
    """
    import multiprocessing as mp
    import numpy as np
    from oracle import oracle as O
    path, sigma, thresh, morph, n1 = argv[0], float(argv[1]), int(argv[2]), int(argv[3]), int(argv[4])
    check_path = argv[5] if len(argv) > 5 and argv[5] != "-" else None
    O.build()
    frames = np.load(path)
    n, h, w = frames.shape
    timing = n1 > 0                                   # (0: only the check below)
    n1 = max(1, min(n1, n))
    cores = _cpu_count()
    if check_path:
        # what the GPU produced for the first k frames of one more (untimed) step, and the background
        # state that step started from: the oracle must reproduce it bit for bit
        ck = np.load(check_path)
        k = int(ck["k"])
        cmask, clabels, ccounts, _ = O.chain_u8(frames[:k], sigma, thresh, morph_ksize=morph,
                                                connectivity=4 if morph else 0, mean=ck["state"],
                                                n_seen=int(ck["n_seen"]), want_mask=True, want_labels=bool(morph))
        chk = {"frames": k, "n_seen_before": int(ck["n_seen"])}
        if morph:
            chk["counts_equal"] = bool(np.array_equal(ccounts, ck["counts"]))
            chk["labels_equal"] = bool(np.array_equal(clabels, ck["labels"]))
            chk["objects"] = int(ccounts.sum())
        else:
            chk["mask_equal"] = bool(np.array_equal(cmask, ck["mask"]))
            chk["foreground_px"] = int((cmask != 0).sum())
        print("CPU_CHECK_JSON " + json.dumps(chk), flush=True)
    if not timing:
        print("CPU_BASELINE_JSON " + json.dumps({"value": None, "skipped": True}), flush=True)
        return
    # -- 1 thread, the first n1 frames
    t0 = time.perf_counter()
    mask, labels, counts, _ = O.chain_u8(frames[:n1], sigma, thresh, morph_ksize=morph, connectivity=4,
                                         want_mask=True, want_labels=True)
    dt1 = time.perf_counter() - t0
    res = {"value": round(n1 / dt1, 3), "unit": "frames/s", "cores": 1, "kind": "port",
           "sample": "%d of the batch's %dx%d frames, full chain incl. labelling, oracle/va_oracle.c "
                     "gcc -O3 -march=x86-64-v2, 1 thread of %d usable host cores (os.cpu_count()=%d), %.1f s"
                     % (n1, w, h, cores, os.cpu_count() or 0, dt1)}
    # -- all cores
    workers = max(1, min(cores, 64))
    diff_path = path + ".diff.npy"
    np.lib.format.open_memmap(diff_path, mode="w+", dtype=np.uint8, shape=frames.shape).flush()
    ctx = mp.get_context("fork")
    with ctx.Pool(workers) as pool:
        pool.map(abs, range(workers))                 # workers up before the clock starts
        t0 = time.perf_counter()
        rows = [(path, diff_path, frames.shape, h * k // workers, h * (k + 1) // workers)
                for k in range(workers) if h * k // workers < h * (k + 1) // workers]
        pool.map(_cpu_bg_strip, rows)
        shards = [(diff_path, frames.shape, n * k // workers, n * (k + 1) // workers, sigma, thresh, morph)
                  for k in range(workers) if n * k // workers < n * (k + 1) // workers]
        par_counts = [c for part in pool.map(_cpu_chain_frames, shards) for c in part]
        dtp = time.perf_counter() - t0
    if list(par_counts)[:n1] != [int(c) for c in counts]:
        raise SystemExit("cpu baseline: sharded chain disagrees with the sequential one")
    res["all_cores"] = {"value": round(n / dtp, 3), "unit": "frames/s", "cores": workers,
                        "sample": "the whole batch of %d frames: background pass sharded over image rows, then "
                                  "frames sharded over %d forked processes, %.1f s" % (n, workers, dtp)}
    os.unlink(diff_path)
    # -- the literal reference recipe, where it exists without cv2
    try:
        from scipy import ndimage
        k = min(4, n1)
        t0 = time.perf_counter()
        mean = np.zeros((h, w), np.double)
        for i in range(k):                                      # video/analysis/video.py:30-33
            mean = mean * i / (i + 1) + frames[i] / (i + 1)
        t_mean = time.perf_counter() - t0
        t0 = time.perf_counter()
        nlab = 0
        for i in range(k):                                      # video/analysis/regions.py:162-169
            lab, num = ndimage.label(mask[n1 - 1 - i])
            areas = [np.sum(lab == l) for l in range(1, num + 1)]
            nlab += num
            if areas:
                int(np.argmax(areas))
        t_lab = time.perf_counter() - t0
        res["literal_recipe"] = {
            "value": round(k / (t_mean + t_lab), 3), "unit": "frames/s", "cores": 1,
            "stages": "NumPy running mean (video/analysis/video.py:33) + scipy.ndimage.label + per-label "
                      "np.sum loop + argmax (video/analysis/regions.py:162-169); blur/threshold/morphology "
                      "are cv2 calls in the reference and cv2 is not installed, so they are NOT in this figure",
            "sample": "%d frames (%d labels): running mean %.3f s/frame, label + area loop %.3f s/frame"
                      % (k, nlab, t_mean / k, t_lab / k)}
    except ImportError:
        res["literal_recipe"] = None
    print("CPU_BASELINE_JSON " + json.dumps(res), flush=True)


def cpu_baseline_f32_main(argv):
    """f32 side workload: oracle EMA + sigma-tap float Gaussian on a few frames, 1 thread"""
    import numpy as np
    from oracle import oracle as O
    path, sigma, rate = argv[0], float(argv[1]), float(argv[2])
    O.build()
    frames = np.load(path)
    n = frames.shape[0]
    t0 = time.perf_counter()
    diff, _ = O.bg_ema_f32(frames, rate=rate)
    ref = O.gaussian_f32(diff, sigma)
    dt = time.perf_counter() - t0
    if len(argv) > 3 and argv[3] != "-":
        # the GPU's output for these very frames (a fresh pipeline, i.e. the same zero state): bit patterns
        ck = np.load(argv[3])
        k = int(ck["k"])
        same = bool(np.array_equal(ck["filtered"].view(np.uint32), ref[:k].view(np.uint32)))
        print("CPU_CHECK_JSON " + json.dumps({"frames": k, "filtered_equal": same,
                                              "max_abs": float(np.abs(ref[:k]).max())}), flush=True)
    res = {"value": round(n / dt, 3), "unit": "frames/s", "cores": 1, "kind": "port",
           "sample": "%d frames %r float32, EMA + sigma=%g Gaussian, oracle/va_oracle.c, 1 thread of %d "
                     "usable host cores, %.1f s" % (n, frames.shape[1:], sigma, _cpu_count(), dt)}
    print("CPU_BASELINE_JSON " + json.dumps(res), flush=True)


def cpu_baseline(sample, child_args, mode="--cpu-baseline-child", check=None):
    """times the CPU legs in a fresh child process (this one has initialised the GPU: no fork, no
    exec here -- a plain subprocess), on a bounded sample saved to /dev/shm (or the default temp
    directory when /dev/shm is too small).  `check`: arrays for the child's bit-for-bit comparison
    of one GPU step with the oracle.  Returns (cpu_baseline object, check object or None); a child
    that cannot run leaves an "error" entry instead of taking the GPU line down with it."""
    import subprocess
    import tempfile
    import numpy as np
    need = sample.nbytes * 2 + (sum(np.asarray(v).nbytes for v in check.values()) if check else 0) + (64 << 20)
    shm = None
    if os.path.isdir("/dev/shm"):
        try:
            st = os.statvfs("/dev/shm")
            if st.f_bavail * st.f_frsize > need:
                shm = "/dev/shm"
        except OSError:
            shm = None
    path = cpath = None
    try:
        fd, path = tempfile.mkstemp(suffix=".npy", prefix="va_bench_sample_", dir=shm)
        os.close(fd)
        np.save(path, sample)
        extra = []
        if check is not None:
            cpath = path + ".check.npz"
            np.savez(cpath, **check)
            extra = [cpath]
        out = subprocess.run([sys.executable, os.path.abspath(__file__), mode, path] +
                             [str(a) for a in child_args] + extra, stdout=subprocess.PIPE, stderr=subprocess.PIPE,
                             universal_newlines=True, timeout=900)
        res, chk = None, None
        for line in out.stdout.splitlines():
            if line.startswith("CPU_BASELINE_JSON "):
                res = json.loads(line[len("CPU_BASELINE_JSON "):])
            elif line.startswith("CPU_CHECK_JSON "):
                chk = json.loads(line[len("CPU_CHECK_JSON "):])
        if res is None:
            res = {"value": None, "error": (out.stderr or out.stdout)[-400:]}
        return res, chk
    except (OSError, subprocess.TimeoutExpired) as e:
        return {"value": None, "error": "%s: %s" % (type(e).__name__, str(e)[-300:])}, None
    finally:
        for q in (path, (path or "") + ".diff.npy", cpath):
            if q and os.path.exists(q):
                os.unlink(q)


def roofline_of(stage, px_bytes_table, units, workload, peak=HBM_PEAK_GBS):
    """`roofline` object for the dominant stage by device time (HIP events on the launch stream
    inside the timed region).  `units` = pixels (or float elements) one launch processes."""
    name, (ms_total, launches) = max(stage.items(), key=lambda kv: kv[1][0])
    avg_ms = ms_total / max(launches, 1)
    alg_bytes = px_bytes_table.get(name, 0) * units
    achieved = alg_bytes / (avg_ms * 1e-3) / 1e9 if avg_ms > 0 else 0.0
    traffic, source = None, None
    tpath = os.path.join(ROOT, "profiles", "traffic.json")
    if os.path.exists(tpath):
        try:
            table = json.load(open(tpath))
            stages = table.get(workload, {})
            traffic = stages.get(name, stages.get(name.replace("_mask8", "")))    # (one kernel, two epilogues)
            if traffic is not None:
                source = ("profiles/traffic.json: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of this "
                          "command (%s), not measured in this run" % table.get("_source", {}).get(workload, "r01_j"))
        except Exception:
            traffic = None
    return {"bound": "hbm", "kernel": name, "achieved": round(achieved, 2), "peak": peak, "unit": "GB/s",
            "frac": round(achieved / peak, 5), "traffic": traffic, "traffic_source": source,
            "avg_launch_ms": round(avg_ms, 4), "alg_bytes_per_launch": int(alg_bytes)}


def bench_f32(args, torch, device, dev_index, rank, world):
    """side workload (not the headline metric): BASELINE.json configs[4], float32 x 3 channels"""
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
    res = {"metric": "frames/sec (EMA bg + sigma=9 blur) 1080p float32x3", "value": round(fps, 2),
           "unit": "frames/s", "n_gpus": 1, "steps": args.steps, "warmup": args.warmup,
           "ms_per_step": round(dt / args.steps * 1e3, 3), "higher_is_better": True, "scaling": "weak",
           "vs_baseline": None, "dtype": "f32", "data": "synthetic",
           "config": {"workload": "%s: %dx%dx%d float32, batch %d, EMA background (rate %g) + sigma=%g "
                                  "Gaussian; inputs resident in HBM" % (args.workload, w, h, c, batch, rate, sigma),
                      "engine": eng.description, "batch": batch},
           "roofline": roofline_of(stage, F32_STAGE_BYTES_PER_ELEM, w * h * c * batch, args.workload),
           "chain": {"alg_bytes_per_frame": alg, "achieved_GBs": round(alg * fps / 1e9, 1),
                     "frac_of_hbm_peak": round(alg * fps / 1e9 / HBM_PEAK_GBS, 5),
                     "stage_avg_ms": {k: round(v[0] / max(v[1], 1), 3) for k, v in stage.items()}}}
    failed = False
    if not args.no_cpu_baseline or not args.no_check:
        # the oracle's verdict on the first frames of the workload, from a fresh pipeline (zero state):
        # outside the timed region, in the CPU child
        kf = max(1, min(args.cpu_frames_f32, batch))
        check = None
        if not args.no_check:
            e2 = FrameEngine(size=(w, h), channels=c, dtype=np.float32, max_batch=kf, background="ema",
                             bg_rate=rate, sigma=sigma, device=dev_index)
            o2 = torch.empty((kf, h, w, c), dtype=torch.float32, device=device)
            e2.run_device(frames.data_ptr(), kf, o2.data_ptr(), None, None, None, None, stream.cuda_stream)
            torch.cuda.synchronize(device)
            check = {"k": np.int64(kf), "filtered": o2.cpu().numpy()}
            e2.close()
        cpu, chk = cpu_baseline(frames[:kf].cpu().numpy(), (sigma, rate), "--cpu-baseline-f32-child", check=check)
        if not args.no_cpu_baseline:
            res["cpu_baseline"] = cpu
        if check is not None:
            res["check"] = chk if chk is not None else {"error": cpu.get("error", "the checker did not run")}
            failed = bool(chk) and not chk.get("filtered_equal")
    print(json.dumps(res), flush=True)
    eng.close()
    if failed:
        raise SystemExit("bench.py: the GPU step does not match the oracle (see \"check\")")


def launch_ranks(args, argv):
    """`python bench.py --gpus N` (N > 1, not yet under torchrun): start the N ranks as fresh child
    processes through torch.distributed.run BEFORE this process makes any GPU call, relay rank 0's
    JSON line, and exit with the launcher's code.  (Never re-exec a process that touched the GPU.)"""
    import socket
    import subprocess
    port = args.master_port
    if port <= 0:
        with socket.socket() as sk:
            sk.bind(("127.0.0.1", 0))
            port = sk.getsockname()[1]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("OMP_NUM_THREADS", "4")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus),
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + argv
    proc = subprocess.Popen(cmd, env=env, stdout=subprocess.PIPE, universal_newlines=True)
    line_out = None
    for line in proc.stdout:
        line = line.rstrip("\n")
        if line.startswith("{") and '"metric"' in line:
            line_out = line
        else:
            print(line, file=sys.stderr, flush=True)
    rc = proc.wait()
    if line_out is not None:
        print(line_out, flush=True)
    if rc != 0 or line_out is None:
        raise SystemExit(rc if rc != 0 else 1)
    raise SystemExit(0)


def main():
    if len(sys.argv) > 1 and sys.argv[1] == "--cpu-baseline-child":
        return cpu_baseline_main(sys.argv[2:])
    if len(sys.argv) > 1 and sys.argv[1] == "--cpu-baseline-f32-child":
        return cpu_baseline_f32_main(sys.argv[2:])
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--workload", default="cfg3_1080p_full_chain",
                    choices=sorted(WORKLOADS) + sorted(F32_WORKLOADS))
    ap.add_argument("--batch", type=int, default=0, help="override frames per step per GPU")
    ap.add_argument("--cpu-frames", type=int, default=160, help="frames in the CPU baseline sample")
    ap.add_argument("--cpu-frames-f32", type=int, default=6, help="frames in the f32 CPU baseline sample")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-check", action="store_true", help="skip the oracle's check of one GPU step")
    ap.add_argument("--check-frames", type=int, default=8, help="frames of one step the oracle recomputes")
    ap.add_argument("--no-extra", action="store_true", help="skip the gauss_valu and e2e_pinned_h2d legs")
    ap.add_argument("--gauss", default="mfma", choices=["mfma", "valu"],
                    help="8-bit Gaussian of the chain: matrix cores (default) or the dot4/dot2 VALU kernel")
    ap.add_argument("--video", default="own", choices=["own", "shared"],
                    help="own: every rank has its own clip (independent cameras; the default and the driver's form); "
                         "shared: ONE clip cut into contiguous shards (video/sharding.py), see --bg-share")
    ap.add_argument("--bg-share", default="static", choices=["static", "exact"],
                    help="--video shared: static = background measured on the first 8 frames by rank 0, broadcast once; "
                         "exact = cumulative running mean handed from shard to shard every step")
    ap.add_argument("--scaling", default="weak", choices=["weak", "strong"],
                    help="--video shared: weak = clip of N x batch frames, strong = clip of batch frames in all")
    ap.add_argument("--dump-counts", default="", help="rank 0 writes the gathered counts of the last step here (.npy)")
    ap.add_argument("--overlap", action="store_true",
                    help="label-image write of step k beside the stages of step k+1 (va_pipeline_overlap)")
    ap.add_argument("--master-port", type=int, default=0, help="rendezvous port for the self-launched ranks")
    ap.add_argument("--launcher", action="store_true",
                    help="go through the torch.distributed.run child launch even for --gpus 1")
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"],
                    help="collective backend; gloo (CPU tensors) only to rehearse the multi-rank "
                         "control flow on a box with fewer GPUs than ranks")
    args = ap.parse_args()

    under_torchrun = "RANK" in os.environ and "WORLD_SIZE" in os.environ
    if not under_torchrun and (args.gpus > 1 or args.launcher):
        return launch_ranks(args, [a for a in sys.argv[1:] if a != "--launcher"])

    import numpy as np
    import torch
    import torch.distributed as dist

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d" % (args.gpus, world))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X (no CPU fallback)")
    if args.backend == "nccl" and local_rank >= torch.cuda.device_count():
        raise SystemExit("rank %d: LOCAL_RANK %d but only %d GPUs visible"
                         % (rank, local_rank, torch.cuda.device_count()))
    dev_index = local_rank % torch.cuda.device_count() if args.backend == "gloo" else local_rank
    torch.cuda.set_device(dev_index)
    device = torch.device("cuda", dev_index)
    if world > 1 or under_torchrun:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if args.backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=device)
        else:
            dist.init_process_group("gloo", rank=rank, world_size=world)
        if dist.get_world_size() != args.gpus:
            raise SystemExit("process group has %d ranks, --gpus %d" % (dist.get_world_size(), args.gpus))
    distributed = dist.is_initialized()

    from video.engine import FrameEngine
    from video.sharding import CountGather, gather_counts

    if args.workload in F32_WORKLOADS:
        if world > 1:
            raise SystemExit("the float32 side workload is a single-GPU bench")
        rc = bench_f32(args, torch, device, dev_index, rank, world)
        if distributed:
            dist.destroy_process_group()
        return rc
    w, h, batch, sigma, thresh, morph, blobs, salt = WORKLOADS[args.workload]
    if args.batch > 0:
        batch = args.batch
    from video.sharding import background_handoff, broadcast_background, shard_range
    shared = args.video == "shared"
    if shared:
        # one clip for all ranks (same seed everywhere), cut into contiguous shards
        total = batch * (world if args.scaling == "weak" else 1)
        full = synth_batch(torch, device, w, h, total, blobs, salt, seed=3)
        fa, fb = shard_range(total, world, rank)
        frames = full[fa:fb].contiguous()
        first8 = full[:8].contiguous()
        del full
        batch = fb - fa
    else:
        total = batch * world
        frames = synth_batch(torch, device, w, h, batch, blobs, salt, seed=3 + rank)
    ccl = morph > 0               # BASELINE configs[1] / [0] stop at the thresholded mask
    overlap = bool(args.overlap) and ccl
    # overlapped runs alternate two label buffers (the consumer of batch k reads one while batch
    # k + 1 is painted into the other)
    labels = [torch.empty((batch, h, w), dtype=torch.int32, device=device) for _ in range(2 if overlap else 1)] if ccl else None
    mask = None if ccl else torch.empty((batch, h, w), dtype=torch.uint8, device=device)
    # two count buffers: the gather of step k (asynchronous, finished one step later) reads one while the
    # labelling of step k + 1 writes the other
    counts2 = [torch.zeros((batch,), dtype=torch.int32, device=device) for _ in range(2)]
    counts = counts2[0]
    steps_morph = (("dilate", "rect", morph), ("erode", "rect", morph)) if morph else ()
    from video import _hip
    L = _hip.lib(dev_index)
    if args.gauss == "valu":      # the north star's "no MFMA" form of the chain (same bits)
        _hip.check(L.va_test_hook_gaussian_u8(1))
    eng_kw = dict(size=(w, h), max_batch=max(batch, 1), background="static" if shared and args.bg_share == "static" else "mean",
                  sigma=sigma, thresh=thresh, morphology=steps_morph, connectivity=4 if ccl else 0, device=dev_index)
    eng = FrameEngine(**eng_kw)
    _hip.check(L.va_test_hook_gaussian_u8(0))
    wire = device if args.backend == "nccl" else None

    def advance_state(state, seen, clip):
        """running-mean state after `clip`, the state update alone (va_bg_update, diff_out = NULL)"""
        st = torch.from_numpy(np.ascontiguousarray(state, np.float64)).to(device)
        _hip.check(L.va_bg_update(1, 0, clip.data_ptr(), None, st.data_ptr(), int(seen), 0.0, clip.shape[0], w * h,
                                  torch.cuda.current_stream(device).cuda_stream))
        return st.cpu().numpy()

    if shared and args.bg_share == "static":
        state = advance_state(np.zeros((h, w)), 0, first8) if rank == 0 else np.zeros((h, w))
        eng.set_background(broadcast_background(state, src=0, device=wire), 0)
    if overlap:
        eng.overlap(True)
    stream = torch.cuda.current_stream(device)

    def run_chain(e, i=0):
        if ccl:
            e.run_device(frames.data_ptr(), batch, None, None, labels[i % len(labels)].data_ptr(),
                         counts2[i % 2].data_ptr(), None, stream.cuda_stream)
        else:
            e.run_device(frames.data_ptr(), batch, None, mask.data_ptr(), None, None, None, stream.cuda_stream)

    def step(i=0):
        if shared and args.bg_share == "exact":
            # exact cumulative mean over the whole clip, every pass: rank r starts from the state after
            # frames [0, start_r), handed on by rank r - 1 (which advanced it with the state update alone)
            st0, seen0 = background_handoff(lambda st, sn: advance_state(st, sn, frames), (h, w), np.float64,
                                            n_local=batch, device=wire)
            eng.set_background(st0, seen0)
        run_chain(eng, i)
        if world > 1:
            # the path's only exchange: object counts of every shard, RCCL over xGMI.  Step i's all-gather is
            # enqueued now and collected after step i + 1 has been launched, so the compute stream never waits
            # for the collective (every step's gather still completes inside the timed region: fence())
            done = gather.finish()
            c = counts2[i % 2] if ccl else counts
            gather.start(c if args.backend == "nccl" else c.cpu())
            return done
        return counts2[i % 2] if ccl else counts

    gather = CountGather(total) if world > 1 else None
    last_counts = [None]

    def fence():
        if gather is not None:
            got = gather.finish()
            if got is not None:
                last_counts[0] = got
        if overlap:
            eng.fence(stream.cuda_stream)      # the label-image writes on the pipeline's own stream
        torch.cuda.synchronize(device)
        if distributed:
            dist.barrier()
        torch.cuda.synchronize(device)

    if world > 1:                 # communicator set-up is not a step: do it before any timing
        gather_counts(counts if args.backend == "nccl" else counts.cpu(), total)
    fence()
    for i in range(args.warmup):
        step(i)
    fence()
    # per-stage HIP events inside the timed region, on every 4th step (an event per stage on every step costs
    # about 2 % of the chain: tools/debug/profile_overhead.py); fewer than 8 steps: every step
    sample = 4 if args.steps >= 8 else 1
    eng.profile(True, every=sample)
    marks = [torch.cuda.Event(enable_timing=True) for _ in range(args.steps + 1)]
    t0 = time.perf_counter()
    marks[0].record(stream)
    for i in range(args.steps):
        all_counts = step(i)
        marks[i + 1].record(stream)           # per-step device time on the launch stream (median below)
    fence()
    dt = time.perf_counter() - t0
    stage = eng.stage_times()
    eng.profile(False)
    step_ms = sorted(marks[i].elapsed_time(marks[i + 1]) for i in range(args.steps))
    median_ms = step_ms[len(step_ms) // 2] if len(step_ms) % 2 else 0.5 * (step_ms[len(step_ms) // 2 - 1] + step_ms[len(step_ms) // 2])
    if distributed:
        tmax = torch.tensor([dt], dtype=torch.float64, device=device if args.backend == "nccl" else "cpu")
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dt = float(tmax.item())
    if world > 1:
        all_counts = last_counts[0]                    # (the last step's gather, collected by fence())
        if all_counts is None or all_counts.numel() != total:
            raise SystemExit("count gather returned %s entries for %d frames"
                             % (None if all_counts is None else all_counts.numel(), total))

    total_frames = total * args.steps
    fps = total_frames / dt
    if args.dump_counts and rank == 0:
        np.save(args.dump_counts, all_counts.cpu().numpy())
    failed = False
    if rank == 0:
        px = w * h
        # frame in + int32 labels out + count (10 368 004 B @1080p), or + u8 mask out
        chain_bytes_per_frame = px * 1 + px * 4 + 4 if ccl else px * 2
        stage_ms = {k: round(v[0] / max(v[1], 1), 4) for k, v in sorted(stage.items())}
        res = {
            "metric": ("frames/sec (bg-sub+blur+thresh+CCL) %s uint8" % ("1080p" if (w, h) == (1920, 1080) else "%dx%d" % (w, h)))
                      if ccl else "frames/sec (bg-sub+blur+thresh) %dx%d uint8" % (w, h),
            "value": round(fps, 2),
            "unit": "frames/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": round(dt / args.steps * 1e3, 4),
            "median_ms_per_step": round(median_ms, 4),
            "higher_is_better": True,
            "scaling": args.scaling if shared else "weak",
            "vs_baseline": None,
            "dtype": "u8",
            "data": "synthetic",
            "config": {"workload": "%s: %dx%d uint8, batch %d/GPU, running-mean bg-sub (f64 state) + "
                                   "sigma=%g Gaussian (q8.8) + threshold %d + %s; inputs resident in HBM"
                                   % (args.workload, w, h, batch, sigma, thresh,
                                      "%dx%d dilate/erode + 4-connected labelling + object counts" % (morph, morph)
                                      if morph else "uint8 mask out (no morphology, no labelling)"),
                       "frames_per_step_per_gpu": batch, "engine": eng.description,
                       "video": ("one clip of %d frames in contiguous shards, background %s" % (total, {
                           "static": "measured on its first 8 frames by rank 0 and broadcast once",
                           "exact": "= cumulative running mean handed from shard to shard every step"}[args.bg_share]))
                                if shared else "one clip per rank",
                       "overlap": "label-image write of step k on the pipeline's own stream beside the stages of "
                                  "step k+1 (two label buffers); stage times then overlap" if overlap else "off",
                       "world_size": dist.get_world_size() if distributed else 1,
                       "backend": (args.backend + (" (RCCL)" if args.backend == "nccl" else "")) if distributed
                                  else "single process"},
            "roofline": roofline_of(stage, STAGE_BYTES_PER_PX, px * batch, args.workload),
            "chain": {"alg_bytes_per_frame": chain_bytes_per_frame,
                      "achieved_GBs": round(chain_bytes_per_frame * fps / world / 1e9, 2),
                      "frac_of_hbm_peak_per_gpu": round(chain_bytes_per_frame * fps / world / 1e9 / HBM_PEAK_GBS, 5),
                      "stage_avg_ms": stage_ms,
                      "stage_events": "HIP events on the launch stream, every %s step of the timed region (%d launches)"
                                      % ("4th" if sample == 4 else "", max(v[1] for v in stage.values()) if stage else 0)},
        }
        if world == 1:
            # ---- everything below is outside the timed region ------------------------------------
            # (1) one more step of the same call, from a known background state, for the oracle
            check = None
            if not args.no_check and not shared:
                k = max(1, min(args.check_frames, batch))
                state0, seen0 = eng.get_background()
                run_chain(eng, args.steps)
                fence()
                check = {"k": np.int64(k), "state": state0, "n_seen": np.int64(seen0)}
                if ccl:
                    check["labels"] = labels[args.steps % len(labels)][:k].cpu().numpy()
                    check["counts"] = counts2[args.steps % 2][:k].cpu().numpy()
                else:
                    check["mask"] = mask[:k].cpu().numpy()
            # (2) the device's own fill / copy rates on the chain's largest buffer (second denominator)
            # (the side legs must never cost the line its headline: a failure is reported in place)
            def leg(name, fn):
                try:
                    res[name] = fn()
                except Exception as e:                       # noqa: BLE001
                    res[name] = {"error": "%s: %s" % (type(e).__name__, str(e)[:300])}
            leg("copy_ceiling_GBs", lambda: copy_ceiling(torch, L, _hip, stream, labels[0] if ccl else mask, frames))
            if "fill" in res["copy_ceiling_GBs"]:
                res["roofline"]["frac_of_copy_ceiling"] = round(
                    res["roofline"]["achieved"] / max(res["copy_ceiling_GBs"]["fill"], 1e-9), 5)
            # (3) the same chain with its Gaussian on the VALU (the north star's "no MFMA"), same bits
            if args.gauss == "mfma" and "mfma" in eng.description and not args.no_extra and not shared:
                leg("gauss_valu", lambda: side_chain(torch, device, L, _hip, eng_kw, run_chain, fence, batch, valu=True))
            # (4) PCIe-inclusive: frames in pinned host memory, counts back (never `value`)
            if not args.no_extra and not shared:
                leg("e2e_pinned_h2d", lambda: e2e_leg(torch, frames, eng_kw, ccl))
            # (5) CPU legs + the oracle's verdict on the step of (1), in a fresh child process
            if not args.no_cpu_baseline or check is not None:
                cpu, chk = cpu_baseline(frames.cpu().numpy(),
                                        (sigma, thresh, morph, 0 if args.no_cpu_baseline else max(args.cpu_frames, 1)),
                                        check=check)
                if not args.no_cpu_baseline:
                    res["cpu_baseline"] = cpu
                if check is not None:
                    res["check"] = chk if chk is not None else {"error": cpu.get("error", "the checker did not run")}
                    # a MISMATCH fails the run; a checker that could not run is reported, not counted as one
                    failed = bool(chk) and not all(v for kk, v in chk.items() if kk.endswith("_equal"))
        print(json.dumps(res), flush=True)
    if distributed:
        dist.barrier()
        dist.destroy_process_group()
    eng.close()
    if failed:
        raise SystemExit("bench.py: the GPU step does not match the oracle (see \"check\")")


def copy_ceiling(torch, L, _hip, stream, big, frames):
    """plain device fill of, and copy into, the chain's largest output buffer: what the memory
    system gives the simplest possible kernels on this box, GB/s (bytes written; the copy also
    reads as many)"""
    nbytes = big.numel() * big.element_size()
    src = torch.empty_like(big)
    out = {}
    for name, fn in (("fill", lambda: L.va_memset(big.data_ptr(), 0, nbytes, stream.cuda_stream)),
                     ("copy", lambda: L.va_memcpy_d2d(big.data_ptr(), src.data_ptr(), nbytes, stream.cuda_stream))):
        best = None
        for _ in range(4):
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record(stream)
            _hip.check(fn())
            b.record(stream)
            b.synchronize()
            ms = a.elapsed_time(b)
            best = ms if best is None else min(best, ms)
        out[name] = round(nbytes / (best * 1e-3) / 1e9, 1)
    out["bytes"] = nbytes
    out["note"] = "hipMemsetAsync / hipMemcpyAsync D2D on the label (mask) buffer, best of 4; copy moves 2x these bytes"
    return out


def side_chain(torch, device, L, _hip, eng_kw, run_chain, fence, batch, valu):
    """the same workload through a second pipeline whose 8-bit Gaussian runs on the VALU"""
    from video.engine import FrameEngine
    _hip.check(L.va_test_hook_gaussian_u8(1 if valu else 0))
    e2 = FrameEngine(**eng_kw)
    _hip.check(L.va_test_hook_gaussian_u8(0))
    for i in range(2):
        run_chain(e2, i)
    fence()
    e2.profile(True)
    t0 = time.perf_counter()
    k = 5
    for i in range(k):
        run_chain(e2, i)
    fence()
    dt = time.perf_counter() - t0
    st = e2.stage_times()
    desc = e2.description
    e2.close()
    g = [v for kk, v in st.items() if kk.startswith("gauss")]
    return {"value": round(batch * k / dt, 1), "unit": "frames/s", "steps": k, "engine": desc,
            "gauss_ms": round(g[0][0] / max(g[0][1], 1), 4) if g else None}


def e2e_leg(torch, frames, eng_kw, ccl):
    """host frames (pinned, produced in place) -> H2D -> chain -> per-frame counts (or the mask) D2H,
    three HIP streams, batches of 64: the PCIe-inclusive rate of the same workload"""
    from video.engine import FrameEngine
    from video.streaming import StreamedEngine
    sub = 64
    kw = dict(eng_kw)
    kw["max_batch"] = sub
    host = frames[:sub].cpu().numpy()
    e = FrameEngine(**kw)
    want = ("counts",) if ccl else ("mask",)
    nb = 12
    with StreamedEngine(e, want=want, slots=3, copy_results=False) as s:
        for _ in range(3):
            buf, _d = s.input_buffer()
            buf[:sub] = host
            s.submit(None, n=sub)
        s.drain()
        t0 = time.perf_counter()
        got = 0
        for _ in range(nb):
            _b, done = s.input_buffer()
            got += len(done) + len(s.submit(None, n=sub))
        got += len(s.drain())
        dt = time.perf_counter() - t0
    e.close()
    fps = nb * sub / dt
    return {"value": round(fps, 1), "unit": "frames/s", "frames": nb * sub, "batch": sub, "out": "+".join(want),
            "h2d_GBs": round(fps * host[0].nbytes / 1e9, 2),
            "note": "video/streaming.py StreamedEngine, frames already in pinned host buffers; never `value`"}


if __name__ == "__main__":
    main()

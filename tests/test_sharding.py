"""CPU: the multi-GPU path's host logic -- frame partition and the count gather -- with a real
2-process torch.distributed group on the gloo backend (the GPU run uses "nccl" = RCCL)."""
import os
import socket
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_shard_ranges_partition_the_video():
    from video.sharding import shard_range, shard_sizes
    for n in (0, 1, 7, 8, 256, 1024, 1031):
        for world in (1, 2, 3, 4, 8):
            ranges = [shard_range(n, world, r) for r in range(world)]
            assert ranges[0][0] == 0 and ranges[-1][1] == n
            assert all(ranges[i][1] == ranges[i + 1][0] for i in range(world - 1))   # contiguous
            sizes = shard_sizes(n, world)
            assert sum(sizes) == n and max(sizes) - min(sizes) <= 1
    assert shard_range(1024, 8, 3) == (384, 512)            # BASELINE cfg#4: 128 frames per GPU
    with pytest.raises(ValueError):
        shard_range(10, 4, 4)


def _worker(rank, world, port, n_frames, out_dir):
    import torch
    import torch.distributed as dist
    sys.path.insert(0, os.path.join(ROOT, "video-analysis_amd"))
    from video.sharding import gather_counts, shard_range
    dist.init_process_group("gloo", init_method="tcp://127.0.0.1:%d" % port, rank=rank,
                            world_size=world)
    a, b = shard_range(n_frames, world, rank)
    full = (np.arange(n_frames) * 7 + 3) % 1000            # the "true" per-frame counts
    local = torch.from_numpy(full[a:b].astype(np.int32))
    got = gather_counts(local, n_frames)
    np.save(os.path.join(out_dir, "rank%d.npy" % rank), got.numpy())
    # the asynchronous form bench.py uses: step k's gather is collected after step k + 1 has started
    from video.sharding import CountGather
    g = CountGather(n_frames)
    assert g.finish() is None
    g.start(local)
    g.start(local + 1)                                   # (collects the first one internally)
    second = g.finish()
    assert g.finish() is None
    np.save(os.path.join(out_dir, "async_rank%d.npy" % rank), second.numpy())
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("n_frames", [16, 17])
def test_gather_counts_two_ranks_gloo(tmp_path, n_frames):
    import torch.multiprocessing as mp
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    mp.spawn(_worker, args=(2, port, n_frames, str(tmp_path)), nprocs=2, join=True)
    full = ((np.arange(n_frames) * 7 + 3) % 1000).astype(np.int32)
    for r in range(2):
        got = np.load(os.path.join(str(tmp_path), "rank%d.npy" % r))
        assert got.dtype == np.int32 and np.array_equal(got, full)      # frame order, every rank
        assert np.array_equal(np.load(os.path.join(str(tmp_path), "async_rank%d.npy" % r)), full + 1)


# ---------------------------------------------------------------------------- one video, several shards
def _run_worker(tmp, world, extra):
    """world 1: a plain child process; world 2: two fresh children through torch.distributed.run
    (gloo; never an exec of this pytest process)"""
    import subprocess
    worker = os.path.join(ROOT, "tests", "shared_video_worker.py")
    env = dict(os.environ)
    env.setdefault("OMP_NUM_THREADS", "2")
    if world == 1:
        cmd = [sys.executable, worker]
    else:
        s = socket.socket()
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
        s.close()
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(world),
               "--master-addr", "127.0.0.1", "--master-port", str(port), worker]
    out = subprocess.run(cmd + ["--out", str(tmp)] + extra, env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT,
                         universal_newlines=True, timeout=600)
    fails = [ln for ln in out.stdout.splitlines() if "FAILED" in ln or "Error" in ln]
    assert out.returncode == 0, "\n".join(fails[:12]) + "\n...\n" + out.stdout[-1500:]


def _assert_sharded_equals_single(tmp, n_frames, world=2):
    from video.sharding import shard_range
    single = np.load(os.path.join(str(tmp), "counts_world1.npy"))
    sharded = np.load(os.path.join(str(tmp), "counts_world%d.npy" % world))
    assert single.shape == (n_frames,) and single.max() >= 1
    assert np.array_equal(single, sharded)                          # gathered counts, frame order, bit for bit
    whole = np.load(os.path.join(str(tmp), "labels_world1_rank0.npy"))
    for r in range(world):
        a, b = shard_range(n_frames, world, r)
        part = np.load(os.path.join(str(tmp), "labels_world%d_rank%d.npy" % (world, r)))
        assert np.array_equal(part, whole[a:b]), "rank %d" % r       # and the label maps of every shard


@pytest.mark.parametrize("bg", ["exact", "static"])
def test_same_video_two_shards_equal_one_run_cpu_twin(tmp_path, oracle, bg):
    """gathered counts (and label maps) of 2 shards of ONE clip == the single-rank
    run of the same frames, with the running mean handed from shard to shard (exact) and with a static
    background broadcast once.  CPU: per-op calls into the oracle behind the product's ABI."""
    import subprocess
    subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle"), "libvideoanalysis_cpu.so"],
                          stdout=subprocess.DEVNULL)
    extra = ["--lib", "cpu", "--bg", bg, "--frames", "13", "--size", "96x72"]       # ragged shards: 7 + 6
    _run_worker(tmp_path, 1, extra)
    _run_worker(tmp_path, 2, extra)
    _assert_sharded_equals_single(tmp_path, 13)


@pytest.mark.gpu
@pytest.mark.parametrize("bg,engine", [("exact", False), ("exact", True), ("static", True)])
def test_same_video_two_shards_equal_one_run_gpu(tmp_path, bg, engine):
    """the same on the MI355X: 2 x 16 frames at 480p, two fresh child processes sharing the one card
    (gloo for the host-side exchange), per-op calls and the fused pipeline; plus the C ABI's own
    RCCL gather at world 2 -- accepted or refused by RCCL with two ranks on one device, the outcome
    is printed (and checked when it succeeds)."""
    extra = ["--lib", "hip", "--bg", bg, "--frames", "32", "--size", "640x480"] + (["--engine"] if engine else [])
    _run_worker(tmp_path, 1, extra)
    _run_worker(tmp_path, 2, extra + (["--comm"] if engine and bg == "exact" else []))
    _assert_sharded_equals_single(tmp_path, 32)
    if engine and bg == "exact":
        single = np.load(os.path.join(str(tmp_path), "counts_world1.npy"))
        for r in range(2):
            note = open(os.path.join(str(tmp_path), "comm_world2_rank%d.txt" % r)).read().strip()
            print("va_comm world 2, rank %d: %s" % (r, note[:300]))
            if note.startswith("ok "):
                assert [int(v) for v in note.split()[1:]] == single.tolist()

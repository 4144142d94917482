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

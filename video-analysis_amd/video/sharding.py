"""Frame sharding across the GPUs of one node.

Every frame is an independent unit for blur / threshold / morphology / labelling, so a video
is cut into contiguous blocks, one per rank, and the data path needs NO collective.  The only
exchange is the final gather of the per-frame object counts (`gather_counts`), an all-gather of
`frames_per_rank` int32 over torch.distributed -- backend "nccl" (= RCCL over xGMI) on the
GPUs, "gloo" in the CPU tests.

The running-mean background is a temporal recurrence; each shard either starts from an exported
state (`FrameEngine.get_background()` / `set_background()`, exact) or from its own first frame
(independent cameras / clips).  bench.py uses the latter: one clip per rank.
"""


def shard_range(n_frames, world_size, rank):
    """[start, stop) of the contiguous block of frames owned by `rank`; the first
    `n_frames % world_size` ranks get one extra frame"""
    if world_size < 1 or not 0 <= rank < world_size:
        raise ValueError("bad rank %r / world_size %r" % (rank, world_size))
    if n_frames < 0:
        raise ValueError("n_frames must be >= 0")
    base, extra = divmod(n_frames, world_size)
    start = rank * base + min(rank, extra)
    return start, start + base + (1 if rank < extra else 0)


def shard_sizes(n_frames, world_size):
    return [b - a for a, b in (shard_range(n_frames, world_size, r) for r in range(world_size))]


def gather_counts(local_counts, n_frames=None, group=None):
    """all-gather the per-frame object counts of every shard; returns a 1-d int32 tensor with
    the counts of frames 0..n_frames-1 in frame order on every rank.

    local_counts: 1-d int32 torch tensor (device tensor for nccl, CPU tensor for gloo).  Shards
    may differ by one frame; they are padded to the largest shard for the collective."""
    import torch
    import torch.distributed as dist

    if not dist.is_initialized() or dist.get_world_size(group) == 1:
        return local_counts.clone()
    world = dist.get_world_size(group)
    if n_frames is None:
        n_frames = local_counts.numel() * world
    sizes = shard_sizes(n_frames, world)
    cap = max(sizes)
    if local_counts.numel() != sizes[dist.get_rank(group)]:
        raise ValueError("rank %d holds %d counts, its shard has %d frames"
                         % (dist.get_rank(group), local_counts.numel(), sizes[dist.get_rank(group)]))
    recv = torch.empty(world * cap, dtype=torch.int32, device=local_counts.device)
    if all(s == cap for s in sizes):          # equal shards: the counts go out as they are
        dist.all_gather_into_tensor(recv, local_counts.contiguous(), group=group)
        return recv
    send = torch.zeros(cap, dtype=torch.int32, device=local_counts.device)
    send[:local_counts.numel()] = local_counts
    dist.all_gather_into_tensor(recv, send, group=group)
    return torch.cat([recv[r * cap:r * cap + sizes[r]] for r in range(world)])

"""Frame sharding across the GPUs of one node.

Every frame is an independent unit for blur / threshold / morphology / labelling, so a video
is cut into contiguous blocks, one per rank, and the data path needs NO collective.  The only
exchange is the final gather of the per-frame object counts (`gather_counts`), an all-gather of
`frames_per_rank` int32 over torch.distributed -- backend "nccl" (= RCCL over xGMI) on the
GPUs, "gloo" in the CPU tests.

The running-mean background is a temporal recurrence (`mean*n/(n+1) + frame/(n+1)`,
video/analysis/video.py:32-33), so the shards of ONE video share it in one of two ways:

* `broadcast_background` -- a background computed beforehand (static model) goes from one rank to
  all, once; every shard then differences against the same state (option 1);
* `background_handoff`   -- exact cumulative semantics: rank r receives (state, n_seen) as it stands
  after frames [0, start_r), advances a copy over its own frames (state update only, the cheap
  part of the chain) and passes it on to rank r + 1 BEFORE it runs its own chain, so only the
  background passes of the shards serialise (option 3).  Gathered results then equal a single-GPU
  run of the whole video bit for bit.

Independent clips (one camera per rank) need neither; bench.py's default does that.
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


def _to_wire(arr, device):
    import torch
    t = torch.from_numpy(arr)
    return t.to(device) if device is not None else t


def broadcast_background(state, src=0, device=None, group=None):
    """every rank returns rank `src`'s background state (a float64/float32 NumPy array of the frame
    shape; other ranks pass an array of the same shape and dtype, its content is ignored).
    device: torch device the collective runs on (a cuda device for nccl = RCCL; None = CPU, gloo)."""
    import numpy as np
    import torch.distributed as dist

    state = np.ascontiguousarray(state)
    if not dist.is_initialized() or dist.get_world_size(group) == 1:
        return state.copy()
    t = _to_wire(state.copy(), device)
    dist.broadcast(t, src=src, group=group)
    return t.cpu().numpy()


def background_handoff(advance, shape, dtype, first_state=None, first_seen=0, n_local=0, device=None,
                       group=None):
    """exact running-mean semantics across the shards of one video.

    advance(state, n_seen) -> state after this rank's `n_local` frames (the state update alone, e.g.
    va_bg_update with diff_out = NULL on the shard's frames; must not modify its argument).
    Rank 0 starts from (`first_state` or zeros, `first_seen`); rank r > 0 blocks until rank r - 1 has
    advanced.  Every rank returns the (state, n_seen) ITS chain has to start from.  The state moves as
    one float tensor plus one int64 over point-to-point send/recv (gloo or nccl)."""
    import numpy as np
    import torch
    import torch.distributed as dist

    world = dist.get_world_size(group) if dist.is_initialized() else 1
    rank = dist.get_rank(group) if dist.is_initialized() else 0
    if rank == 0:
        state = np.zeros(shape, dtype) if first_state is None else np.array(first_state, dtype).reshape(shape)
        seen = int(first_seen)
    else:
        st = _to_wire(np.empty(shape, dtype), device)
        sn = torch.zeros(1, dtype=torch.int64, device=st.device)
        dist.recv(st, src=rank - 1, group=group)
        dist.recv(sn, src=rank - 1, group=group)
        state, seen = st.cpu().numpy(), int(sn.item())
    if rank + 1 < world:
        nxt = np.ascontiguousarray(advance(state, seen), dtype).reshape(shape)
        st = _to_wire(nxt, device)
        dist.send(st, dst=rank + 1, group=group)
        dist.send(torch.tensor([seen + int(n_local)], dtype=torch.int64, device=st.device), dst=rank + 1, group=group)
    return state, seen


class CountGather(object):
    """the per-step count gather, off the critical path: `start(counts)` enqueues the all-gather of this
    step's per-frame counts (asynchronously: the collective runs on the backend's own stream once the
    counts are ready) and `finish()` hands back the gathered counts of the step started LAST, in frame
    order.  A caller that finishes step k's gather only after it has launched step k + 1's chain never
    stalls its compute stream on the collective.  Give consecutive `start` calls different count
    buffers (the next step's labelling overwrites the one it uses)."""

    def __init__(self, n_frames, group=None):
        import torch.distributed as dist
        self.group = group
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.rank = dist.get_rank(group) if dist.is_initialized() else 0
        self.n_frames = int(n_frames)
        self.sizes = shard_sizes(self.n_frames, self.world)
        self.cap = max(self.sizes) if self.sizes else 0
        self._pending = None            # (work handle or None, recv tensor, send tensor kept alive)

    def start(self, local_counts):
        import torch
        import torch.distributed as dist
        if self._pending is not None:
            self.finish()
        if self.world == 1:
            self._pending = (None, local_counts, None)
            return
        if local_counts.numel() != self.sizes[self.rank]:
            raise ValueError("rank %d holds %d counts, its shard has %d frames"
                             % (self.rank, local_counts.numel(), self.sizes[self.rank]))
        recv = torch.empty(self.world * self.cap, dtype=torch.int32, device=local_counts.device)
        send = local_counts.contiguous()
        if send.numel() != self.cap:                    # ragged shards: pad to the largest one
            send = torch.zeros(self.cap, dtype=torch.int32, device=local_counts.device)
            send[:local_counts.numel()] = local_counts
        work = dist.all_gather_into_tensor(recv, send, group=self.group, async_op=True)
        self._pending = (work, recv, send)

    def finish(self):
        """gathered counts of the last started step (None if nothing is pending); makes the current
        stream wait for the collective, not the host"""
        import torch
        if self._pending is None:
            return None
        work, recv, _send = self._pending
        self._pending = None
        if work is None:
            return recv
        work.wait()
        if all(sz == self.cap for sz in self.sizes):
            return recv
        return torch.cat([recv[r * self.cap:r * self.cap + self.sizes[r]] for r in range(self.world)])

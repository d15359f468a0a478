"""Bin sharding across the GPUs of one node (one process per GPU).

Every frequency bin's (correlate, GEVD, filter) is independent, so rank g
owns the contiguous bin range [lo, hi) and the only exchange is one all-gather of the per-bin
filters.  On GPUs the gather is RCCL (`Engine.allgather_filters_dev`, ncclAllGather over xGMI);
`allgather_filters_host` is the same reassembly through any torch.distributed backend and is what
the CPU (gloo) tests drive.  The reference has no distributed code at all.
"""
import numpy as np


def shard_bins(n_bins, world, rank):
    """Contiguous, balanced split of range(n_bins): returns (lo, hi) for `rank`.

    The first n_bins % world ranks get one extra bin; an equal split (what RCCL's all-gather needs)
    results whenever world divides n_bins, e.g. 4096 bins on 8 GPUs -> 512 each (BASELINE config 4).
    """
    if not 0 <= rank < world:
        raise ValueError("rank out of range")
    base, extra = divmod(n_bins, world)
    lo = rank * base + min(rank, extra)
    hi = lo + base + (1 if rank < extra else 0)
    return lo, hi


def padded_shard(n_bins, world):
    """Bins per rank when every rank must hold the same count (ncclAllGather): ceil(n_bins/world)."""
    return -(-n_bins // world)


def allgather_filters_host(w_shard, n_bins, group=None):
    """Reassemble w (n_bins, nV, L) on every rank from per-rank shards, through torch.distributed.

    w_shard: this rank's (hi-lo, nV, L) complex array.  Shards are padded to equal length for the
    collective and trimmed afterwards, so ragged splits work too.
    """
    import torch
    import torch.distributed as dist
    world = dist.get_world_size(group)
    per = padded_shard(n_bins, world)
    w_shard = np.ascontiguousarray(w_shard)
    tail = w_shard.shape[1:]
    buf = np.zeros((per,) + tail, dtype=w_shard.dtype)
    buf[: w_shard.shape[0]] = w_shard
    t = torch.from_numpy(buf.view(np.float32 if w_shard.dtype == np.complex64 else np.float64))
    out = [torch.empty_like(t) for _ in range(world)]
    dist.all_gather(out, t, group=group)
    parts = []
    for g in range(world):
        lo, hi = shard_bins(n_bins, world, g)
        parts.append(out[g].numpy().view(w_shard.dtype)[: hi - lo])
    return np.concatenate(parts, axis=0)

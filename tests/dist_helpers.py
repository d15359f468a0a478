"""Test-only: the all-gather of the per-bin filters through torch.distributed (gloo on CPU), standing in for
`Engine.allgather_filters_dev` (RCCL) on a box without GPUs.  Uses the product's own `shard_bins` / `padded_shard`."""
import numpy as np

from ap_vast_unofficial_amd.sharding import padded_shard, shard_bins


def allgather_filters_host(w_shard, n_bins, group=None):
    """Reassemble w (n_bins, nV, L) on every rank from per-rank shards.

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

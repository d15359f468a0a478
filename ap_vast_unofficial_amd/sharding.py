"""Bin sharding across the GPUs of one node (one process per GPU).

Every frequency bin's (correlate, GEVD, filter) is independent, so rank g
owns the contiguous bin range [lo, hi) and the only exchange is one all-gather of the per-bin
filters: RCCL (`Engine.allgather_filters_dev`, ncclAllGather over xGMI).  The CPU (gloo) tests drive the
same reassembly through torch.distributed with a helper of their own (tests/dist_helpers.py); nothing in
this package imports torch.  The reference has no distributed code at all.
"""


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

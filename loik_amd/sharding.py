"""Multi-GPU sharding of the batch: instances are independent, so every rank owns a contiguous slice of the batch
and solves it with no exchange step -- there is NO collective on the data path.  The process group
(gloo) only carries the timing barrier, the max-over-ranks of the elapsed time and the sum of per-rank counters."""
import numpy as np


def shard_bounds(total, rank, world):
    """contiguous partition of `total` instances over `world` ranks: [lo, hi) of `rank`"""
    lo = (total * rank) // world
    hi = (total * (rank + 1)) // world
    return lo, hi


def shard_workload(wl, rank, world):
    """slice every per-instance array of a workload dict (leading dimension == batch) for `rank`"""
    B = wl["q"].shape[0]
    lo, hi = shard_bounds(B, rank, world)
    out = {}
    for k, v in wl.items():
        if isinstance(v, np.ndarray) and v.ndim >= 1 and v.shape[0] == B and k not in ("lb", "ub", "H_ref", "v_ref", "c_ids"):
            out[k] = v[lo:hi]
        elif isinstance(v, np.ndarray) and k in ("lb", "ub") and v.ndim == 2:
            out[k] = v[lo:hi]
        else:
            out[k] = v
    return out


def aggregate(dist, elapsed, counters):
    """max over ranks of the elapsed time, sum over ranks of every counter; identity without a process group"""
    if dist is None:
        return elapsed, dict(counters)
    import torch
    t = torch.tensor([elapsed], dtype=torch.float64)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    keys = sorted(counters)
    c = torch.tensor([float(counters[k]) for k in keys], dtype=torch.float64)
    dist.all_reduce(c, op=dist.ReduceOp.SUM)
    return float(t[0]), {k: float(c[i]) for i, k in enumerate(keys)}

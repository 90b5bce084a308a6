"""Query sharding across ranks (one process per GPU, torch.distributed).  The index is replicated, queries are
cut into contiguous shards (query i -> rank floor(i * world / n), the same rule awry_set_devices uses inside
one process) and there is NO collective on the search path: the only communication is the gather of results
for callers that want the whole batch's answer on every rank.

`count_fn(qbytes, qoff) -> uint64[n]` and `locate_fn(qbytes, qoff) -> (hit_off, gpos, pos)` are the engine
entry points (FmIndex.parallel_count_csr / parallel_locate_csr on the rank's GPU)."""
import numpy as np


def shard_bounds(n, world, rank):
    return n * rank // world, n * (rank + 1) // world


def slice_csr(qbytes, qoff, lo, hi):
    qoff = np.asarray(qoff, dtype=np.uint64)
    b0, b1 = int(qoff[lo]), int(qoff[hi])
    return np.asarray(qbytes, dtype=np.uint8)[b0:b1], qoff[lo:hi + 1] - qoff[lo]


def _all_gather_var(arr, dist, group=None):
    """all_gather of 1-D uint64/int64 arrays of different lengths -> list of arrays (rank order)"""
    import torch
    world = dist.get_world_size(group)
    a = np.ascontiguousarray(arr).view(np.int64).reshape(-1)
    lens = [torch.zeros(1, dtype=torch.int64) for _ in range(world)]
    dist.all_gather(lens, torch.tensor([a.size], dtype=torch.int64), group=group)
    m = max(int(x.item()) for x in lens)
    pad = torch.zeros(max(m, 1), dtype=torch.int64)
    pad[:a.size] = torch.from_numpy(a.copy())
    bufs = [torch.zeros(max(m, 1), dtype=torch.int64) for _ in range(world)]
    dist.all_gather(bufs, pad, group=group)
    return [b[:int(k.item())].numpy().view(np.uint64) for b, k in zip(bufs, lens)]


def sharded_count(count_fn, qbytes, qoff, dist=None, group=None):
    """every rank counts its contiguous shard; returns the whole batch's counts, in input order, on every rank"""
    n = len(qoff) - 1
    if dist is None or not dist.is_initialized() or dist.get_world_size(group) == 1:
        return np.asarray(count_fn(qbytes, qoff), dtype=np.uint64)
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    lo, hi = shard_bounds(n, world, rank)
    mine = np.asarray(count_fn(*slice_csr(qbytes, qoff, lo, hi)), dtype=np.uint64)
    return np.concatenate(_all_gather_var(mine, dist, group))


def sharded_locate(locate_fn, qbytes, qoff, dist=None, group=None):
    """-> (hit_off uint64[n+1], gpos uint64[total], pos uint64[total, 2]) for the whole batch on every rank;
    per-shard CSR offsets are rebased on the host"""
    n = len(qoff) - 1
    if dist is None or not dist.is_initialized() or dist.get_world_size(group) == 1:
        return locate_fn(qbytes, qoff)
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    lo, hi = shard_bounds(n, world, rank)
    off, gpos, pos = locate_fn(*slice_csr(qbytes, qoff, lo, hi))
    counts = np.concatenate(_all_gather_var(np.diff(np.asarray(off, dtype=np.uint64)), dist, group))
    g = np.concatenate(_all_gather_var(np.asarray(gpos, dtype=np.uint64), dist, group))
    p = np.concatenate(_all_gather_var(np.asarray(pos, dtype=np.uint64).reshape(-1), dist, group)).reshape(-1, 2)
    hit_off = np.zeros(n + 1, dtype=np.uint64)
    hit_off[1:] = np.cumsum(counts, dtype=np.uint64)
    return hit_off, g, p

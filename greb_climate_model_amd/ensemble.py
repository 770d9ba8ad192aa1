"""Ensemble axis = the multi-GPU axis: members are independent GREB runs -- what
separate `ens_id` processes are in the reference (src/greb.f90:153,1064-1068) -- farmed over
ranks with no data-path collective; the only exchange is the gather of monthly means to rank 0.

Pure host logic + torch.distributed plumbing (gloo on CPU in tests, nccl == RCCL on the node).
"""
from __future__ import annotations

import numpy as np


def co2_sweep(n_members: int, lo: float = 280.0, hi: float = 1120.0) -> np.ndarray:
    """BASELINE config 4's sweep generalised to any member count: 8 members give
    280, 400, ..., 1120 ppm exactly."""
    if n_members == 1:
        return np.asarray([680.0], np.float32)
    return (lo + (hi - lo) * np.arange(n_members, dtype=np.float64) / (n_members - 1)).astype(np.float32)


def partition(n_members: int, world: int, rank: int) -> np.ndarray:
    """Global member ids owned by `rank`: contiguous blocks, sizes differing by at most one."""
    base, rem = divmod(n_members, world)
    start = rank * base + min(rank, rem)
    return np.arange(start, start + base + (1 if rank < rem else 0))


def gather_monthly(local, n_members: int, group=None):
    """Gather per-rank monthly means [m_local, ...] to rank 0 -> [n_members, ...] (None elsewhere).
    Ragged member counts are padded to the largest block for the collective and trimmed after."""
    import torch
    import torch.distributed as dist
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    counts = [len(partition(n_members, world, r)) for r in range(world)]
    mmax = max(counts)
    pad = local
    if local.shape[0] < mmax:
        pad = torch.zeros((mmax,) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
        pad[: local.shape[0]] = local
    bufs = [torch.empty_like(pad) for _ in range(world)] if rank == 0 else None
    dist.gather(pad.contiguous(), bufs, dst=0, group=group)
    if rank != 0:
        return None
    return torch.cat([b[:c] for b, c in zip(bufs, counts)], dim=0)


def ensemble_stats(local, group=None):
    """On-device ensemble mean and (population) variance of per-member fields across ALL ranks
    (SURVEY.md 8f-4): two all-reduces (RCCL on the node) of the member-sum and member-sum-of-squares,
    accumulated in fp64 so the result does not depend on how members are dealt to ranks.
    local: [m_local, ...] -> (mean[...], var[...]) on every rank."""
    import torch
    import torch.distributed as dist
    x = local.to(torch.float64)
    s1 = x.sum(0)
    s2 = (x * x).sum(0)
    n = torch.tensor([float(local.shape[0])], dtype=torch.float64, device=local.device)
    if dist.is_available() and dist.is_initialized():
        for t in (s1, s2, n):
            dist.all_reduce(t, op=dist.ReduceOp.SUM, group=group)
    mean = s1 / n
    var = (s2 / n - mean * mean).clamp_min(0.0)
    return mean.to(local.dtype), var.to(local.dtype)


class MonthlyGather:
    """Year-by-year gather of the monthly means to rank 0, overlapped with the next year's
    integration: each submitted year is one asynchronous `gather` on the collective's own stream
    (RCCL over xGMI on the node: 7 peers x 1.1 MB per member-year into rank 0, every peer on its own
    link), so the timed region only waits for the last year's transfer.  Equal member counts per rank."""

    def __init__(self, members_per_rank: int, years: int, tail_shape, dtype, device, group=None):
        import torch
        import torch.distributed as dist
        self.dist, self.group = dist, group
        self.world, self.rank = dist.get_world_size(group), dist.get_rank(group)
        self.m, self.years = members_per_rank, years
        self.buf = None
        if self.rank == 0:  # [year][rank][member, ...]: every (year, rank) block is contiguous
            self.buf = torch.empty((years, self.world, members_per_rank) + tuple(tail_shape), dtype=dtype, device=device)
        self.work = []

    def submit(self, year: int, local_year):
        """local_year: [members_per_rank, ...] of this rank for `year` (must stay alive until finish())."""
        gl = [self.buf[year, r] for r in range(self.world)] if self.rank == 0 else None
        self.work.append(self.dist.gather(local_year, gl, dst=0, group=self.group, async_op=True))

    def finish(self):
        """Wait for every transfer; rank 0 gets a view [n_members_total, years, ...], others None."""
        for w in self.work:
            w.wait()
        self.work = []
        if self.rank != 0:
            return None
        y, w, m = self.buf.shape[:3]
        return self.buf.permute(1, 2, 0, *range(3, self.buf.dim())).reshape((w * m, y) + tuple(self.buf.shape[3:]))

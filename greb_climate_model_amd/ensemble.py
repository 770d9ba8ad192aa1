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


def splitmix64(seed: int, n: int) -> np.ndarray:
    """n deterministic uniforms in [0,1) from SplitMix64."""
    out, x, M = [], seed & (2**64 - 1), 2**64 - 1
    for _ in range(n):
        x = (x + 0x9E3779B97F4A7C15) & M
        z = x
        z = ((z ^ (z >> 30)) * 0xBF58476D1CE4E5B9) & M
        z = ((z ^ (z >> 27)) * 0x94D049BB133111EB) & M
        z ^= z >> 31
        out.append((z >> 11) / float(1 << 53))
    return np.asarray(out)


PERTURBED = ("da_ice", "a_no_ice", "a_cloud", "kappa")


def perturbed_physics(n_members: int, params, seed: int = 20261004, spread: float = 0.1) -> np.ndarray:
    """BASELINE config 5's members: albedo (da_ice, a_no_ice, a_cloud) and diffusivity (kappa) drawn uniformly within
    +-`spread` of the namelist value, member index as the stream position.  Returns float32
    [n_members][4] in the order of greb_member_overrides -- what N separate `ens_id` processes with N different
    &PHYSICS_PAR groups are in the reference (src/greb.f90:128-132,153)."""
    u = splitmix64(seed, 4 * n_members).reshape(n_members, 4)
    base = np.asarray([getattr(params, k) for k in PERTURBED], np.float64)
    return (base[None] * (1.0 - spread + 2.0 * spread * u)).astype(np.float32)


def max_chain_sweeps(kappa: float, nx: int, ny: int, dt_crcl: float = 1800.0, pi: float = 3.1416) -> int:
    """The largest number of DEPENDENT zonal diffusion sweeps any latitude row needs per diffusion call for this
    diffusivity (src/greb.f90:578-580, 652-653, in fp32 as the reference evaluates it; dtdff2 == 0 is the polar-row case
    SURVEY.md App. B describes: one sweep).  It is what a member costs in latency: 8 at 96x48, 225 at 384x192 -- and
    1 800 at 384x192 once kappa falls below 7.27e5."""
    f = np.float32
    dtc, kap = f(dt_crcl), f(kappa)
    deg = f(2.0) * f(pi) * f(6.371e6) / f(360.0)
    dlon, dlat = f(360.0) / f(nx), f(180.0) / f(ny)
    nint = lambda x: int(np.floor(abs(float(x)) + 0.5) * (1 if x >= 0 else -1))
    worst = 1
    for k in range(ny):
        lat = dlat * f(k + 1) - dlat / f(2.0) - f(90.0)
        dxlat = dlon * deg * f(np.cos(np.float32(f(2.0) * f(pi) / f(360.0) * lat)))
        if dxlat > f(2.5e5):
            continue  # full-row branch: no sub-cycling (:592)
        dd = max(1, nint(dtc / (f(1.0) * (dxlat * dxlat) / kap)))
        dtdff2 = int(dtc / f(dd))
        worst = max(worst, 1 if dtdff2 == 0 else max(1, nint(dtc / f(dtdff2))))
    return worst


def latency_groups(kappas, nx: int, ny: int, ratio: float = 3.0) -> list:
    """Member indices grouped by what bounds them.  All members of one engine advance in lock step (one launch per
    sub-step), so the launch is as long as the slowest member's longest chain; members whose chains are `ratio` times
    longer than the ensemble's median go into a group of their own, to be run by a second engine BESIDE the first
    (members are independent: the reference runs every one of them as its own process).  Returns one or two index
    arrays, the large group first."""
    sweeps = np.asarray([max_chain_sweeps(float(k), nx, ny) for k in kappas])
    long_ = sweeps > ratio * np.median(sweeps)
    if not long_.any() or long_.all():
        return [np.arange(len(sweeps))]
    return [np.flatnonzero(~long_), np.flatnonzero(long_)]


def run_beside(jobs) -> list:
    """Run the callables in `jobs` concurrently, one host thread each (the engine calls release the GIL and every engine
    has its own HIP stream), and return their results in order; the first exception is re-raised."""
    import threading
    out, err = [None] * len(jobs), [None] * len(jobs)

    def work(i):
        try:
            out[i] = jobs[i]()
        except BaseException as e:  # noqa: BLE001 -- handed to the caller
            err[i] = e

    ts = [threading.Thread(target=work, args=(i,)) for i in range(len(jobs))]
    for t in ts:
        t.start()
    for t in ts:
        t.join()
    for e in err:
        if e is not None:
            raise e
    return out


def partition(n_members: int, world: int, rank: int) -> np.ndarray:
    """Global member ids owned by `rank`: contiguous blocks, sizes differing by at most one."""
    base, rem = divmod(n_members, world)
    start = rank * base + min(rank, rem)
    return np.arange(start, start + base + (1 if rank < rem else 0))


def partition_root(members: int, root_members: int, world: int, rank: int) -> np.ndarray:
    """Global member ids owned by `rank` when rank 0 -- which also hosts the receive side of every gather -- carries
    `root_members` members and every other rank `members`: contiguous blocks in rank order."""
    if rank == 0:
        return np.arange(0, root_members)
    start = root_members + (rank - 1) * members
    return np.arange(start, start + members)


def rank_report(values, group=None, device="cpu") -> dict:
    """Every rank's timings on rank 0: `values` = {name: seconds} of this rank -> {name: {"per_rank": [...], "min": ,
    "max": , "rank_of_max": }} (all-gather of one small tensor over the run's own backend).  A straggler -- rank 0, if
    RCCL's receive kernels take compute units from its integration -- shows here, where the max-over-ranks `dt` of the
    benchmark contract cannot show it."""
    import torch
    import torch.distributed as dist
    names = sorted(values)
    mine = torch.tensor([float(values[k]) for k in names], dtype=torch.float64, device=device)
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    if world > 1:
        parts = [torch.empty_like(mine) for _ in range(world)]
        dist.all_gather(parts, mine, group=group)
        allv = torch.stack(parts).cpu().numpy()
    else:
        allv = mine.cpu().numpy()[None]
    return {k: {"per_rank": [round(float(x), 6) for x in allv[:, i]], "min": round(float(allv[:, i].min()), 6),
                "max": round(float(allv[:, i].max()), 6), "rank_of_max": int(allv[:, i].argmax())} for i, k in enumerate(names)}


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


def ranks_seen(device="cpu", group=None) -> int:
    """How many distinct ranks the collective backend really connects: every rank contributes its id to one
    all_gather (RCCL on the node, gloo in the CPU tests); 1 without a process group."""
    import torch
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()):
        return 1
    mine = torch.tensor([dist.get_rank(group)], dtype=torch.int64, device=device)
    ids = [torch.empty_like(mine) for _ in range(dist.get_world_size(group))]
    dist.all_gather(ids, mine, group=group)
    return len({int(t.item()) for t in ids})


def gather_order_check(member_values, block: int = 0) -> dict:
    """Are the gathered members in GLOBAL member order?  member_values[g] is a quantity that grows with the member
    index g -- in a CO2 sweep (co2_sweep) the annual-mean surface temperature of the last year -- so a block that
    landed in another rank's slot, or members permuted inside a block, show up as inversions.  Neighbouring members of a
    very fine sweep (4 096 members: 0.2 ppm apart) differ by less than 1e-3 K, so besides the neighbour test the values
    are compared `stride` members apart (n / 512, at least 1), where the signal is far above any rounding; with `block`
    (members per rank) every rank's block mean must grow with the rank as well.  Verified = no inversion at that
    stride, none between blocks, all finite."""
    v = np.asarray(member_values, np.float64)
    n = int(v.size)
    inv = int(np.count_nonzero(np.diff(v) <= 0)) if n > 1 else 0
    stride = max(1, n // 512)
    inv_s = int(np.count_nonzero(v[stride:] - v[:-stride] <= 0)) if n > stride else 0
    inv_b = 0
    if block and n % block == 0 and n // block > 1:
        inv_b = int(np.count_nonzero(np.diff(v.reshape(-1, block).mean(axis=1)) <= 0))
    ok = bool(inv_s == 0 and inv_b == 0 and np.isfinite(v).all())
    return {"gather_verified": ok, "inversions": inv, "inversions_at_stride": inv_s, "stride": stride,
            "block_inversions": inv_b, "members_checked": n}


def local_moments(x):
    """Per-element fp64 sum and sum of squares, min and max over this GPU's members: one pass of the HIP kernel
    greb_ensemble_moments_dev over x [m_local, ...] (float32, CUDA, contiguous).  No CPU path."""
    import ctypes as C

    import torch

    from . import engine
    if not (x.is_cuda and x.dtype == torch.float32 and x.is_contiguous()):
        raise engine.GrebError(-1, "local_moments: needs a contiguous float32 CUDA tensor (no CPU path)")
    m, tail = x.shape[0], tuple(x.shape[1:])
    n = int(x[0].numel()) if m else 0
    s1 = torch.zeros(tail, dtype=torch.float64, device=x.device)
    s2 = torch.zeros(tail, dtype=torch.float64, device=x.device)
    lo = torch.full(tail, float("inf"), dtype=torch.float32, device=x.device)
    hi = torch.full(tail, float("-inf"), dtype=torch.float32, device=x.device)
    if m:
        f = engine.lib().greb_ensemble_moments_dev
        f.argtypes = [C.c_void_p, C.c_int, C.c_size_t, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
        rc = f(x.data_ptr(), m, n, s1.data_ptr(), s2.data_ptr(), lo.data_ptr(), hi.data_ptr(),
               torch.cuda.current_stream(x.device).cuda_stream)
        if rc:
            raise engine.GrebError(rc, "greb_ensemble_moments_dev")
    return s1, s2, lo, hi


def ensemble_summary(local, group=None, moments_fn=local_moments):
    """Ensemble mean, (population) variance, min and max of per-member fields across ALL ranks:
    each rank reduces its own members on the device (moments_fn, the HIP kernel), then three all-reduces (RCCL on
    the node) combine the fp64 partial sums / the ranges, so the result does not depend on how members are
    dealt to ranks.  local: [m_local, ...] -> dict of tensors [...] on every rank."""
    import torch
    import torch.distributed as dist
    s1, s2, lo, hi = moments_fn(local)
    n = torch.tensor([float(local.shape[0])], dtype=torch.float64, device=local.device)
    if dist.is_available() and dist.is_initialized():
        for t in (s1, s2, n):
            dist.all_reduce(t, op=dist.ReduceOp.SUM, group=group)
        dist.all_reduce(lo, op=dist.ReduceOp.MIN, group=group)
        dist.all_reduce(hi, op=dist.ReduceOp.MAX, group=group)
    mean = s1 / n
    var = (s2 / n - mean * mean).clamp_min(0.0)
    return {"mean": mean.to(local.dtype), "var": var.to(local.dtype), "min": lo, "max": hi, "n": int(n.item())}


def ensemble_stats(local, group=None, moments_fn=local_moments):
    """(mean, variance) of ensemble_summary."""
    s = ensemble_summary(local, group, moments_fn)
    return s["mean"], s["var"]


def ensemble_quantiles(x, probs):
    """Quantiles across the members of x [n_members, ...] (all on this GPU, e.g. rank 0 after the gather), numpy's
    default definition (linear interpolation of the order statistics): HIP kernel greb_ensemble_quantiles_dev
    (per-point bitonic sort of the members in LDS).  Returns [len(probs), ...]."""
    import ctypes as C

    import numpy as np
    import torch

    from . import engine
    if not (x.is_cuda and x.dtype == torch.float32 and x.is_contiguous()):
        raise engine.GrebError(-1, "ensemble_quantiles: needs a contiguous float32 CUDA tensor (no CPU path)")
    pr = np.ascontiguousarray(probs, np.float32)
    out = torch.empty((len(pr),) + tuple(x.shape[1:]), dtype=torch.float32, device=x.device)
    f = engine.lib().greb_ensemble_quantiles_dev
    f.argtypes = [C.c_void_p, C.c_int, C.c_size_t, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]
    rc = f(x.data_ptr(), x.shape[0], int(x[0].numel()), pr.ctypes.data, len(pr), out.data_ptr(),
           torch.cuda.current_stream(x.device).cuda_stream)
    if rc:
        raise engine.GrebError(rc, "greb_ensemble_quantiles_dev")
    return out


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

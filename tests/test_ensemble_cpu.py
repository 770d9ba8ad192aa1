"""CPU, world_size 2 over gloo: the ensemble axis.  Members are sharded over
ranks with no data-path collective; the only exchange is the gather of monthly means to rank 0."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from greb_climate_model_amd import ensemble


def test_co2_sweep_matches_config4():
    assert np.allclose(ensemble.co2_sweep(8), [280, 400, 520, 640, 760, 880, 1000, 1120])
    assert ensemble.co2_sweep(1)[0] == 680.0
    s = ensemble.co2_sweep(512)
    assert s[0] == 280.0 and s[-1] == 1120.0 and np.all(np.diff(s) > 0)


@pytest.mark.parametrize("n,world", [(8, 2), (7, 2), (64, 8), (5, 8), (1, 4)])
def test_partition_covers_every_member_once(n, world):
    parts = [ensemble.partition(n, world, r) for r in range(world)]
    assert sorted(np.concatenate(parts).tolist()) == list(range(n))
    assert max(len(p) for p in parts) - min(len(p) for p in parts) <= 1


def _worker(rank, world, port, n_members, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        ids = ensemble.partition(n_members, world, rank)
        # stand-in for the engine's output: member id encoded in the data, [m_local, years, 12, 5, 8]
        local = torch.stack([torch.full((2, 12, 5, 8), float(i)) + torch.arange(8.0) for i in ids]) \
            if len(ids) else torch.zeros((0, 2, 12, 5, 8))
        out = ensemble.gather_monthly(local, n_members)
        # the per-rank member reduction is a HIP kernel (no CPU path); here only the cross-rank composition is
        # under test, so a torch restatement of the partial moments stands in for it
        def ref_moments(x):
            x64 = x.to(torch.float64)
            tail = x.shape[1:]
            lo = x.min(0).values if x.shape[0] else torch.full(tail, float("inf"))
            hi = x.max(0).values if x.shape[0] else torch.full(tail, float("-inf"))
            return x64.sum(0), (x64 * x64).sum(0), lo, hi
        summ = ensemble.ensemble_summary(local, moments_fn=ref_moments)
        mean, var = summ["mean"], summ["var"]
        full = torch.stack([torch.full((2, 12, 5, 8), float(i)) + torch.arange(8.0) for i in range(n_members)])
        assert torch.allclose(mean, full.mean(0), atol=1e-6) and torch.allclose(var, full.var(0, unbiased=False), atol=1e-5)
        assert torch.equal(summ["min"], full.min(0).values) and torch.equal(summ["max"], full.max(0).values) and summ["n"] == n_members
        if rank == 0:
            ok = out.shape[0] == n_members and all(
                torch.equal(out[i], torch.full((2, 12, 5, 8), float(i)) + torch.arange(8.0)) for i in range(n_members))
            q.put(bool(ok))
        else:
            assert out is None
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("n_members", [8, 7])
def test_gather_monthly_world2(n_members):
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, n_members, q)) for r in range(2)]
    for p in procs: p.start()
    for p in procs: p.join(120)
    assert all(p.exitcode == 0 for p in procs)
    assert q.get(timeout=5) is True


def _worker_pipeline(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        M, K = 3, 4
        g = ensemble.MonthlyGather(M, K, (12, 5, 8), torch.float32, "cpu")
        keep = []
        for y in range(K):  # member id and year encoded in the data
            t = torch.stack([torch.full((12, 5, 8), float(100 * (rank * M + i) + y)) for i in range(M)])
            keep.append(t)
            g.submit(y, t)
        out = g.finish()
        if rank == 0:
            ok = tuple(out.shape) == (world * M, K, 12, 5, 8) and all(
                float(out[mm, y, 0, 0, 0]) == 100 * mm + y for mm in range(world * M) for y in range(K))
            q.put(bool(ok))
        else:
            assert out is None
    finally:
        dist.destroy_process_group()


def test_monthly_gather_pipeline_world2():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker_pipeline, args=(r, 2, port, q)) for r in range(2)]
    for p in procs: p.start()
    for p in procs: p.join(120)
    assert all(p.exitcode == 0 for p in procs)
    assert q.get(timeout=5) is True


def test_partition_root_blocks():
    from greb_climate_model_amd import ensemble
    blocks = [ensemble.partition_root(8, 3, 4, r) for r in range(4)]
    assert [len(b) for b in blocks] == [3, 8, 8, 8]
    assert np.array_equal(np.concatenate(blocks), np.arange(27))
    assert np.array_equal(ensemble.partition_root(8, 8, 4, 2), ensemble.partition(32, 4, 2))

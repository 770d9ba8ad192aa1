#!/usr/bin/env python3
"""Throughput of the any-grid engine at a given grid: python tools/bench_grid.py nx ny members [members ...]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from greb_climate_model_amd import engine, ensemble, workload
import torch
nx, ny = int(sys.argv[1]), int(sys.argv[2])
inp = workload.make_inputs(nx, ny)
p = engine.params_default(); p.ipx, p.ipy = nx - 4, ny - 40
for M in [int(x) for x in sys.argv[3:]]:
    e = engine.Engine(inp, p, n_members=M)
    e.flux_correction(1)
    buf = torch.empty((M, 1, 12, 5, e.np), dtype=torch.float32, device="cuda")
    t = time.perf_counter(); e.run(1, ensemble.co2_sweep(M)[:, None], monthly_dev_ptr=buf.data_ptr()); torch.cuda.synchronize(); dt = time.perf_counter() - t
    print(f"{nx}x{ny} members={M}: {dt:.3f} s/yr = {dt / 730 / 24 * 1e6:.2f} us per sub-step, {M / dt:.2f} member-yr/s; {e.describe()}; finite={bool(torch.isfinite(buf).all())}", flush=True)
    e.close(); del buf

#!/usr/bin/env python3
"""Throughput of the any-grid engine at 384x192 (BASELINE configs 3 and 5): member-years/s vs member count."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from greb_climate_model_amd import engine, ensemble, workload, abi
import torch
inp = workload.make_inputs(384, 192)
p = engine.params_default(); p.ipx, p.ipy = 380, 150
for M in [int(x) for x in (sys.argv[1:] or ["1", "8", "32"])]:
    ov = None
    e = engine.Engine(inp, p, n_members=M)
    t = time.perf_counter(); e.flux_correction(1); tf = time.perf_counter() - t
    buf = torch.empty((M, 1, 12, 5, e.np), dtype=torch.float32, device="cuda")
    lv = ensemble.co2_sweep(M)[:, None]
    t = time.perf_counter(); e.run(1, lv, monthly_dev_ptr=buf.data_ptr()); torch.cuda.synchronize(); dt = time.perf_counter() - t
    print(f"384x192 members={M}: flux year (shared, 1 member) {tf:.2f}s; scenario year {dt:.2f}s -> {M/dt:.2f} member-years/s "
          f"finite={bool(torch.isfinite(buf).all())}", flush=True)
    e.close(); del buf

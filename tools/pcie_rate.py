#!/usr/bin/env python3
"""PCIe-inclusive rate: the same ensemble run but with the monthly means delivered to HOST memory
through greb_engine_run's default (host-pointer) path.  Reported in DESIGN.md, never as bench `value`."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from greb_climate_model_amd import engine, ensemble, workload
M, K = int(sys.argv[1]) if len(sys.argv) > 1 else 256, 4
inp = workload.make_inputs()
p = engine.params_default(); p.ipx, p.ipy = 95, 38
t = time.perf_counter(); e = engine.Engine(inp, p, n_members=M); t_create = time.perf_counter() - t
e.flux_correction(1)
lv = np.repeat(ensemble.co2_sweep(M)[:, None], K, 1)
e.run(1, lv[:, :1])
t = time.perf_counter(); mon, yr = e.run(K, lv); dt = time.perf_counter() - t
print(f"members={M} years={K}: create(upload 94 MB)={t_create:.2f}s  run+D2H={dt:.3f}s -> {M*K/dt:.1f} yr/s PCIe-inclusive, "
      f"{mon.nbytes/1e9:.2f} GB of monthly means to host")

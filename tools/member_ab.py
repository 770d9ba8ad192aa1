#!/usr/bin/env python3
"""A/B of release-library variants on the bench's main leg (512 members, 96x48, scenario years): GREB_LIB=<.so> python tools/member_ab.py
Prints simulated-years/s; run several variants in one gpurun call, alternating, on the same box."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from greb_climate_model_amd import engine, ensemble, workload
if os.environ.get("GREB_LIB"):
    engine._lib_path = os.path.abspath(os.environ["GREB_LIB"])
M = int(sys.argv[1]) if len(sys.argv) > 1 else 512
inp = workload.make_inputs()
p = engine.params_default(); p.ipx, p.ipy = 95, 38
e = engine.Engine(inp, p, n_members=M)
e.flux_correction(1)
co2 = np.linspace(280.0, 1120.0, M).astype(np.float32)
buf = torch.empty((M, 1, 12, 5, e.np), dtype=torch.float32, device="cuda")
rates = []
for i in range(4):
    torch.cuda.synchronize(); t = time.perf_counter()
    e.run(1, co2[:, None], monthly_dev_ptr=buf.data_ptr()); torch.cuda.synchronize()
    rates.append(M / (time.perf_counter() - t))
print(f"{os.environ.get('GREB_LIB', 'release'):45s} {M} members: " + " ".join(f"{r:.0f}" for r in rates) + f"  yr/s (best {max(rates):.0f}) finite={bool(torch.isfinite(buf).all())}")
e.close()

#!/usr/bin/env python3
"""A stretch of the 384x192 engine for rocprofv3 (kernel trace or one --pmc pass): python tools/prof_g384.py [members] [steps]
members > 1: the perturbed-physics draws of config 5 without the 1 800-sweep ones (tools/g384_ab.py)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from greb_climate_model_amd import engine, ensemble, workload
if os.environ.get("GREB_TUNING_LIB"):
    engine.use_tuning_build()
M = int(sys.argv[1]) if len(sys.argv) > 1 else 1
inp = workload.make_inputs(384, 192)
p = engine.params_default(); p.ipx, p.ipy = 380, 152
ov = None
if M > 1:
    o = ensemble.perturbed_physics(64, p); o = o[o[:, 3] >= 7.27e5][:M]
    ov = [dict(zip(ensemble.PERTURBED, map(float, r))) for r in o]
e = engine.Engine(inp, p, n_members=M, overrides=ov)
buf = torch.empty((M, 1, 12, 5, e.np), dtype=torch.float32, device="cuda")
e.run(1, 680.0, monthly_dev_ptr=buf.data_ptr()); torch.cuda.synchronize()
e.close()

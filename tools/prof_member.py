#!/usr/bin/env python3
"""One scenario year of the fused 96x48 member kernel for a rocprofv3 --pmc pass: python tools/prof_member.py [members]
With GREB_TUNING_LIB=1 the -DGREB_TUNING library is loaded, whose GREB_DEBUG_NSUB=0 leaves a model step without its 24
circulation sub-steps: the point-physics phase (and the wind staging) on its own."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from greb_climate_model_amd import engine, ensemble, workload
if os.environ.get("GREB_TUNING_LIB"):
    engine.use_tuning_build()
M = int(sys.argv[1]) if len(sys.argv) > 1 else 512
inp = workload.make_inputs()
p = engine.params_default(); p.ipx, p.ipy = 95, 38
e = engine.Engine(inp, p, n_members=M)
buf = torch.empty((M, 1, 12, 5, e.np), dtype=torch.float32, device="cuda")
e.run(1, ensemble.co2_sweep(M)[:, None], monthly_dev_ptr=buf.data_ptr()); torch.cuda.synchronize()
e.run(1, ensemble.co2_sweep(M)[:, None], monthly_dev_ptr=buf.data_ptr()); torch.cuda.synchronize()
e.close()

#!/usr/bin/env python3
"""One scenario month-ish of the 384x192 engine for rocprofv3 --kernel-trace --stats: python tools/prof_g384.py [members]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from greb_climate_model_amd import engine, workload
M = int(sys.argv[1]) if len(sys.argv) > 1 else 1
inp = workload.make_inputs(384, 192)
p = engine.params_default(); p.ipx, p.ipy = 380, 152
e = engine.Engine(inp, p, n_members=M)
buf = torch.empty((M, 1, 12, 5, e.np), dtype=torch.float32, device="cuda")
e.run(1, 680.0, monthly_dev_ptr=buf.data_ptr()); torch.cuda.synchronize()
e.close()

#!/usr/bin/env python3
"""One 384x192 member, a tenth of a model year (73 steps): for rocprofv3 --kernel-trace --stats."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from greb_climate_model_amd import engine, workload
M = int(sys.argv[1]) if len(sys.argv) > 1 else 1
inp = workload.make_inputs(384, 192)
p = engine.params_default(); p.ipx, p.ipy = 380, 150
e = engine.Engine(inp, p, n_members=M)
t = time.perf_counter(); e.flux_correction(1); print("flux year", round(time.perf_counter() - t, 3), "s", flush=True)
e.close()

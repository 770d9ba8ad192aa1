#!/usr/bin/env python3
"""Time the fused circulation loop alone (greb_circulation_batched with a long sub-step count).
Run under rocprofv3 --kernel-trace --stats to read the kernel duration; prints wall time too.
  GREB_DEBUG_SKIP bit0/1/2 skip sub / full / chain work (results wrong; timing only)."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from greb_climate_model_amd import engine, workload, abi
engine.use_tuning_build()  # GREB_DEBUG_* knobs exist only in the -DGREB_TUNING library

nsub = int(sys.argv[1]) if len(sys.argv) > 1 else 2400
batch = int(sys.argv[2]) if len(sys.argv) > 2 else 256
strict = len(sys.argv) > 3 and sys.argv[3] == "strict"
inp = workload.make_inputs()
p = engine.params_default()
p.dt = 1800 * nsub
rng = np.random.default_rng(0)
X = np.stack([inp.tclim[i % 730] for i in range(batch)])
W = np.stack([np.exp(-inp.z_topo / 8400.0).astype(np.float32)] * batch)
U = np.stack([inp.uclim[i % 730] for i in range(batch)]); V = np.stack([inp.vclim[i % 730] for i in range(batch)])
engine.circulation(X[:2], W[:2], U[:2], V[:2], p, strict=strict)  # warm
for rep in range(2):
    t = time.perf_counter(); out = engine.circulation(X, W, U, V, p, strict=strict); dt = time.perf_counter() - t
    print(f"skip={os.environ.get('GREB_DEBUG_SKIP','0')} nsub={nsub} batch={batch} wall={dt*1e3:.1f} ms -> {dt/nsub*1e6:.3f} us/sub-step (incl. copies) finite={np.isfinite(out).all()}")

#!/usr/bin/env python3
"""384x192 A/B helper (tuning build; GREB_LIB = a variant library): one scenario year of one member (config 3) and of
the 62 perturbed-physics members without 1 800-sweep polar rows (config 5's pair-kernel regime)."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from greb_climate_model_amd import engine, ensemble, workload

engine.use_tuning_build()
if os.environ.get("GREB_LIB"):
    engine._lib_path = os.path.abspath(os.environ["GREB_LIB"])
inp = workload.make_inputs(384, 192)
p = engine.params_default(); p.ipx, p.ipy = 380, 152
for M in [int(x) for x in (sys.argv[1:] or ["1", "62"])]:
    ov = None
    if M > 1:
        o = ensemble.perturbed_physics(64, p); o = o[o[:, 3] >= 7.27e5][:M]
        ov = [dict(zip(ensemble.PERTURBED, map(float, r))) for r in o]
    pers = {None: None, '0': False, '1': True}[os.environ.get('PERSISTENT')]  # default: the engine's own trial
    e = engine.Engine(inp, p, n_members=M, overrides=ov, strict=bool(os.environ.get('STRICT')), persistent=pers)
    buf = torch.empty((M, 1, 12, 5, e.np), dtype=torch.float32, device="cuda")
    torch.cuda.synchronize(); t = time.perf_counter(); e.run(1, 680.0, monthly_dev_ptr=buf.data_ptr()); torch.cuda.synchronize(); dt = time.perf_counter() - t
    form = [c["form"] for c in e.describe().get("circulation", []) if c["members_run"] == M]
    print(f"members {M}: {M / dt:.2f} member-yr/s, {dt / (730 * 24) * 1e6:.2f} us per sub-step, {form}, finite={bool(torch.isfinite(buf).all())}", flush=True)
    e.close(); del buf

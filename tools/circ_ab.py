#!/usr/bin/env python3
"""384x192 engine: us per circulation sub-step with the call as ONE launch (greb_circ_rows.hip) against one launch per
sub-step (greb_step_rows.hip), same box, same run.  usage: circ_ab.py [members ...]"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from greb_climate_model_amd import engine, ensemble, workload
import torch
if os.environ.get("TUNING"):  # the -DGREB_TUNING library: GREB_CIRC_HEAD / GREB_CIRC_TAIL / GREB_CIRC_CHAIN_MIN ... are read
    engine.use_tuning_build()
ONLY = os.environ.get("ONLY")  # "persistent": skip the per-sub-step leg
inp = workload.make_inputs(384, 192)
p = engine.params_default(); p.ipx, p.ipy = 380, 150
for M in [int(x) for x in (sys.argv[1:] or ["1", "8", "62"])]:
    row = []
    for persistent in (False, True):
        if ONLY == "persistent" and not persistent:
            row.append((float("nan"), True)); continue
        e = engine.Engine(inp, p, n_members=M, persistent=persistent)
        e.flux_correction(1)
        buf = torch.empty((M, 1, 12, 5, e.np), dtype=torch.float32, device="cuda")
        lv = ensemble.co2_sweep(M)[:, None]
        t = time.perf_counter(); e.run(1, lv, monthly_dev_ptr=buf.data_ptr()); torch.cuda.synchronize(); dt = time.perf_counter() - t
        row.append((dt, bool(torch.isfinite(buf).all())))
        e.close(); del buf
    e = engine.Engine(inp, p, n_members=M)  # the default: the engine's own trial
    e.flux_correction(1)
    buf = torch.empty((M, 1, 12, 5, e.np), dtype=torch.float32, device="cuda")
    t = time.perf_counter(); e.run(1, ensemble.co2_sweep(M)[:, None], monthly_dev_ptr=buf.data_ptr()); torch.cuda.synchronize(); dt = time.perf_counter() - t
    chosen = [(c["members_run"], c["form"], c["trial_ms_per_3_steps"]) for c in e.describe().get("circulation", [])]
    e.close(); del buf
    (a, fa), (b, fb) = row
    print(f"   default: {dt:.3f} s/yr = {dt / 730 / 24 * 1e6:.2f} us per sub-step; trial {chosen}")
    print(f"384x192 members={M}: per sub-step launch {a:.3f} s/yr = {a / 730 / 24 * 1e6:.2f} us per sub-step ({M / a:.2f} member-yr/s); "
          f"one launch per call {b:.3f} s/yr = {b / 730 / 24 * 1e6:.2f} us per sub-step ({M / b:.2f} member-yr/s); finite {fa} {fb}", flush=True)

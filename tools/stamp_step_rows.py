#!/usr/bin/env python3
"""Where the dearest task of the 384x192 row-strip sub-step (the 232-sweep polar row) spends its cycles: s_memtime
stamps of the -DGREB_TUNING build (greb_step_rows.hip: GREB_STEP_STAMP), one member."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from greb_climate_model_amd import engine, workload
engine.use_tuning_build()
L = engine.lib()
inp = workload.make_inputs(384, 192)
p = engine.params_default(); p.ipx, p.ipy = 380, 152
M = int(sys.argv[1]) if len(sys.argv) > 1 else 1
e = engine.Engine(inp, p, n_members=M, row_strips=True)
buf = torch.empty((M, 1, 12, 5, e.np), dtype=torch.float32, device="cuda")
assert L.greb_tuning_step_stamps(None) == 0
e.run(1, 680.0, monthly_dev_ptr=buf.data_ptr()); torch.cuda.synchronize()
st = (C.c_ulonglong * 16)()
assert L.greb_tuning_step_stamps(st) == 0
s = list(st)
names = ["own row + wind landed (from loop entry)", "flux / window set-up", "diffusion chain (225 sweeps)", "advection chain (7 sweeps)", "meridional part + store"]
if M == 1:  # task 0 is the 232-sweep polar row only while every task has a SIMD to itself (step_rows_tasks)
    for n, a, b in zip(names, s[:6], s[1:6]):
        print(f"{n:45s} {b - a:8d} cycles")
    print(f"{'loop entry -> row stored':45s} {s[5] - s[0]:8d} cycles = {(s[5] - s[0]) / 2.4e3:.2f} us at 2.4 GHz")
if s[10]:
    print(f"last task of the launch (a streaming strip of {s[10]} rows), {M} member(s): window fill {s[11] - s[8]} cycles, "
          f"then {(s[9] - s[11]) / s[10]:.0f} cycles per row; whole task {(s[9] - s[8]) / 2.4e3:.2f} us at 2.4 GHz")
    n = s[10]
    print(f"   per row: issue + window advance {s[12] / n:.0f}, zonal part {s[13] / n:.0f}, meridional part + store {s[14] / n:.0f}, next winds {s[15] / n:.0f} cycles")
e.close()

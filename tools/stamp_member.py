#!/usr/bin/env python3
"""In-kernel stamps of the fused member kernel (diagnostic -DGREB_TUNING build only; the release kernel contains no
stamp code).  Prints, as medians over the members (= workgroups = CUs):
  * the in-kernel shader clock: delta s_memtime / delta s_memrealtime x 100 MHz (MI355X_MICROARCH.md, DVFS item 6)
  * cycles per model step spent in wind staging / the 24 circulation sub-steps / point physics + accumulation
  * per wave: busy cycles per sub-step (barrier release -> arrival at the next barrier) and the share of the
    sub-step it waits at the barrier -- which wave is the critical path
  python tools/stamp_member.py [members=512] [years=2]"""
import ctypes as C
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from greb_climate_model_amd import engine, ensemble, workload

engine.use_tuning_build()
if os.environ.get("GREB_LIB"):  # a variant build of the tuning library (deal experiments)
    engine._lib_path = os.path.abspath(os.environ["GREB_LIB"])
M = int(sys.argv[1]) if len(sys.argv) > 1 else 512
years = int(sys.argv[2]) if len(sys.argv) > 2 else 2
inp = workload.make_inputs()
p = engine.params_default(); p.ipx, p.ipy = 95, 38
e = engine.Engine(inp, p, n_members=M)
e.flux_correction(1)
levels = ensemble.co2_sweep(M)
mon = torch.empty((M, 1, 12, 5, e.np), dtype=torch.float32, device="cuda")
for _ in range(max(1, int(2.5 * 3800 / M / 1) // 20 if M >= 256 else 1)):  # >= 2 s of back-to-back launches first
    e.run(1, levels[:, None], monthly_dev_ptr=mon.data_ptr())
st = torch.zeros((M, 8, 8), dtype=torch.int64, device="cuda")
f = engine.lib().greb_tuning_set_stamps
f.argtypes = [C.c_void_p, C.c_void_p]
assert f(e.h, st.data_ptr()) == 0
tot = np.zeros((M, 8, 8), np.float64)
for _ in range(years):
    e.run(1, levels[:, None], monthly_dev_ptr=mon.data_ptr())
    torch.cuda.synchronize()
    tot += st.cpu().numpy().astype(np.float64)
f(e.h, None)
e.close()
tot /= years
cyc, rt, wind, circ, busy, phys, nsteps, nsub = [tot[:, :, i] for i in range(8)]
clk = np.median(cyc[:, 0] / rt[:, 0]) * 100.0  # MHz
ns, nsb = nsteps[0, 0], nsub[0, 0]
print(f"members {M}: launch = {np.median(cyc[:, 0]) / 1e6:.1f} Mcycles = {np.median(rt[:, 0]) / 1e5:.2f} ms; "
      f"in-kernel clock {clk:.0f} MHz (median over workgroups; min {np.min(cyc[:, 0] / rt[:, 0]) * 100:.0f}, max {np.max(cyc[:, 0] / rt[:, 0]) * 100:.0f})")
w0 = lambda x: np.median(x[:, 0]) / ns
print(f"per model step (wave 0): wind staging {w0(wind):.0f} cyc, circulation {w0(circ):.0f} cyc ({w0(circ) / nsb:.0f} per sub-step), "
      f"physics+accumulation {w0(phys):.0f} cyc; total {np.median(cyc[:, 0]) / ns:.0f} cyc = {np.median(cyc[:, 0]) / ns / clk:.2f} us")
sub = np.median(circ[:, 0]) / ns / nsb
print(f"sub-step = {sub:.0f} cycles = {sub / clk:.3f} us; per wave busy cycles per sub-step and barrier wait share:")
for w in range(8):
    b = np.median(busy[:, w]) / ns / nsb
    print(f"  wave {w} (SIMD {w % 4}, {'polar chains' if w == 6 else 'bulk'}): busy {b:6.0f} cyc  waits {100 * (1 - b / sub):4.1f} %")
print(f"member-years/s at this launch time: {M / (np.median(rt[:, 0]) / 1e8) / max(1, -(-M // 256)):.0f}")

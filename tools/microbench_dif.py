#!/usr/bin/env python3
"""Time the standalone batched diffusion sweep with HIP events (same method as bench.py's roofline leg)."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from greb_climate_model_amd import engine
if os.environ.get("GREB_TUNING_LIB") or os.environ.get("GREB_LIB"):  # GREB_DEBUG_* knobs exist only in the -DGREB_TUNING library; the counter passes
    engine.use_tuning_build()         # (tools/prof_rows.sh, verify_round.sh) run the release library, whose code hash they record
if os.environ.get("GREB_LIB"):
    engine._lib_path = os.path.abspath(os.environ["GREB_LIB"])  # a variant library (A/B of compile-time choices)

batch = int(sys.argv[1]) if len(sys.argv) > 1 else 16384
nx, ny = (int(sys.argv[2]), int(sys.argv[3])) if len(sys.argv) > 3 else (96, 48)
p = engine.params_default()
n = batch * nx * ny
g = torch.Generator(device="cuda").manual_seed(1)
# OFFS=a,b,c: shift the three arrays by that many floats inside their allocations (does the relative alignment of the
# streams matter?  measured: no)
offs = [int(x) for x in os.environ.get("OFFS", "0,0,0").split(",")]
T1 = (250.0 + 50.0 * torch.rand(n + offs[0], device="cuda", generator=g))[offs[0]:]
wz = (0.3 + 0.7 * torch.rand(n + offs[1], device="cuda", generator=g))[offs[1]:]
dX = torch.empty(n + offs[2], device="cuda")[offs[2]:]
st = torch.cuda.current_stream()
for strict in (False, True):
    # warm-up: after an idle second the first ~10 ms of back-to-back launches run through a power-management transient
    # (tools/launch_spread.py: launches 10-60 up to 1.35 x the settled time); time the settled state
    warm = int(os.environ.get("WARM", "150"))
    engine.diffusion_dev(p, nx, ny, batch, T1.data_ptr(), wz.data_ptr(), dX.data_ptr(), strict, warm, st.cuda_stream)
    torch.cuda.synchronize()
    reps = []
    for _ in range(int(os.environ.get("REPS", "5"))):  # each repetition times 20 back-to-back launches
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(st)
        engine.diffusion_dev(p, nx, ny, batch, T1.data_ptr(), wz.data_ptr(), dX.data_ptr(), strict, 20, st.cuda_stream)
        e1.record(st); torch.cuda.synchronize()
        reps.append(e0.elapsed_time(e1) / 20)
    ms = float(np.median(reps))
    if os.environ.get("SHOW_REPS"):
        print("   reps:", " ".join(f"{r:.4f}" for r in reps))
    print(f"diffusion {nx}x{ny} batch={batch} strict={strict} skip={os.environ.get('GREB_DEBUG_SKIP','0')}: {ms:.4f} ms/sweep (min {min(reps):.4f} max {max(reps):.4f}) -> {12.0*n/ms/1e6:.1f} GB/s algorithmic")
a = torch.empty(n, device="cuda")
for _ in range(3): a.copy_(T1)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record(st)
for _ in range(10): a.copy_(T1)
e1.record(st); torch.cuda.synchronize()
print(f"torch copy: {10*2*n*4/(e0.elapsed_time(e1)*1e-3)/1e9:.1f} GB/s")

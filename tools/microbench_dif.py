#!/usr/bin/env python3
"""Time the standalone batched diffusion sweep with HIP events (same method as bench.py's roofline leg)."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from greb_climate_model_amd import engine
engine.use_tuning_build()  # GREB_DEBUG_* knobs exist only in the -DGREB_TUNING library

batch = int(sys.argv[1]) if len(sys.argv) > 1 else 16384
nx, ny = (int(sys.argv[2]), int(sys.argv[3])) if len(sys.argv) > 3 else (96, 48)
p = engine.params_default()
n = batch * nx * ny
g = torch.Generator(device="cuda").manual_seed(1)
T1 = 250.0 + 50.0 * torch.rand(n, device="cuda", generator=g)
wz = 0.3 + 0.7 * torch.rand(n, device="cuda", generator=g)
dX = torch.empty(n, device="cuda")
st = torch.cuda.current_stream()
for strict in (False, True):
    engine.diffusion_dev(p, nx, ny, batch, T1.data_ptr(), wz.data_ptr(), dX.data_ptr(), strict, 3, st.cuda_stream)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(st)
    engine.diffusion_dev(p, nx, ny, batch, T1.data_ptr(), wz.data_ptr(), dX.data_ptr(), strict, 20, st.cuda_stream)
    e1.record(st); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 20
    print(f"diffusion {nx}x{ny} batch={batch} strict={strict} skip={os.environ.get('GREB_DEBUG_SKIP','0')}: {ms:.4f} ms/sweep -> {12.0*n/ms/1e6:.1f} GB/s algorithmic")
a = torch.empty(n, device="cuda")
for _ in range(3): a.copy_(T1)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record(st)
for _ in range(10): a.copy_(T1)
e1.record(st); torch.cuda.synchronize()
print(f"torch copy: {10*2*n*4/(e0.elapsed_time(e1)*1e-3)/1e9:.1f} GB/s")

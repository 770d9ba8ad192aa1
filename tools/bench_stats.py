#!/usr/bin/env python3
"""Ensemble-statistics kernels at the benchmark's shape: 512 members x one year of monthly means
(12 x 5 x 4608 = 276 480 elements, 566 MB): achieved HBM GB/s of the moments pass, time of the quantile pass."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from greb_climate_model_amd import ensemble

M = int(sys.argv[1]) if len(sys.argv) > 1 else 512
n = 12 * 5 * 4608
x = 280.0 + 10.0 * torch.randn((M, n), dtype=torch.float32, device="cuda")
def timed(fn, reps=10):
    fn(); torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / reps
t = timed(lambda: ensemble.local_moments(x))
byt = M * n * 4 + n * (8 + 8 + 4 + 4) * 2  # read the members once; outputs are zero-filled by torch, then written
print(f"moments: {M} members x {n} elements: {t:.3f} ms -> {M * n * 4 / t / 1e6:.0f} GB/s of member data "
      f"({M * n * 4 / t / 1e6 / 8000:.2f} of 8 TB/s; incl. output init {byt / t / 1e6:.0f} GB/s)")
tq = timed(lambda: ensemble.ensemble_quantiles(x, [0.05, 0.5, 0.95]), reps=3)
print(f"quantiles (3 probabilities): {tq:.2f} ms")
tt = timed(lambda: (x.double().sum(0), (x.double() ** 2).sum(0), x.min(0), x.max(0)), reps=3)
print(f"same moments with torch ops: {tt:.2f} ms")

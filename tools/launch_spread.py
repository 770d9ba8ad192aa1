#!/usr/bin/env python3
"""Per-launch times of the batched diffusion sweep from an idle GPU: where does the launch-to-launch spread of the
roofline line come from?  Three starts, each after 1 s of idle: (a) straight away, (b) after 60 ms of an arithmetic-only
kernel burst (fp32 matmuls), (c) after 60 ms of a memory-only burst (copies).  Prints every launch time.
  python tools/launch_spread.py [nx ny batch]"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from greb_climate_model_amd import engine

nx, ny, batch = (int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])) if len(sys.argv) > 3 else (96, 48, 16384)
p = engine.params_default()
n = batch * nx * ny
g = torch.Generator(device="cuda").manual_seed(1)
T1 = 250.0 + 50.0 * torch.rand(n, device="cuda", generator=g)
wz = 0.3 + 0.7 * torch.rand(n, device="cuda", generator=g)
dX = torch.empty(n, device="cuda")
A = torch.rand(4096, 4096, device="cuda"); B = torch.rand(4096, 4096, device="cuda"); Cm = torch.empty_like(A)
big = torch.empty(n, device="cuda")
st = torch.cuda.current_stream()
N = 120


def timed():
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(N + 1)]
    ev[0].record(st)
    for i in range(N):
        engine.diffusion_dev(p, nx, ny, batch, T1.data_ptr(), wz.data_ptr(), dX.data_ptr(), False, 1, st.cuda_stream)
        ev[i + 1].record(st)
    torch.cuda.synchronize()
    return np.array([ev[i].elapsed_time(ev[i + 1]) for i in range(N)])


def burst(kind):
    t0 = time.perf_counter()
    while time.perf_counter() - t0 < 0.06:
        for _ in range(4):
            if kind == "alu":
                torch.mm(A, B, out=Cm)
            else:
                big.copy_(T1)
        torch.cuda.synchronize()


engine.diffusion_dev(p, nx, ny, batch, T1.data_ptr(), wz.data_ptr(), dX.data_ptr(), False, 3, st.cuda_stream)
torch.mm(A, B, out=Cm); big.copy_(T1)
torch.cuda.synchronize()
for label, pre in (("idle 1 s", None), ("idle 1 s + 60 ms arithmetic burst", "alu"), ("idle 1 s + 60 ms copy burst", "mem"),
                   ("no idle (straight after the previous block)", "none")):
    if pre != "none":
        time.sleep(1.0)
    if pre in ("alu", "mem"):
        burst(pre)
    ms = timed()
    cum = np.cumsum(ms)
    print(f"== {nx}x{ny} batch {batch}, start: {label}")
    print("   launches  1-10 :", " ".join(f"{x:.4f}" for x in ms[:10]))
    print("   launches 11-20 :", " ".join(f"{x:.4f}" for x in ms[10:20]))
    print(f"   mean of launches 1-20 {ms[:20].mean():.4f}  21-60 {ms[20:60].mean():.4f}  61-120 {ms[60:].mean():.4f}  median of all {np.median(ms):.4f}  min {ms.min():.4f}")
    settled = np.nonzero(ms < 1.02 * np.median(ms[60:]))[0]
    print(f"   first launch within 2 % of the settled time: #{settled[0] + 1 if len(settled) else -1} (after {cum[settled[0]] if len(settled) else -1:.1f} ms of launches)")

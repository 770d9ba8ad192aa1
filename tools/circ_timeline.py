#!/usr/bin/env python3
"""The one-launch circulation call at 384x192 (greb_circ_rows.hip), tuning build: per task the length of the last launch
and the time it spent waiting for its neighbours / draining its stores, summed over `steps` launches.
python tools/circ_timeline.py [members] [model steps]"""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
os.environ.setdefault("GREB_DEBUG_NSTEPS", sys.argv[2] if len(sys.argv) > 2 else "40")
from greb_climate_model_amd import engine, ensemble, workload
engine.use_tuning_build()
L = engine.lib()
M = int(sys.argv[1]) if len(sys.argv) > 1 else 62
steps = int(os.environ["GREB_DEBUG_NSTEPS"])
inp = workload.make_inputs(384, 192)
p = engine.params_default(); p.ipx, p.ipy = 380, 152
ov = None
if M > 1:
    o = ensemble.perturbed_physics(64, p); o = o[o[:, 3] >= 7.27e5][:M]
    ov = [dict(zip(ensemble.PERTURBED, map(float, r))) for r in o]
kappa = np.array([r["kappa"] for r in ov], np.float32) if ov else None
field, k0, k1, chain, dep = engine.circulation_launch_plan(p, 384, 192, M, kappa, 2048)
n = len(field)
chain = chain.astype(bool)
e = engine.Engine(inp, p, n_members=M, overrides=ov, persistent=True)  # (no trial: every launch of the run is the one-launch form)
buf = torch.empty((M, 1, 12, 5, e.np), dtype=torch.float32, device="cuda")
L.greb_tuning_circ_timeline.argtypes = [C.c_void_p, C.c_int]
assert L.greb_tuning_circ_timeline(None, n) == 0
torch.cuda.synchronize()
import time
t = time.perf_counter(); e.run(1, 680.0, monthly_dev_ptr=buf.data_ptr()); torch.cuda.synchronize(); dt = time.perf_counter() - t
out = (C.c_ulonglong * (5 * n))()
assert L.greb_tuning_circ_timeline(out, n) == 0
a = np.array(out[:2 * n], np.int64).reshape(n, 2)
if (a == 0).any() or (a[:, 1] < a[:, 0]).any():
    sys.exit(f"incomplete stamps: {int((a == 0).sum())} zero entries of {a.size} -- is the tuning library current?")
hw = np.array(out[2 * n:3 * n], np.uint64)
wait = np.array(out[3 * n:4 * n], np.int64) / 100.0 / steps / 24   # us per sub-step
drain = np.array(out[4 * n:5 * n], np.int64) / 100.0 / steps / 24
t0 = a[:, 0].min()
st, en = (a[:, 0] - t0) / 100.0, (a[:, 1] - t0) / 100.0
print(f"{M} members, {n} tasks ({int(chain.sum())} chain tasks), {steps} model steps in {dt:.3f} s = {dt / steps / 24 * 1e6:.2f} us per sub-step incl. the point-physics launch")
print(f"last launch: {en.max():.1f} us = {en.max() / 24:.2f} us per sub-step; starts within {st.max():.1f} us")
rows = k1 - k0
simd = ((hw >> np.uint64(4)) & np.uint64(3)).astype(int); cu = ((hw >> np.uint64(8)) & np.uint64(15)).astype(int)
se = ((hw >> np.uint64(13)) & np.uint64(7)).astype(int); xcc = ((hw >> np.uint64(32)) & np.uint64(15)).astype(int)
key = ((xcc * 8 + se) * 16 + cu) * 4 + simd
u, cnt = np.unique(key, return_counts=True)
print(f"SIMDs used {len(u)}; tasks per SIMD min {cnt.min()} max {cnt.max()}")
shared = np.isin(key, u[cnt > 1])
def line(name, sel):
    if sel.sum() == 0: return
    print(f"{name:>34s}: {int(sel.sum()):5d} tasks, rows {rows[sel].mean():5.1f}; per sub-step: waiting {wait[sel].mean():6.2f} us (min {wait[sel].min():.2f}, max {wait[sel].max():.2f}), draining {drain[sel].mean():5.2f} us")
big = chain & np.isin(k0, (1, 190))
line("chain tasks rows 1/190", big)
line("other chain tasks", chain & ~big)
line("strips alone on their SIMD", ~chain & ~shared)
line("strips sharing a SIMD", ~chain & shared)
line("chain tasks sharing a SIMD", chain & shared)
dur = (en - st) / 24
line("all", np.ones(n, bool))
pairs = {}
for j in range(n): pairs.setdefault(key[j], []).append(j)
kinds = {}
for v in pairs.values():
    kk = tuple(sorted(("C" if chain[j] and k0[j] in (1, 190) else ("c" if chain[j] else "s")) for j in v))
    kinds.setdefault(kk, []).append(v)
both = [v for v in pairs.values() if len(v) == 2]
dist = np.array([abs(v[0] - v[1]) for v in both]) if both else np.zeros(0, int)
vals, cnts = np.unique(dist, return_counts=True)
n_simd = len(u) if len(u) >= 1024 else 1024
meant = sum(1 for v in both if max(v) == n_simd + 4 * (min(v) // 4) + (min(v) % 4 + 3) % 4)
print(f"PREMISE task 4c + w shares its SIMD with task {n_simd} + 4c + (w + 3) mod 4: {meant} of {len(both)} SIMDs that hold two tasks; "
      "index distances (distance: count): " + " ".join(f"{a}:{b}" for a, b in sorted(zip(vals, cnts), key=lambda x: -x[1])[:6]))
same_cu = sum(1 for j in range(0, n - n % 4, 4) if len({key[j + q] // 4 for q in range(4)}) == 1 and len({key[j + q] % 4 for q in range(4)}) == 4)
print(f"workgroups whose four tasks sit on the four SIMDs of one CU: {same_cu} of {n // 4}")
print("(for chain tasks 'draining' is the time in the zonal chains, their publish included)")
print("what shares a SIMD (C: chain task rows 1/190, c: other chain task, s: strip): mean wait per sub-step of its tasks")
for kk, vs in sorted(kinds.items()):
    w = np.mean([wait[j] for v in vs for j in v]); wmin = np.min([min(wait[j] for j in v) for v in vs])
    print(f"   {'+'.join(kk):>5s}: {len(vs):4d} SIMDs, wait mean {w:5.2f} us, least-waiting task {wmin:5.2f} us")
i = np.argsort(wait)[:8]
print("the tasks that wait least (the critical ones):")
for j in i:
    print(f"   task {j:5d} field {field[j]:4d} rows {k0[j]:3d}..{k1[j]:3d} {'chain' if chain[j] else 'strip'} {'shared SIMD' if shared[j] else 'alone'}: waits {wait[j]:.2f} us, drains {drain[j]:.2f} us per sub-step")
e.close()

#!/usr/bin/env python3
"""Per-task start / end times of one launch of the 384x192 sub-step kernel (tuning build, s_memrealtime stamps):
how full the wavefront slots are over the launch and which strips end it.  python tools/step_timeline.py [members]"""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from greb_climate_model_amd import engine, ensemble, workload
engine.use_tuning_build()
L = engine.lib()
M = int(sys.argv[1]) if len(sys.argv) > 1 else 62
inp = workload.make_inputs(384, 192)
p = engine.params_default(); p.ipx, p.ipy = 380, 152
ov = None
if M > 1:
    o = ensemble.perturbed_physics(64, p); o = o[o[:, 3] >= 7.27e5][:M]
    ov = [dict(zip(ensemble.PERTURBED, map(float, r))) for r in o]
kappa = np.array([r["kappa"] for r in ov], np.float32) if ov else None
field, k0, k1 = engine.substep_launch_order(p, 384, 192, M, kappa)
n = len(field)
e = engine.Engine(inp, p, n_members=M, overrides=ov, persistent=False)  # (the per-sub-step kernel is what this tool looks at)
buf = torch.empty((M, 1, 12, 5, e.np), dtype=torch.float32, device="cuda")
L.greb_tuning_step_timeline.argtypes = [C.c_void_p, C.c_int]
assert L.greb_tuning_step_timeline(None, n) == 0
e.run(1, 680.0, monthly_dev_ptr=buf.data_ptr()); torch.cuda.synchronize()
out = (C.c_ulonglong * (3 * n))()
assert L.greb_tuning_step_timeline(out, n) == 0
a = np.array(out[:2 * n], np.int64).reshape(n, 2)
if (a == 0).any() or (a[:, 1] < a[:, 0]).any() or a.max() - a.min() > 100 * 1000 * 1000:  # (a launch is not a second long)
    sys.exit(f"incomplete stamps: {int((a == 0).sum())} zero entries of {a.size} -- is the tuning library current?")
hw = np.array(out[2 * n:], np.uint64)
t0 = a[:, 0].min()
st, en = (a[:, 0] - t0) / 100.0, (a[:, 1] - t0) / 100.0  # us
rows = k1 - k0
print(f"{M} members, {n} tasks: launch = {en.max():.1f} us from the first task's start")
edges = np.arange(0, en.max() + 2, 2.0)
print("wavefronts resident at t (us):", " ".join(f"{int(t)}:{int(((st <= t) & (en > t)).sum())}" for t in edges))
print("starts: within 1 us %d, 1-5 us %d, later %d (latest %.1f us)" % ((st < 1).sum(), ((st >= 1) & (st < 5)).sum(), (st >= 5).sum(), st.max()))
order = np.argsort(-en)
print("the ten tasks that end last:")
for i in order[:10]:
    print(f"   task {i:5d} field {field[i]:4d} rows {k0[i]:3d}..{k1[i]:3d}  start {st[i]:6.1f} us  end {en[i]:6.1f} us  ({en[i] - st[i]:.1f} us)")
dur = en - st
stream = (k0 >= 18) & (k1 <= 174)
print(f"streaming strips: {stream.sum()} tasks, {rows[stream].mean():.1f} rows, duration mean {dur[stream].mean():.1f} us (min {dur[stream].min():.1f}, max {dur[stream].max():.1f}); per row {1e3 * dur[stream].sum() / rows[stream].sum():.0f} ns")
cap = ~stream
print(f"polar strips: {cap.sum()} tasks, duration mean {dur[cap].mean():.1f} us (max {dur[cap].max():.1f})")
# the same SIMD: bits of HW_ID: wave 3:0, simd 5:4, cu 11:8, sh 12, se 15:13 (gfx9)
simd = ((hw >> np.uint64(4)) & np.uint64(3)).astype(int); cu = ((hw >> np.uint64(8)) & np.uint64(15)).astype(int)
se = ((hw >> np.uint64(13)) & np.uint64(7)).astype(int); xcc = ((hw >> np.uint64(32)) & np.uint64(15)).astype(int)
key = ((xcc * 8 + se) * 16 + cu) * 4 + simd
u, cnt = np.unique(key, return_counts=True)
print(f"SIMDs used {len(u)}; tasks per SIMD min {cnt.min()} max {cnt.max()}; CUs used {len(np.unique(key // 4))}")
print("placement of the first 40 tasks (task: xcc/se/cu/simd):", " ".join(f"{i}:{xcc[i]}/{se[i]}/{cu[i]}/{simd[i]}" for i in range(40)))
pairs = {}
for i in range(n): pairs.setdefault(key[i], []).append(i)
both = [v for v in pairs.values() if len(v) == 2]
d = np.array([abs(v[0] - v[1]) for v in both])
vals, c = np.unique(d, return_counts=True)
n_simd = 4 * int(engine.device_info(0)["cus"])
print(f"PREMISE tasks i and i + {n_simd} share a SIMD: {int((d == n_simd).sum())} of {len(both)} SIMDs that hold two tasks")
print("index distance of the two tasks that share a SIMD (distance: count):", " ".join(f"{a}:{b}" for a, b in sorted(zip(vals, c), key=lambda x: -x[1])[:12]))
capk = {i for i in range(n) if not stream[i]}
print("SIMDs holding two polar strips:", sum(1 for v in both if v[0] in capk and v[1] in capk), " one polar + one streaming:", sum(1 for v in both if (v[0] in capk) != (v[1] in capk)), " two streaming:", sum(1 for v in both if v[0] not in capk and v[1] not in capk))
for name, sel in (("paired with another polar strip", [v for v in both if v[0] in capk and v[1] in capk]), ("paired with a streaming strip", [v for v in both if (v[0] in capk) != (v[1] in capk)])):
    ds = [dur[i] for v in sel for i in v if i in capk]
    if ds: print(f"polar strips {name}: mean {np.mean(ds):.1f} us, max {np.max(ds):.1f} us")
e.close()

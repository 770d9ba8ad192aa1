#!/usr/bin/env python3
"""Run one of BASELINE.json's configurations on the MI355X engine and print a JSON summary.

  python tools/run_config.py 2            single MI355X, 1 member, default 2xCO2 3+50 yr, vs the reference's
                                          golden statistics (tests/golden/run_default_g96.npz)
  python tools/run_config.py 3 [--years Y]  384x192 (bilinear-upsampled inputs), 1 member, 3+Y yr (default 50)
  python tools/run_config.py 4 [--years Y]  8 CO2 levels 280..1120 ppm, 3+Y yr (default 100); one process per GPU
                                          under torchrun (members dealt to ranks, RCCL gather), or all 8 on one GPU
  python tools/run_config.py 5 [--years Y] [--members M]  64 perturbed-physics members (da_ice, a_no_ice, a_cloud,
                                          kappa +-10 %, SplitMix64 seed 20261004), 384x192, 3+Y yr (default 50)
Config 1 is the reference Fortran on the CPU (oracle/_ref/greb_ref; bench.py times it as cpu_baseline).
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("config", type=int, choices=[2, 3, 4, 5])
    ap.add_argument("--years", type=int, default=None)
    ap.add_argument("--members", type=int, default=None)
    ap.add_argument("--strict", action="store_true")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist
    from greb_climate_model_amd import abi, engine, ensemble, workload

    world = int(os.environ.get("WORLD_SIZE", "1")); rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    # rehearsal knobs for a one-GPU box (N ranks sharing the card over gloo), as in bench.py; never set on a real node
    backend = os.environ.get("GREB_BENCH_BACKEND", "nccl")
    if "GREB_BENCH_DEVICE" in os.environ:
        local_rank = int(os.environ["GREB_BENCH_DEVICE"])
    torch.cuda.set_device(local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend)

    cfg = args.config
    g384 = cfg in (3, 5)
    inp = workload.make_inputs(384, 192) if g384 else workload.make_inputs()
    p = engine.params_default()
    p.ipx, p.ipy = (380, 150) if g384 else (95, 38)
    tf = 3
    years = args.years or {2: 50, 3: 50, 4: 100, 5: 50}[cfg]
    n_total = {2: 1, 3: 1, 4: 8, 5: args.members or 64}[cfg]
    ids = ensemble.partition(n_total, world, rank)
    overrides, co2 = None, np.full((len(ids), years), 680.0, np.float32)
    if cfg == 4:
        co2 = np.repeat(ensemble.co2_sweep(8)[ids][:, None], years, 1)
    if cfg == 5:
        ov = ensemble.perturbed_physics(n_total, p)
        overrides = [{k: float(ov[g, j]) for j, k in enumerate(ensemble.PERTURBED)} for g in ids]
    out = {"config": cfg, "grid": [inp.nx, inp.ny], "members_total": n_total, "members_this_rank": len(ids),
           "time_flux": tf, "time_scnr": years, "n_gpus": world, "arithmetic": "strict" if args.strict else "fast"}
    t0 = time.perf_counter()
    # Members of one engine advance in lock step; members whose polar chains are several times longer than the others'
    # (config 5: kappa < 7.27e5 at 384x192, 1 800 sweeps) get an engine of their own, run beside the first
    kap = [(o or {}).get("kappa", p.kappa) for o in overrides] if overrides else [p.kappa] * len(ids)
    groups = ensemble.latency_groups(kap, inp.nx, inp.ny)
    co2a = np.asarray(co2, np.float32)  # [members of this rank][years]
    engines = [engine.Engine(inp, p, n_members=len(g), overrides=[overrides[i] for i in g] if overrides else None,
                             device=local_rank, strict=args.strict) for g in groups]
    yfs = ensemble.run_beside([lambda e=e: e.flux_correction(tf) for e in engines])
    t1 = time.perf_counter()
    np_ = engines[0].np
    mon_dev = torch.empty((len(ids), years, 12, 5, np_), dtype=torch.float32, device="cuda")
    parts = [mon_dev if len(groups) == 1 else torch.empty((len(g), years, 12, 5, np_), dtype=torch.float32, device="cuda") for g in groups]
    yrs = ensemble.run_beside([lambda e=e, b=b, g=g: e.run(years, co2a[g], monthly_dev_ptr=b.data_ptr())[1]
                               for e, b, g in zip(engines, parts, groups)])
    yf = np.empty((len(ids),) + yfs[0].shape[1:], yfs[0].dtype); yr = np.empty((len(ids),) + yrs[0].shape[1:], yrs[0].dtype)
    for g, a, b, part in zip(groups, yfs, yrs, parts):
        yf[g] = a; yr[g] = b
        if len(groups) > 1:
            mon_dev[torch.as_tensor(g, device="cuda")] = part
    del parts
    out["engines"] = [int(len(g)) for g in groups]
    gathered = ensemble.gather_monthly(mon_dev, n_total) if world > 1 else mon_dev
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    e = engines[0]
    out.update(flux_phase_s=round(t1 - t0, 3), scenario_s=round(t2 - t1, 3),
               member_years_per_s=round(n_total * (tf + years) / (t2 - t0), 2),
               scenario_member_years_per_s=round(n_total * years / (t2 - t1), 2))
    if rank == 0:
        m = gathered.reshape(n_total, years * 12, 5, inp.ny, inp.nx)
        out["finite"] = bool(torch.isfinite(m).all())
        out["last_year_global_mean_tsurf_C"] = [round(float(x), 4) for x in yr[:, -1, 0]]
        out["last_december_mean"] = {k: round(float(m[0, -1, i].double().mean()), 6)
                                     for i, k in enumerate(("Tsurf", "Tair", "Tocean", "q", "albedo"))}
        if cfg == 2:  # compare with the reference Fortran's statistics for the same namelist
            with np.load(os.path.join(ROOT, "tests", "golden", "run_default_g96.npz")) as z:
                stats, sel, months, yref = z["stats"], z["monthly_sel"], z["months"], z["yearly"]
            mine = m[0].double().mean((2, 3)).cpu().numpy()
            out["max_abs_diff_of_monthly_field_means_vs_reference"] = [float(np.abs(mine[:, i] - stats[:, i, 0]).max()) for i in range(5)]
            full = m[0][[int(k) - 1 for k in months]].cpu().numpy().astype(np.float64)
            out["rms_vs_reference_months_1_12_300_600"] = [float(np.sqrt(((full[:, i] - sel[:, i]) ** 2).mean())) for i in range(5)]
            out["max_abs_diff_yearly_console_values"] = float(np.abs(np.concatenate([yf[0], yr[0]]) - yref).max())
        print(json.dumps(out))
    for e in engines:
        e.close()
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()

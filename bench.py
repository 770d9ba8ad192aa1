#!/usr/bin/env python3
"""bench.py -- simulated-years/s of the GREB time-integration engine on MI355X.

  python bench.py --gpus N --steps K --warmup W [--members M] [--strict]

A "step" is ONE SIMULATED MODEL-YEAR (730 model steps = 35 040 diffusion + 35 040 advection
sweeps per member) for every ensemble member resident on the GPU.  Members are independent
96x48 GREB integrations that differ in their CO2 level (BASELINE config 4's sweep, 280-1120
ppm), M per GPU (default 512: the single grid occupies one CU, so a GPU is only full with
hundreds of members -- SURVEY.md 0.6).  value = members x years / wall over all ranks
(weak scaling: M per GPU fixed).  At N>1 the monthly means of every member are gathered to
rank 0 over RCCL inside the timed region (north_star's only collective).

Extra objects on the JSON line:
  roofline      the standalone batched diffusion kernel (the kernel BASELINE's second metric is
                defined on): algorithmic 12 B/point/sweep over its HIP-event-timed launches
  cpu_baseline  the reference Fortran itself (oracle/_ref/greb_ref, built from
                /root/reference in the build container; "reference") or, if that binary is
                absent, the C restatement ("port"), 1 core, bounded sample of the same workload
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=4)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--members", type=int, default=512, help="ensemble members per GPU")
    ap.add_argument("--root-members", type=int, default=None,
                    help="N > 1: members rank 0 integrates (default: --members).  Rank 0 also hosts the receive side of every "
                         "gather; if the per-rank report shows it as the straggler, give it a smaller share")
    ap.add_argument("--strict", action="store_true", help="reference operation order (bit-exact stencils)")
    ap.add_argument("--no-cpu", action="store_true")
    ap.add_argument("--no-roofline", action="store_true")
    ap.add_argument("--no-g384", action="store_true", help="skip the 384x192 object (BASELINE configs 3 and 5)")
    ap.add_argument("--roofline-batch", type=int, default=16384)
    ap.add_argument("--launch-timeout", type=float, default=3000.0,
                    help="seconds after which a self-launched multi-rank run is stopped (exit 124)")
    ap.add_argument("--dry-launch", action="store_true",
                    help="bring the ranks up (gloo, no GPU call), print the member partition, exit")
    return ap.parse_args(argv)


def launch_ranks(args, argv):
    """`python bench.py --gpus N` run directly: this parent starts N rank processes (one per GPU), relays rank 0's
    JSON line and fails if any rank fails.  It never touches the GPU (no torch import at all), and it starts the
    ranks as CHILD processes -- a process that has initialised the GPU must never be re-exec'ed.  The reference's
    own ensemble convention is the same: N processes with N ens_ids (src/greb.f90:153,1064-1068)."""
    import socket
    import subprocess
    import threading
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    procs = []
    for r in range(args.gpus):
        # HSA_ENABLE_IPC_MODE_LEGACY=0: the hosts of this pool support only dmabuf IPC; without it RCCL's peer-to-peer
        # set-up (and any CUDA-tensor sharing across processes) fails with `hipIpcGetMemHandle: invalid argument`.  The
        # image exports it already; it is repeated here so that a launch from a scrubbed environment still works.
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(args.gpus), LOCAL_WORLD_SIZE=str(args.gpus),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + argv, env=env,
                                      stdout=subprocess.PIPE if r == 0 else sys.stderr, text=True))
    # rank 0's stdout is drained while it runs: a child that fills the pipe buffer (library warnings, a long JSON line)
    # would otherwise block in write() and the other ranks would wait for it in a collective for ever
    chunks = []
    reader = threading.Thread(target=lambda: chunks.extend(iter(procs[0].stdout.readline, "")), daemon=True)
    reader.start()
    rc = 0
    deadline = time.monotonic() + args.launch_timeout
    try:
        pending = set(range(args.gpus))
        while pending:
            for r in sorted(pending):
                code = procs[r].poll()
                if code is None:
                    continue
                pending.discard(r)
                if code != 0 and rc == 0:  # one rank failed: the others would wait in a collective for ever
                    rc = code if code > 0 else 1
                    print(f"bench.py: rank {r} exited with {code}; stopping the other ranks", file=sys.stderr)
                    for q in pending:
                        procs[q].terminate()
            if pending and time.monotonic() > deadline:
                print(f"bench.py: the ranks did not finish within {args.launch_timeout:.0f} s; stopping them", file=sys.stderr)
                rc = rc or 124
                for q in pending:
                    procs[q].terminate()
                deadline = float("inf")
            if pending:
                try:
                    procs[min(pending)].wait(timeout=0.2)
                except subprocess.TimeoutExpired:
                    pass
    finally:
        for p in procs:
            if p.poll() is None:
                p.kill()
    reader.join(timeout=5)
    out0 = "".join(chunks)
    sys.stdout.write(out0)
    sys.stdout.flush()
    if rc == 0 and not any(line.startswith("{") for line in out0.splitlines()):
        print("bench.py: rank 0 printed no JSON line", file=sys.stderr)
        rc = 1
    return rc


def dry_launch(world, rank, members, root_members=None):
    """--dry-launch: rendezvous over gloo, report how the ensemble is dealt to the ranks and push a toy monthly record
    (each member's own CO2 level) through the timed run's gather path and its order check; no GPU call."""
    import torch
    import torch.distributed as dist
    from greb_climate_model_amd import ensemble
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    if world > 1:
        dist.init_process_group("gloo", rank=rank, world_size=world)
    R = members if root_members is None else root_members
    total = R + (world - 1) * members
    ids = ensemble.partition_root(members, R, world, rank)
    levels = ensemble.co2_sweep(total)
    t_start = time.perf_counter()
    mine = {"rank": rank, "local_rank": int(os.environ.get("LOCAL_RANK", "0")), "first": int(ids[0]), "count": int(len(ids)),
            "co2_first": float(levels[ids[0]])}
    parts, check = [mine], {"ranks_seen": 1, "backend": "none", **ensemble.gather_order_check(levels)}
    if world > 1:
        parts = [None] * world
        dist.all_gather_object(parts, mine)
        g = ensemble.MonthlyGather(members, 1, (1, 1, 4), torch.float32, "cpu")
        toy = torch.zeros((members, 1, 1, 4), dtype=torch.float32)  # (rank 0's block is padded to the common size)
        toy[: len(ids)] = torch.tensor(levels[ids], dtype=torch.float32).reshape(-1, 1, 1, 1)
        g.submit(0, toy)
        t_wait = time.perf_counter()
        got = g.finish()
        t_wait = time.perf_counter() - t_wait
        check = {"ranks_seen": ensemble.ranks_seen("cpu"), "backend": dist.get_backend()}
        check["per_rank_s"] = ensemble.rank_report({"total": time.perf_counter() - t_start, "gather_wait": t_wait, "integrate": 0.0})
        if rank == 0:
            valid = gathered_valid(got[:, 0].double().mean(dim=(1, 2, 3)).numpy(), members, R)
            check.update(ensemble.gather_order_check(valid, block=members if R == members else 0))
        dist.destroy_process_group()
    if rank == 0:
        print(json.dumps({"dry_launch": True, "n_gpus": world, "members_per_gpu": members, "root_members": R, "members_total": total,
                          "partition": parts, **check}))


def gathered_valid(per_member, members, root_members):
    """Rank 0's block of the gather buffer is padded to `members`; drop the padding (global member order is kept)."""
    return np.concatenate([per_member[:root_members], per_member[members:]])


def main():
    argv = sys.argv[1:]
    args = parse_args(argv)
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        sys.exit(launch_ranks(args, argv))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus != world:
        print(f"bench.py: --gpus {args.gpus} but WORLD_SIZE is {world}: refusing to measure a different GPU count "
              f"than the one asked for", file=sys.stderr)
        sys.exit(2)
    if args.dry_launch:
        return dry_launch(world, rank, args.members, args.root_members)

    import torch
    import torch.distributed as dist

    # rehearsal knobs for a one-GPU box (N ranks sharing the card over gloo); never set by the driver
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

    from greb_climate_model_amd import engine, ensemble, workload
    variant_lib = None
    if os.environ.get("GREB_BENCH_LIB"):  # A/B of a variant library (tools/build_member_variant.sh); never set by the driver
        variant_lib = engine._lib_path = os.path.abspath(os.environ["GREB_BENCH_LIB"])
    K, W, M = args.steps, args.warmup, args.members
    inp = workload.make_inputs()
    params = engine.params_default()
    params.ipx, params.ipy = 95, 38

    # ---------------- ensemble: contiguous blocks of global members in rank order, CO2 swept over 280..1120 ppm
    R = M if (args.root_members is None or world == 1) else args.root_members
    if not 1 <= R <= M:
        print(f"bench.py: --root-members {R} must be in 1..--members ({M})", file=sys.stderr)
        sys.exit(2)
    total_members = R + (world - 1) * M
    ids = ensemble.partition_root(M, R, world, rank)  # M members per rank (rank 0: R)
    levels = ensemble.co2_sweep(total_members)[ids]
    my_m = len(ids)
    eng = engine.Engine(inp, params, n_members=my_m, device=local_rank, strict=args.strict)
    eng.flux_correction(1)  # shared by all members (same physics): one member integrated, state broadcast
    np_ = eng.np
    # (N > 1 writes each year into its own contiguous [M, 1, 12, 5, np] buffer below: no [M, K, ...] tensor there)
    monthly = torch.empty((M, K, 12, 5, np_), dtype=torch.float32, device="cuda") if world == 1 else None
    gathered = None
    times = {"integrate": 0.0, "gather_wait": 0.0}

    def years_with_gather(n_years, bufs, g, clock=None):
        """n_years model years, one engine call per year; year y's gather (RCCL over xGMI) overlaps year y+1."""
        for y in range(n_years):
            t = time.perf_counter()
            eng.run(1, levels[:, None], monthly_dev_ptr=bufs[y].data_ptr())  # (rank 0 with R < M fills the first R members)
            if clock is not None:
                clock["integrate"] += time.perf_counter() - t
            g.submit(y, bufs[y][:, 0])
        t = time.perf_counter()
        out = g.finish()
        if clock is not None:
            clock["gather_wait"] += time.perf_counter() - t
        return out

    if world == 1:
        if W > 0:
            wbuf = torch.empty((M, W, 12, 5, np_), dtype=torch.float32, device="cuda")
            eng.run(W, np.repeat(levels[:, None], W, 1), monthly_dev_ptr=wbuf.data_ptr())
            del wbuf
    else:
        # the warm-up takes the timed path, gather included: the first collective of a kind creates RCCL's
        # peer-to-peer channels (seconds), which must not land in the timed years.  W = 0 still warms the
        # channels with one small gather.
        wy = max(W, 1)
        wshape = (M, 1, 12, 5, np_) if W > 0 else (1, 1, 1, 1, 8)
        wbufs = [torch.zeros(wshape, dtype=torch.float32, device="cuda") for _ in range(wy)]
        wg = ensemble.MonthlyGather(wshape[0], wy, wshape[2:], torch.float32, "cuda")
        if W > 0:
            years_with_gather(W, wbufs, wg)
        else:
            wg.submit(0, wbufs[0][:, 0]); wg.finish()
        del wbufs, wg

    def barrier():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()

    gather = ensemble.MonthlyGather(M, K, (12, 5, np_), torch.float32, "cuda") if world > 1 else None
    year_bufs = None
    if world > 1:  # each year in its own contiguous buffer (a [M, 12, 5, np] view of an [M, K, ...] tensor is not)
        year_bufs = [torch.zeros((M, 1, 12, 5, np_), dtype=torch.float32, device="cuda") for _ in range(K)]
    barrier()
    t0 = time.perf_counter()
    if world == 1:
        eng.run(K, np.repeat(levels[:, None], K, 1), monthly_dev_ptr=monthly.data_ptr())
    else:
        gathered = years_with_gather(K, year_bufs, gather, times)
    torch.cuda.synchronize()
    times["total_before_barrier"] = time.perf_counter() - t0  # this rank's own work, before it waits for the others
    barrier()
    dt = time.perf_counter() - t0
    per_rank = None
    if world > 1:
        tmax = torch.tensor([dt], dtype=torch.float64, device="cuda")
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dt = float(tmax.item())
        per_rank = ensemble.rank_report(times, device="cuda")
    value = total_members * K / dt

    # N > 1: prove what the collective did -- how many ranks it connected, and that every member landed in its own
    # slot of rank 0's tensor (the CO2 sweep grows with the global member index, and so does last year's mean Tsurf)
    multi = None
    if world > 1:
        multi = {"ranks_seen": ensemble.ranks_seen("cuda"), "backend": dist.get_backend()}
        multi["per_rank_s"] = per_rank  # integrate / gather_wait / total_before_barrier: per rank, min, max, rank of max
        multi["root_members"] = R
        if rank == 0:
            valid = gathered_valid(gathered[:, -1, :, 0].double().mean(dim=(1, 2)).cpu().numpy(), M, R)
            multi.update(ensemble.gather_order_check(valid, block=M if R == M else 0))
    # (rank 0 of an N > 1 run holds the gathered ensemble; the other ranks their own last year)
    mine = monthly if world == 1 else (gathered if gathered is not None else year_bufs[-1][:my_m])
    finite = bool(torch.isfinite(mine).all().item())
    tmean = float((monthly[:, -1, :, 0] if world == 1 else year_bufs[-1][:my_m, 0, :, 0]).mean().item())

    extra = {}
    if rank == 0:
        # single-member latency figure (one member = one CU busy), for the configs with 1 member/GPU
        e1 = engine.Engine(inp, params, n_members=1, device=local_rank, strict=args.strict)
        e1.flux_correction(1)
        t1 = time.perf_counter(); e1.run(2, 680.0); t1 = time.perf_counter() - t1
        extra["single_member_years_per_s"] = round(2 / t1, 2)
        e1.close()

    roof = None
    if rank == 0 and not args.no_roofline:
        roof = roofline_diffusion(torch, engine, params, args.roofline_batch, args.strict)
    cpu = cpu_all = delivered = g384 = None
    if rank == 0 and world == 1:
        delivered = host_delivered(torch, eng, levels, M, min(K, 4))
    if rank == 0 and world == 1 and not args.no_cpu:
        cpu = cpu_baseline(inp)
        cpu_all = cpu_baseline_all_cores(inp)

    if rank == 0 and world == 1 and not args.no_g384:
        eng.close()  # the 384x192 ensemble wants the HBM (41 GB of per-member flux corrections at 64 members)
        del monthly
        torch.cuda.empty_cache()
        g384 = g384_object(torch, engine, ensemble, workload, local_rank, args.strict)

    if rank == 0:
        out = {
            "metric": "simulated-years/sec at 96x48 grid (ensemble aggregate); diffusion HBM GB/s vs roofline",
            "value": round(value, 2), "unit": "simulated-years/s", "n_gpus": world, "steps": K, "warmup": W,
            "ms_per_step": round(1e3 * dt / K, 3), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": f"96x48 GREB ensemble, {M} members/GPU x {world} GPU" + (f" (rank 0: {R})" if R != M else "") + ", CO2 sweep 280-1120 ppm, "
                                   f"{K} scenario years after 1 flux-correction year, dt=12h, dt_crcl=0.5h",
                       "members_per_gpu": M, "arithmetic": "strict" if args.strict else "fast",
                       "gather": "rccl gather of monthly means to rank 0, one per year, overlapped with the next year" if world > 1 else "none (1 GPU)"},
            "years_per_s_per_gpu": round(value / world, 2),
            "finite": finite, "last_year_mean_tsurf_K": round(tmean, 3),
            "device": engine.device_info(local_rank),
        }
        out.update(extra)
        if variant_lib:  # NOT the product library: say so in the line itself
            out["lib"] = variant_lib
        if multi is not None:
            out.update(multi)
        # the engine itself is not HBM-bound: its algorithmic arithmetic, 77 flop per point, tracer and
        # circulation sub-step = 12.4 GFLOP per member-year at 96x48, against the fp32 vector peak (an FMA counts 2)
        gflop_my = 2 * 77.0 * 96 * 48 * 24 * 730 / 1e9
        out["engine_arithmetic"] = {"algorithmic_gflop_per_member_year": round(gflop_my, 2),
                                    "achieved_tflops_per_gpu": round(value / world * gflop_my / 1e3, 2),
                                    "peak_fp32_vector_tflops": 157.3,
                                    "frac": round(value / world * gflop_my / 1e3 / 157.3, 4),
                                    "bound": "VALU pipe: the four SIMDs of the sub-step loop are balanced to 1 % (DESIGN.md 4.1)"}
        if roof is not None:
            out["roofline"] = roof
        if delivered is not None:
            out["host_delivered"] = delivered
        if g384 is not None:
            out["g384"] = g384
        if cpu is not None:
            out["cpu_baseline"] = cpu
        if cpu_all is not None:
            out["cpu_baseline_all_cores"] = cpu_all
        print(json.dumps(out))
    eng.close()
    if world > 1:
        dist.destroy_process_group()


def timed_sweeps(torch, engine, params, nx, ny, batch, strict, bufs, launches=20, warm_ms=80.0):
    """Per-launch times (ms) of the batched diffusion sweep, HIP events on the stream it is launched on.
    Settled state: after idle time the first ~10 ms of back-to-back launches go through a power-management transient
    (tools/launch_spread.py, profiles/r03_launch_spread.txt: launches 2-9 run at the settled rate, launches 10-60 up to
    1.35 x slower, then it is over; a 60 ms burst of ANY kernel in front removes it) -- so the same launch is repeated
    for `warm_ms` first.  That transient, not the kernel, was the 15 % mean-to-best spread of round 2's roofline line."""
    T1, wz, dX = bufs
    stream = torch.cuda.current_stream()
    run = lambda k: engine.diffusion_dev(params, nx, ny, batch, T1.data_ptr(), wz.data_ptr(), dX.data_ptr(), strict, k, stream.cuda_stream)
    run(3); torch.cuda.synchronize()
    t0 = time.perf_counter()
    while (time.perf_counter() - t0) * 1e3 < warm_ms:
        run(20); torch.cuda.synchronize()
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(launches + 1)]
    ev[0].record(stream)
    for i in range(launches):
        run(1)
        ev[i + 1].record(stream)
    torch.cuda.synchronize()
    return [ev[i].elapsed_time(ev[i + 1]) for i in range(launches)]


def after_idle(torch, engine, params, nx, ny, batch, strict, bufs, launches=60):
    """The same launches straight after one second of idle: what a caller who launches from a cold start sees."""
    T1, wz, dX = bufs
    stream = torch.cuda.current_stream()
    torch.cuda.synchronize(); time.sleep(1.0)
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(launches + 1)]
    ev[0].record(stream)
    for i in range(launches):
        engine.diffusion_dev(params, nx, ny, batch, T1.data_ptr(), wz.data_ptr(), dX.data_ptr(), strict, 1, stream.cuda_stream)
        ev[i + 1].record(stream)
    torch.cuda.synchronize()
    ms = [ev[i].elapsed_time(ev[i + 1]) for i in range(launches)]
    return {"launches": launches, "mean_ms": round(float(np.mean(ms)), 4), "max_ms": round(float(np.max(ms)), 4),
            "mean_ms_launches_2_to_9": round(float(np.mean(ms[1:9])), 4), "mean_ms_launches_21_to_60": round(float(np.mean(ms[20:])), 4)}


# the committed counter records the line quotes (tools/verify_round.sh profiles writes them, each with the hash of the
# kernel code it was collected on; tests/test_profiles_cpu.py fails when the built library has other code)
TRAFFIC_FILE, G384_DIF_FILE, G384_STEP_FILE = "r04_roofline_traffic.json", "r04_g384_diffusion_pmc.json", "r04_g384_substep_pmc.json"


def pmc_record(name):
    """Counter figures of a kernel from the committed rocprofv3 --pmc passes (profiles/<name>.json, written by
    tools/pmc_rows_summary.py / tools/verify_round.sh: rocprofv3 cannot wrap itself around this process)."""
    try:
        with open(os.path.join(ROOT, "profiles", name)) as f:
            return json.load(f)
    except (OSError, ValueError):
        return None


def sweep_stats(ms, algo):
    avg = float(np.mean(ms)) * 1e-3
    return {"achieved": round(algo / avg / 1e9, 1), "peak": 8000.0, "unit": "GB/s", "frac": round(algo / avg / 1e9 / 8000.0, 4),
            "algorithmic_bytes_per_launch": int(algo), "avg_launch_ms": round(avg * 1e3, 4),
            "median_launch_ms": round(float(np.median(ms)), 4), "min_launch_ms": round(float(np.min(ms)), 4),
            "max_launch_ms": round(float(np.max(ms)), 4), "stddev_launch_ms": round(float(np.std(ms)), 5),
            "frac_at_median": round(algo / (float(np.median(ms)) * 1e-3) / 1e9 / 8000.0, 4),
            "launch_ms": [round(float(x), 4) for x in ms]}


def roofline_diffusion(torch, engine, params, batch, strict, sweeps=20):
    from greb_climate_model_amd import codesha
    """Standalone batched diffusion sweep (src/greb.f90:556-723) timed with HIP events on the
    stream it is launched on.  Algorithmic bytes = 12 B/point/field-sweep (SURVEY.md 8d):
    read T1, read wz, write dX.  batch*3*18 KB >= 0.9 GB so the 256 MB Infinity Cache cannot
    serve it."""
    nx, ny = 96, 48
    n = batch * nx * ny
    g = torch.Generator(device="cuda").manual_seed(1)
    T1 = 250.0 + 50.0 * torch.rand(n, device="cuda", generator=g)
    wz = 0.3 + 0.7 * torch.rand(n, device="cuda", generator=g)
    dX = torch.empty(n, device="cuda")
    stream = torch.cuda.current_stream()
    ms = timed_sweeps(torch, engine, params, nx, ny, batch, strict, (T1, wz, dX), sweeps)
    cold = after_idle(torch, engine, params, nx, ny, batch, strict, (T1, wz, dX))
    # measured copy bandwidth of this box as the second denominator
    a = torch.empty(n, device="cuda");
    for _ in range(30): a.copy_(T1)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(stream)
    for _ in range(10): a.copy_(T1)
    e1.record(stream); torch.cuda.synchronize()
    copy_gbs = 10 * 2 * n * 4 / (e0.elapsed_time(e1) * 1e-3) / 1e9
    algo = 12.0 * n
    out = {"kernel": "diffusion_stream_kernel<strict>" if strict else "diffusion_stream_kernel<fast>", "bound": "hbm"}
    out.update(sweep_stats(ms, algo))
    # HBM traffic per launch from the committed PMC run: FETCH_SIZE x 2 (gfx950 counts 64 B per 128-B request,
    # MI355X_MICROARCH.md) + WRITE_SIZE, scaled to this batch
    tj = pmc_record(TRAFFIC_FILE)
    out["traffic"] = int(tj["traffic_bytes_per_launch"] * batch / tj["batch"]) if tj else None
    # ... and which machine code that run profiled: the hash in the record against the kernel in the library loaded here
    out["traffic_source"] = codesha.source_check(tj, "profiles/" + TRAFFIC_FILE, engine._lib_path)
    out.update({"batch": batch, "timing": f"{sweeps} launches, each between two HIP events, after >= 80 ms of the same launch back to back",
                "after_1s_idle": cold, "measured_copy_GBps": round(copy_gbs, 1),
                "frac_of_measured_copy": round(out["achieved"] / copy_gbs, 4)})
    return out


def g384_object(torch, engine, ensemble, workload, device, strict):
    """BASELINE configs 3 and 5 at their own grid, 384x192 (inputs: the synthetic workload bilinearly refined,
    SURVEY.md C.1), one scenario year each after one flux-correction year:
      config3   1 member (any-grid engine on row strips: per model step ONE launch for the 24 sub-steps of the
                circulation call, greb_circ_rows.hip, + 1 point-physics launch -- or 24 + 1 launches, whichever the
                engine's own trial found faster); bound by the longest dependent chain (row 2: 225 + 7 sweeps per sub-step)
      config5   64 perturbed-physics members (da_ice, a_no_ice, a_cloud, kappa +-10 %, ensemble.perturbed_physics), as
                drawn, and without the members whose kappa < 7.27e5 gives the two polar rows 1 800 dependent diffusion
                sweeps per call instead of none (the reference's integer dtdff2 is 1 instead of 0 there,
                src/greb.f90:652-654 -- inherited semantics, kept; those members set the length of every launch)
      diffusion the standalone batched diffusion sweep at this grid against the same 12 B/point definition as the
                96x48 roofline line (884 736 B per field, batch 1 024 = 0.9 GB per launch): every row is sub-cycled
                here, so the kernel is bound by the chain rows' arithmetic, not by HBM"""
    import gc
    from greb_climate_model_amd import codesha
    nx, ny = 384, 192
    inp = workload.make_inputs(nx, ny)
    p = engine.params_default(); p.ipx, p.ipy = 380, 152
    out = {"grid": [nx, ny], "arithmetic": "strict" if strict else "fast"}

    def year_rate(n_members, overrides):
        e = engine.Engine(inp, p, n_members=n_members, overrides=overrides, device=device, strict=strict)
        e.flux_correction(1)
        buf = torch.empty((n_members, 1, 12, 5, e.np), dtype=torch.float32, device="cuda")
        torch.cuda.synchronize()
        t = time.perf_counter()
        e.run(1, 680.0, monthly_dev_ptr=buf.data_ptr())
        torch.cuda.synchronize()
        dt = time.perf_counter() - t
        ok = bool(torch.isfinite(buf).all().item())
        form = [c for c in e.describe().get("circulation", []) if c["members_run"] == n_members]
        e.close(); del buf
        gc.collect(); torch.cuda.empty_cache()
        return n_members / dt, dt, ok, (form[0] if form else None)

    def form_fields(form, dt):
        """us per circulation sub-step of the timed year (its point-physics launch included), and which launch form ran"""
        d = {"us_per_substep": round(dt / (730 * 24) * 1e6, 2)}
        if form:
            d.update({"circulation_form": form["form"], "tasks_per_launch": form["tasks_of_one_launch_per_call"],
                      "trial_ms_per_3_steps_substep_vs_call": form["trial_ms_per_3_steps"]})
        return d

    r, dt, ok, form = year_rate(1, None)
    chain_us = 232 * 140 / 2.4e3  # 225 + 7 dependent sweeps of ~140 cycles at ~2.4 GHz (profiles/r03_chain_rate.txt)
    out["config3_single_member"] = {"years_per_s": round(r, 3), "finite": ok, **form_fields(form, dt),
                                    "bound": f"latency of one wavefront's 232-sweep polar row: ~{chain_us:.1f} us of the {dt / (730 * 24) * 1e6:.1f} us per sub-step "
                                             "(138-147 cycles per dependent sweep, one instruction per 4.0 cycles: profiles/r03_chain_rate.txt); in the one-launch form "
                                             "that row stays in registers for the whole call (profiles/r04_circ_timeline.txt)"}
    ov = ensemble.perturbed_physics(64, p)
    as_dicts = lambda rows: [dict(zip(ensemble.PERTURBED, map(float, row))) for row in rows]
    slow = ov[:, 3] < 7.27e5
    # Members of one engine advance in lock step, so one launch is as long as the slowest member's longest chain: the
    # members with 1 800-sweep polar rows get an engine of their own, run BESIDE the others (ensemble.latency_groups)
    groups = ensemble.latency_groups(ov[:, 3], nx, ny)
    engines = [engine.Engine(inp, p, n_members=len(g), overrides=as_dicts(ov[g]), device=device, strict=strict) for g in groups]
    ensemble.run_beside([lambda e=e: e.flux_correction(1) for e in engines])
    bufs = [torch.empty((len(g), 1, 12, 5, engines[0].np), dtype=torch.float32, device="cuda") for g in groups]
    torch.cuda.synchronize()
    t = time.perf_counter()
    ensemble.run_beside([lambda e=e, b=b: e.run(1, 680.0, monthly_dev_ptr=b.data_ptr()) for e, b in zip(engines, bufs)])
    torch.cuda.synchronize()
    dt = time.perf_counter() - t
    ok = all(bool(torch.isfinite(b).all().item()) for b in bufs)
    for e in engines:
        e.close()
    del bufs, engines
    gc.collect(); torch.cuda.empty_cache()
    out["config5_64_members"] = {"member_years_per_s": round(64 / dt, 2), "finite": ok, "members_with_1800_sweep_polar_rows": int(slow.sum()),
                                 "engines": [int(len(g)) for g in groups],
                                 "bound": "latency of the 1800-sweep polar chains of the kappa < 7.27e5 members (their own engine, "
                                          "run beside the engine of the other members)"}
    keep = ov[~slow]
    r, dt, ok, form = year_rate(len(keep), as_dicts(keep))
    pj = pmc_record(G384_STEP_FILE)
    floor = f"{pj['valu_insts_per_substep'] / 1e6:.1f} M vector instructions per sub-step = {pj['issue_floor_us']:.1f} us on 1 024 perfectly balanced SIMDs ({pj['source']})" if pj else "no committed counter pass found"
    out["config5_without_those_members"] = {"members": int(len(keep)), "member_years_per_s": round(r, 2), "finite": ok, **form_fields(form, dt),
                                            "bound": "instruction issue of the row strips: one vector instruction per SIMD every 4 cycles, shared by the two wavefronts "
                                                     f"a SIMD holds (177-200 VGPRs, 19.5 KB of LDS each); {floor}"}
    # standalone diffusion sweep, HIP events on the launching stream
    batch = 1024
    n = batch * nx * ny
    g = torch.Generator(device="cuda").manual_seed(2)
    T1 = 250.0 + 50.0 * torch.rand(n, device="cuda", generator=g)
    wz = 0.3 + 0.7 * torch.rand(n, device="cuda", generator=g)
    dX = torch.empty(n, device="cuda")
    ms = timed_sweeps(torch, engine, p, nx, ny, batch, strict, (T1, wz, dX), 20)
    d = {"kernel": "dif_rows_kernel<strict>" if strict else "dif_rows_kernel<fast> (wavefront-sized row strips, greb_rows.hip)", "batch": batch}
    d.update(sweep_stats(ms, 12.0 * n))
    pj = pmc_record(G384_DIF_FILE)
    d["traffic_source"] = codesha.source_check(pj, "profiles/" + G384_DIF_FILE, engine._lib_path)
    if pj:  # the bound as the counters of the committed passes give it (same kernel, same batch)
        d["traffic"] = int(pj["traffic_bytes_per_launch"])
        d["bound"] = (f"hbm: HBM-side traffic {pj['traffic_bytes_per_launch'] / (12.0 * n):.3f} x the algorithmic bytes, "
                      f"{pj['waves_per_simd']:.1f} wavefronts resident per SIMD, VALU active {pj['valu_active_pct']:.0f} % of the SIMD-cycles "
                      f"({pj['valu_insts_per_field']:.0f} vector instructions per field), {pj['wave_cycles_parked_pct']:.0f} % of the wave-cycles "
                      f"parked at s_waitcnt (profiles/{G384_DIF_FILE.replace('.json', '.txt')})")
    else:
        d["traffic"] = None
        d["bound"] = "hbm (no committed counter pass found)"
    out["diffusion_sweep"] = d
    return out


def host_delivered(torch, eng, levels, M, years):
    """The same ensemble years with the monthly means DELIVERED TO HOST MEMORY inside the timed region -- where the
    reference's path ends (its output() writes them to disk, src/greb.f90:962-987).  `value` keeps them in HBM; this
    is the PCIe-inclusive rate: year y's 1.1 MB per member leave on a copy stream while year y+1 integrates."""
    np_ = eng.np
    buf = torch.empty((M, years, 12, 5, np_), dtype=torch.float32, pin_memory=True)
    co2 = np.repeat(levels[:, None], years, 1)
    eng.run(1, levels[:, None], out=buf.numpy().reshape(-1)[: M * 12 * 5 * np_])  # warm the copy path
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    eng.run(years, co2, out=buf.numpy().reshape(-1))
    dt = time.perf_counter() - t0
    return {"value": round(M * years / dt, 2), "unit": "simulated-years/s", "years": years,
            "bytes_to_host_per_year": int(M * 12 * 5 * np_ * 4), "host_memory": "pinned",
            "finite": bool(torch.isfinite(buf).all().item())}


def usable_cores():
    """Host cores this process may use: the affinity mask, capped by a cgroup CPU quota if there is one."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        with open("/sys/fs/cgroup/cpu.max") as f:
            quota, period = f.read().split()
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except (OSError, ValueError):
        try:
            with open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us") as f, open("/sys/fs/cgroup/cpu/cpu.cfs_period_us") as g:
                q, per = int(f.read()), int(g.read())
            if q > 0:
                n = min(n, max(1, q // per))
        except (OSError, ValueError):
            pass
    return n


def cpu_baseline_all_cores(inp, years=(1, 3), cap=64):
    """The reference's own ensemble convention on the host: N independent single-threaded processes, one per core
    (src/greb.f90:153,1064-1068 -- N namelists with N ens_ids; its only threading hook, the omp sections at
    :299-304, is not enabled by its Makefile).  N = min(usable cores, 64); bounded sample = 1+3 model-years each;
    aggregate rate = N x 4 / wall of the slowest."""
    import shutil
    import subprocess
    import tempfile
    from greb_climate_model_amd import workload
    from oracle import oracle as O  # checker/baseline only -- never on the product path
    if not os.path.exists(O.REF_BIN):
        return None
    n = max(1, min(usable_cores(), cap))
    tf, ts = years
    root = tempfile.mkdtemp(prefix="greb_cpu_all_")
    try:
        inp.write_input_dir(os.path.join(root, "input"))
        procs = []
        for i in range(n):
            wd = os.path.join(root, f"m{i:03d}")
            os.makedirs(os.path.join(wd, "output"))
            os.symlink(os.path.join(root, "input"), os.path.join(wd, "input"))
            workload.write_namelist(os.path.join(wd, "namelist"), tf, ts, (280.0 + 840.0 * i / max(n - 1, 1),), 95, 38,
                                    ens_id=f"{i:03d}")
        t0 = time.perf_counter()
        for i in range(n):
            procs.append(subprocess.Popen([O.REF_BIN], cwd=os.path.join(root, f"m{i:03d}"), stdout=subprocess.DEVNULL,
                                          stderr=subprocess.DEVNULL))
        rcs = [p.wait() for p in procs]
        wall = time.perf_counter() - t0
        ok = all(rc == 0 for rc in rcs) and all(
            os.path.getsize(os.path.join(root, f"m{i:03d}", "output", f"scenario_{i:03d}")) == ts * 60 * 96 * 48 * 4 for i in range(n))
        if not ok:
            return {"error": "a reference process failed", "cores": n}
        return {"value": round(n * (tf + ts) / wall, 2), "unit": "simulated-years/s", "cores": n, "kind": "reference",
                "sample": f"{n} independent reference processes (amdflang -O2), one per core, {tf}+{ts} model-years each, "
                          f"96x48, CO2 sweep, same synthetic inputs", "wall_s": round(wall, 2)}
    finally:
        shutil.rmtree(root, ignore_errors=True)


def cpu_baseline(inp):
    """The reference's own loop on this box's host cores: 1 core (it is single-threaded,
    Makefile:12), bounded sample = 1 flux-correction year + 9 scenario years of the same workload."""
    import platform
    from oracle import oracle as O  # checker/baseline only -- never on the product path
    cpu = platform.processor() or ""
    try:
        with open("/proc/cpuinfo") as f:
            for line in f:
                if line.startswith("model name"):
                    cpu = line.split(":", 1)[1].strip(); break
    except OSError:
        pass
    ncores = os.cpu_count()
    if os.path.exists(O.REF_BIN):
        mon, out, wall = O.run_reference_binary(inp, 1, 9)
        # wall includes reading 94 MB of inputs; the model itself is > 95 % of it
        return {"value": round(10 / wall, 3), "unit": "simulated-years/s", "cores": 1, "kind": "reference",
                "sample": "reference Fortran (amdflang -O2) 1+9 model-years, 96x48, same synthetic inputs, 1 member",
                "host_cpu": cpu, "host_cores_available": ncores}
    O.build(ref=False)
    o = O.Oracle(inp)
    t = time.perf_counter(); o.flux_correction(1); o.run(9, 680.0); wall = time.perf_counter() - t
    return {"value": round(10 / wall, 3), "unit": "simulated-years/s", "cores": 1, "kind": "port",
            "sample": "C restatement (gcc -O3 -ffp-contract=off) 1+9 model-years, 96x48, same synthetic inputs, 1 member",
            "host_cpu": cpu, "host_cores_available": ncores}


if __name__ == "__main__":
    main()

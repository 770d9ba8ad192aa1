"""GPU: the N > 1 path of bench.py rehearsed on ONE card -- two rank processes over gloo, both on device 0 (the driver runs
the real thing over RCCL on an 8-GPU node; this is every line of it except the backend).  Caught in round 4: rank 1 of an
N > 1 run crashed on a tensor only rank 0 and the one-GPU run had."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_two_ranks_on_one_card_over_gloo():
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    env.update(GREB_BENCH_BACKEND="gloo", GREB_BENCH_DEVICE="0")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--members", "48",
                        "--root-members", "16", "--no-cpu", "--no-roofline", "--no-g384"], env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, (r.stdout + r.stderr)[-3000:]  # EVERY rank must exit cleanly, not only the one that prints
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["ranks_seen"] == 2 and d["backend"] == "gloo" and d["finite"] is True
    assert d["root_members"] == 16 and d["members_checked"] == 64 and d["gather_verified"] is True and d["inversions"] == 0
    assert d["value"] > 0 and abs(d["value"] - 64 * 2 / (d["ms_per_step"] * 2e-3)) < 1e-6 * d["value"] + 0.02
    pr = d["per_rank_s"]
    assert set(pr) == {"integrate", "gather_wait", "total_before_barrier"}
    for v in pr.values():
        assert len(v["per_rank"]) == 2 and v["per_rank"][v["rank_of_max"]] == v["max"] >= v["min"] > 0


@pytest.mark.parametrize("args", [["4", "--years", "2"], ["5", "--years", "1", "--members", "6"]])
def test_run_config_as_two_ranks_on_one_card(args):
    """tools/run_config.py 4 / 5 (BASELINE's 8-GPU configs) as the driver's launcher would start them -- torch.distributed.run,
    one process per rank -- with two ranks sharing this card over gloo: members dealt to the ranks, each rank's engine(s),
    the gather of the monthly means to rank 0.  (Two PROCESSES with one-launch circulation calls on one device: their
    tasks together are far fewer than its wavefront slots.)"""
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    env.update(GREB_BENCH_BACKEND="gloo", GREB_BENCH_DEVICE="0")
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                        "--master-port", str(29541 + 2 * int(args[0])), os.path.join(ROOT, "tools", "run_config.py"), *args], env=env, capture_output=True,
                       text=True, timeout=900)
    assert r.returncode == 0, (r.stdout + r.stderr)[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    d = json.loads(lines[0])
    n = 8 if args[0] == "4" else 6
    assert d["n_gpus"] == 2 and d["members_total"] == n and d["members_this_rank"] == n // 2 and d["finite"] is True
    t = d["last_year_global_mean_tsurf_C"]
    assert len(t) == n // 2 and all(-20.0 < x < 30.0 for x in t)
    if args[0] == "4":
        assert t == sorted(t) and t[-1] - t[0] > 1.0  # more CO2, warmer (this rank's block of the sweep)

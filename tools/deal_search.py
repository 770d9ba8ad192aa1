#!/usr/bin/env python3
"""Deal experiments for the fused member kernel's FAST schedule (greb_member.hip: deal_fast).

  python tools/deal_search.py build          here (no GPU): one -DGREB_TUNING library per candidate deal under
                                             greb_climate_model_amd/variants/ (git-ignored, travels with gpurun)
  python tools/deal_search.py run [members]  on the GPU box: tools/stamp_member.py for every built variant, one line each

A deal is written as the seven bulk wave slots in wave order (the polar wave -- "P<w>:" prefix, default wave 6 -- is
left out), each a '+'-joined list of passes: S0..S2 = ST (two sub-cycled rows), T0..T2 = FT (two full rows), H = S1, F0..F4 = F1, '-' = idle."""
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
VARDIR = os.path.join(ROOT, "greb_climate_model_amd", "variants")

DEALS = {  # a leading "P<w>:" puts the polar chains on wave w (default 6); the slot list omits that wave
    "p6g":   "P6:S0+F0 S2+F1 T1+F3 T2+H+F2 S1 T0 F4",
    "q1":    "P6:S0+F0 S2+F1 T1+F3 T2+F2 S1 T0 H+F4",
    "q2":    "P6:S0+F0 S2+F1 T1+F3 T2+H S1 T0 F2+F4",
    "q3":    "P6:S0+F0 S2+H T1+F3 T2+F1 S1 T0+F2 F4",
    "q4":    "P6:S0+H S2+F1 T1+F3 T2+F0+F2 S1 T0 F4",
    "q5":    "P6:S0+F0 S2+F1 T1+H T2+F3+F2 S1 T0 F4",
    "q6":    "P6:S0+F0 S2+F1 T1+F3 T0+H+F2 S1 T2 F4",
    # round 3: every SIMD one ST + one FT + one F1 (888 instructions each, where p6g gives SIMD 0 1 008 and SIMD 3 841), the
    # chain SIMD H + 2 F1 beside the chains.  Measured (512 members, one gpurun call, p6g = 4 839 cycles per sub-step):
    # n1 5 002, n5 5 060, n2 5 108, n7 5 155, n4 5 286, n6 5 310 -- equal instruction counts are not equal pipe time
    # (the ST passes are the packed-heavy ones), and an FT pass as the younger wave of an ST wave takes 4 100-4 650 cycles
    "n1":    "P6:S0+F0 S1+F1 H+F3+F4 S2+F2 T0 T1 T2",
    "n2":    "P6:T0+F0 T1+F1 H+F3+F4 T2+F2 S0 S1 S2",
    "n4":    "P6:S0+T0 S1+T1 H+F0+F1 S2+T2 F2 F3 F4",
    "n5":    "P6:S0+F0 S1+F1 F3+F4+H S2+F2 T0 T1 T2",
    "n6":    "P6:S0 S1 H+F3+F4 S2 T0+F0 T1+F1 T2+F2",
    "n7":    "P6:F0+S0 F1+S1 H+F3+F4 F2+S2 T0 T1 T2",
    # after the chain wave got lighter (no test between its sweeps: 4 437 -> 4 132 busy cycles): work towards SIMD 2
    "r1":    "P6:S0+F0 S2+F1 T1+F3+F2 T2+H S1 T0 F4",
    "r2":    "P6:S0 S2+F1 T1+F3+F0 T2+H+F2 S1 T0 F4",
    "r3":    "P6:S0+F0 S2 T1+F3+F1 T2+H+F2 S1 T0 F4",
    "r4":    "P6:S0 S2+F1 T1+F3 T2+H+F2 S1 T0 F0+F4",
}


def polar_wave(deal: str) -> int:
    return int(deal.split(":")[0][1:]) if deal.startswith("P") else 6


def table(deal: str) -> str:
    pw = polar_wave(deal)
    slots = deal.split(":")[-1].split()
    assert len(slots) == 7, deal
    waves = slots[:pw] + ["-"] + slots[pw:]
    rows = []
    for w in waves:
        ps = [] if w == "-" else w.split("+")
        assert len(ps) <= 3, w
        cells = []
        for x in ps:
            kind = {"S": "kST", "T": "kFT", "H": "kS1", "F": "kF1"}[x[0]]
            cells.append("{%s,%d}" % (kind, int(x[1:]) if len(x) > 1 else 0))
        cells += ["{kNone,0}"] * (3 - len(cells))
        rows.append("{" + ",".join(cells) + "}")
    return ",".join(rows)


def build_one(name: str) -> str:
    from greb_climate_model_amd import build
    out = os.path.join(VARDIR, f"libgreb_hip_deal_{name}.so")
    cmd = [build.hipcc(), *build.HIPCC_FLAGS, "-DGREB_TUNING", "-DGREB_DEAL_FAST=" + table(DEALS[name]),
           f"-DGREB_POLAR_WAVE_FAST={polar_wave(DEALS[name])}", "-o", out,
           *[os.path.join(build.CSRC, s) for s in build.SOURCES]]
    subprocess.run(cmd, check=True, cwd=build.CSRC, stdout=subprocess.DEVNULL)
    return out


def main() -> None:
    mode = sys.argv[1] if len(sys.argv) > 1 else "build"
    names = [n for n in (sys.argv[3:] if mode == "run" else sys.argv[2:]) if n in DEALS] or list(DEALS)
    if mode == "build":
        os.makedirs(VARDIR, exist_ok=True)
        with ThreadPoolExecutor(4) as ex:
            for out in ex.map(build_one, names):
                print("built", os.path.relpath(out, ROOT))
        return
    members = sys.argv[2] if len(sys.argv) > 2 else "512"
    for n in names:
        lib = os.path.join(VARDIR, f"libgreb_hip_deal_{n}.so")
        if not os.path.exists(lib):
            continue
        r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "stamp_member.py"), members, "2"],
                           env=dict(os.environ, GREB_LIB=lib), capture_output=True, text=True)
        sub = [l for l in r.stdout.splitlines() if l.startswith("sub-step")]
        busy = [l.split("busy")[1].split("cyc")[0].strip() for l in r.stdout.splitlines() if "busy" in l and "wave" in l]
        print(f"{n:6s} {DEALS[n]:48s} {sub[0] if sub else r.stderr[-200:]}  busy {' '.join(busy)}", flush=True)


if __name__ == "__main__":
    main()

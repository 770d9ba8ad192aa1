#!/usr/bin/env python3
"""Summary of tools/prof_rows.sh's counter passes for the 384x192 diffusion sweep: per kernel the per-dispatch counter
means and the figures derived from them by the rules of MI355X_MICROARCH.md (FETCH_SIZE x 2 on gfx950; SQ_* wave
counters in quad-cycles; GRBM_GUI_ACTIVE summed over the 8 XCDs)."""
import collections, csv, glob, json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from greb_climate_model_amd import build, codesha

d = sys.argv[1]
json_out = sys.argv[2] if len(sys.argv) > 2 else None  # record of the FAST row-strip kernel for bench.py's bound string
cnt = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(os.path.join(d, "*", "run_counter_collection.csv")):
    per = collections.defaultdict(float)
    for r in csv.DictReader(open(f)):
        per[(r["Kernel_Name"], r["Dispatch_Id"], r["Counter_Name"])] += float(r["Counter_Value"])
    for (k, _, c), v in per.items():
        cnt[k][c].append(v)
dur = {}
st = os.path.join(d, "kt", "run_kernel_stats.csv")
if os.path.exists(st):
    for r in csv.DictReader(open(st)):
        dur[r["Name"]] = (float(r["AverageNs"]) * 1e-3, float(r["MinNs"]) * 1e-3, float(r["MaxNs"]) * 1e-3, int(r["Calls"]))
NF = int(os.environ.get("FIELDS", 1024))  # fields per launch (tools/prof_step.sh: 2 per member)
ALGO = 12.0 * NF * 384 * 192
best_step = (0.0, None)
for k in sorted(cnt):
    if "greb" not in k:
        continue
    c = {n: sum(v) / len(v) for n, v in cnt[k].items()}
    print("==", k[:100])
    t = next((v for n, v in dur.items() if n[:60] == k[:60]), None)
    if t:
        print(f"   kernel-trace: avg {t[0]:.1f} us  min {t[1]:.1f}  max {t[2]:.1f}  ({t[3]} launches; unsettled clocks: the run is short)")
    for n in sorted(c):
        print(f"   {n:24s} {c[n]:.5g} per launch")
    if "FETCH_SIZE" in c and "WRITE_SIZE" in c:
        rd, wr = c["FETCH_SIZE"] * 2 * 1024, c["WRITE_SIZE"] * 1024
        print(f"   HBM-side traffic         read {rd / 1e6:.1f} MB (FETCH_SIZE x 2) + write {wr / 1e6:.1f} MB = {(rd + wr) / 1e6:.1f} MB"
              f" = {(rd + wr) / ALGO:.3f} x algorithmic ({ALGO / 1e6:.1f} MB); reads alone {rd / (ALGO * 2 / 3):.3f} x")
    if t and "SQ_ACTIVE_INST_VALU" in c and "GRBM_GUI_ACTIVE" in c:
        cyc = c["GRBM_GUI_ACTIVE"] / 8
        print(f"   busy cycles per XCD      {cyc:.4g} (GRBM_GUI_ACTIVE / 8)")
        print(f"   VALU active              {100 * c['SQ_ACTIVE_INST_VALU'] * 4 / (1024 * cyc):.1f} % of the SIMD-cycles (SQ_ACTIVE_INST_VALU x 4 / (1 024 SIMDs x cycles))")
        if "SQ_INSTS_VALU" in c:
            print(f"   VALU instructions        {c['SQ_INSTS_VALU'] / NF:.0f} per field; one per SIMD every {1024 * cyc / c['SQ_INSTS_VALU']:.2f} cycles")
    if "SQ_WAVE_CYCLES" in c and "GRBM_GUI_ACTIVE" in c:
        cyc = c["GRBM_GUI_ACTIVE"] / 8
        print(f"   waves resident           {c['SQ_WAVE_CYCLES'] * 4 / cyc / 1024:.2f} per SIMD on average (SQ_WAVE_CYCLES x 4 / cycles / 1 024); {c.get('SQ_WAVES', 0):.0f} waves launched")
        for n, label in (("SQ_ACTIVE_INST_ANY", "issuing"), ("SQ_WAIT_INST_ANY", "issue-stalled"), ("SQ_WAIT_ANY", "parked (s_waitcnt)")):
            if n in c:
                print(f"   wave-cycles {label:20s} {100 * c[n] / c['SQ_WAVE_CYCLES']:.1f} %")
    if "SQ_LDS_IDX_ACTIVE" in c and "SQ_LDS_BANK_CONFLICT" in c:
        print(f"   LDS bank-conflict cycles {100 * c['SQ_LDS_BANK_CONFLICT'] / max(c['SQ_LDS_IDX_ACTIVE'], 1):.1f} % of the LDS-active cycles")
    if "TCC_HIT_sum" in c and "TCC_MISS_sum" in c:
        print(f"   L2 hit rate              {100 * c['TCC_HIT_sum'] / (c['TCC_HIT_sum'] + c['TCC_MISS_sum']):.1f} %")
    if json_out and "dif_rows_kernel<false" in k and "FETCH_SIZE" in c and "SQ_BUSY_CYCLES" in c:
        cyc = c["SQ_BUSY_CYCLES"] / 32  # one SQ per shader engine, 32 of them: same pass as the wave counters
        rec = {"kernel": "greb::dif_rows_kernel<false, 0>",
               "source": "rocprofv3 --pmc passes (each on its own) on `python tools/microbench_dif.py 1024 384 192`, MI355X, "
                         "tools/prof_rows.sh; FETCH_SIZE / WRITE_SIZE in KiB, FETCH_SIZE x 2 on gfx950 (MI355X_MICROARCH.md, HBM); "
                         "SQ_* wave counters in quad-cycles; SQ_BUSY_CYCLES summed over the 32 shader engines",
               "batch": 1024, "fetch_size_kb_per_launch": round(c["FETCH_SIZE"]), "write_size_kb_per_launch": round(c["WRITE_SIZE"]),
               "traffic_bytes_per_launch": round((c["FETCH_SIZE"] * 2 + c["WRITE_SIZE"]) * 1024),
               "algorithmic_bytes_per_launch": int(ALGO),
               "valu_insts_per_field": round(c["SQ_INSTS_VALU"] / 1024),
               "valu_active_pct": round(100 * c["SQ_ACTIVE_INST_VALU"] * 4 / (1024 * cyc), 1),
               "waves_per_simd": round(c["SQ_WAVE_CYCLES"] * 4 / cyc / 1024, 2),
               "wave_cycles_issuing_pct": round(100 * c["SQ_ACTIVE_INST_ANY"] / c["SQ_WAVE_CYCLES"], 1),
               "wave_cycles_issue_stalled_pct": round(100 * c["SQ_WAIT_INST_ANY"] / c["SQ_WAVE_CYCLES"], 1),
               "wave_cycles_parked_pct": round(100 * c["SQ_WAIT_ANY"] / c["SQ_WAVE_CYCLES"], 1),
               "code": codesha.record(build.LIB, "dif_rows_kernelILb0E")}
        json.dump(rec, open(json_out, "w"), indent=1)
    # the engine's circulation kernels at 384x192 (tools/prof_step.sh: FIELDS = 2 x members): the issue floor bench.py quotes
    step_json = os.environ.get("STEP_JSON")
    sub = "circ_rows_kernel<false" in k or "step_rows_kernel<false" in k
    if step_json and sub and "SQ_INSTS_VALU" in c and "SQ_BUSY_CYCLES" in c:
        one_call = "circ_rows_kernel" in k
        nsub = 24 if one_call else 1  # a circ_rows_kernel launch is a whole circulation call
        cyc = c["SQ_BUSY_CYCLES"] / 32
        rec = {"kernel": "greb::circ_rows_kernel<false>" if one_call else "greb::step_rows_kernel<false>", "fields": NF,
               "source": f"profiles/r04_g384_substep_pmc.txt: rocprofv3 --pmc passes (each on its own), tools/prof_step.sh",
               "valu_insts_per_substep": round(c["SQ_INSTS_VALU"] / nsub),
               "issue_floor_us": round(c["SQ_INSTS_VALU"] / nsub * 4 / 1024 / 2.4e3, 1),  # one instruction per SIMD per 4 cycles, 1 024 SIMDs, ~2.4 GHz
               "valu_active_pct": round(100 * c["SQ_ACTIVE_INST_VALU"] * 4 / (1024 * cyc), 1) if "SQ_ACTIVE_INST_VALU" in c else None,
               "waves_per_simd": round(c["SQ_WAVE_CYCLES"] * 4 / cyc / 1024, 2) if "SQ_WAVE_CYCLES" in c else None,
               "code": codesha.record(build.LIB, "circ_rows_kernelILb0E" if one_call else "step_rows_kernelILb0E")}
        weight = len(cnt[k]["SQ_BUSY_CYCLES"]) * c["SQ_BUSY_CYCLES"]  # the form the engine settled on is the one that ran the year
        if weight > best_step[0]:
            best_step = (weight, rec)
if best_step[1] is not None:
    json.dump(best_step[1], open(os.environ["STEP_JSON"], "w"), indent=1)

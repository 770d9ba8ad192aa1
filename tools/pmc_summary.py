#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc counter_collection.csv files: per kernel, per counter, summed over dispatches.

  pmc_summary.py a.csv b.csv ...                         the raw sums
  pmc_summary.py --derive sq.csv lds.csv kernel_stats.csv  derived figures of the single-member dispatch of the fused
      member kernel (flux phase, one workgroup = one CU, so the per-dispatch sums are that CU's own), by the rules of
      MI355X_MICROARCH.md: effective clock = GRBM_GUI_ACTIVE / 8 / duration (the counter sums over the 8 XCDs);
      SQ_ACTIVE_INST_* / SQ_WAIT_* / SQ_WAVE_CYCLES count quad-cycles (x4 = cycles)."""
import collections
import csv
import sys


def load(f):
    agg = collections.defaultdict(lambda: collections.defaultdict(float))
    ndisp = collections.defaultdict(set)
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
        ndisp[k].add(r["Dispatch_Id"])
    return agg, ndisp


def raw(files):
    for f in files:
        agg, ndisp = load(f)
        print("==", f)
        for k, d in agg.items():
            if "greb" not in k:
                continue
            print(k[:90], " dispatches:", len(ndisp[k]))
            for c, v in sorted(d.items()):
                print(f"    {c:28s} {v:.5g}   per-dispatch {v / len(ndisp[k]):.5g}")


def derive(sq, lds, stats):
    key = "member_kernel<false, true, false>"  # flux-correction year of ONE member (shared physics): 8 waves on one CU
    c = {}
    for f in (sq, lds):
        agg, ndisp = load(f)
        for k, d in agg.items():
            if key in k:
                for name, v in d.items():
                    c[name] = v / len(ndisp[k])
    dur = None
    for r in csv.DictReader(open(stats)):
        if key in r["Name"]:
            dur = float(r["AverageNs"]) * 1e-9
    print("== derived, single-member dispatch of", key, "(one model year, one CU)")
    if dur and "GRBM_GUI_ACTIVE" in c:
        print(f"    effective clock            {c['GRBM_GUI_ACTIVE'] / 8 / dur / 1e9:.3f} GHz  (GRBM_GUI_ACTIVE / 8 / {dur * 1e3:.2f} ms)")
    busy = c.get("SQ_BUSY_CYCLES")
    if busy:
        if "SQ_ACTIVE_INST_VALU" in c:
            print(f"    VALU active per SIMD       {100 * c['SQ_ACTIVE_INST_VALU'] * 4 / (4 * busy):.1f} %  (SQ_ACTIVE_INST_VALU x4 / (4 SIMDs x SQ_BUSY_CYCLES))")
        if "SQ_INSTS_VALU" in c:
            print(f"    VALU instructions          {c['SQ_INSTS_VALU'] / 1e6:.2f} M per member-year = {c['SQ_INSTS_VALU'] / (730 * 24):.0f} per sub-step incl. the point physics;"
                  f" one per SIMD every {4 * busy / c['SQ_INSTS_VALU']:.2f} cycles")
        if "SQ_INSTS_LDS" in c:
            print(f"    LDS instructions           {c['SQ_INSTS_LDS'] / 1e6:.2f} M per member-year")
    wc = c.get("SQ_WAVE_CYCLES")
    if wc:
        for n, label in (("SQ_ACTIVE_INST_ANY", "issuing"), ("SQ_WAIT_INST_ANY", "issue-stalled"), ("SQ_WAIT_ANY", "parked (s_waitcnt / s_barrier)")):
            if n in c:
                print(f"    wave cycles {label:30s} {100 * c[n] / wc:.1f} %")
    if "SQ_LDS_IDX_ACTIVE" in c and "SQ_LDS_BANK_CONFLICT" in c:
        print(f"    LDS bank-conflict cycles   {100 * c['SQ_LDS_BANK_CONFLICT'] / c['SQ_LDS_IDX_ACTIVE']:.1f} % of the LDS-active cycles")
        if busy:
            print(f"    LDS active                 {100 * c['SQ_LDS_IDX_ACTIVE'] / busy:.1f} % of the busy cycles")


def traffic_json(fetch, write):
    """profiles/rNN_roofline_traffic.json: HBM-side bytes per launch of the 96x48 roofline kernel (bench.py reads it)."""
    import json, os
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from greb_climate_model_amd import build, codesha
    key = "diffusion_stream_kernel<false"
    vals = {}
    for f, name in ((fetch, "FETCH_SIZE"), (write, "WRITE_SIZE")):
        agg, ndisp = load(f)
        for k, d in agg.items():
            if key in k:
                vals[name] = d[name] / len(ndisp[k])
    fk, wk = vals["FETCH_SIZE"], vals["WRITE_SIZE"]
    print(json.dumps({"kernel": "greb::diffusion_stream_kernel<false, 96, 48>",
                      "source": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) on `python tools/microbench_dif.py 16384`, "
                                "MI355X (tools/verify_round.sh profiles).  Units: KiB per launch; FETCH_SIZE x2 on gfx950 "
                                "(MI355X_MICROARCH.md, HBM section)",
                      "batch": 16384, "fetch_size_kb_per_launch": round(fk), "write_size_kb_per_launch": round(wk),
                      "gfx950_fetch_correction": 2.0, "traffic_bytes_per_launch": round((2 * fk + wk) * 1024),
                      "algorithmic_bytes_per_launch": 12 * 16384 * 96 * 48,
                      # which machine code the passes ran on: the library of THIS tree (the passes and this summary belong
                      # to one tools/verify_round.sh run); bench.py and tests/test_profiles_cpu.py compare it with theirs
                      "code": codesha.record(build.LIB, "diffusion_stream_kernelILb0ELi96ELi48E")}, indent=1))


if __name__ == "__main__":
    if sys.argv[1:2] == ["--derive"]:
        derive(*sys.argv[2:5])
    elif sys.argv[1:2] == ["--traffic-json"]:
        traffic_json(*sys.argv[2:4])
    else:
        raw(sys.argv[1:])

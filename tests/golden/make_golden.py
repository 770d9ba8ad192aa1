#!/usr/bin/env python3
"""Mint golden vectors from the compiled reference.  BUILD CONTAINER ONLY (needs oracle/_ref).

Runs the reference Fortran -- oracle/_ref/libgreb_ref.so per routine and oracle/_ref/greb_ref for
whole runs, both built by oracle/Makefile with amdflang -O2 straight from
/root/reference/src/greb.f90 -- on the synthetic workload (greb_climate_model_amd/workload.py) and
writes small fixtures (inputs + the reference's outputs; data only, no reference text):

  routine_g96.npz      a1-a8 per routine at ityr = 1, 365, 730, plus the grid tables
  run_short_g96.npz    1+2-yr default-namelist run: all 24 months x 5 fields + stdout scalars
  run_default_g96.npz  3+50-yr default namelist (BASELINE config 1): months 1,12,300,600 + per-month
                       statistics of all 600 months + sha256 of the full output file
  ensemble_g96.npz     BASELINE config 4 in miniature: 8 CO2 levels, 1+3 yr, last December
  MANIFEST.json        compiler line, timings, and the oracle-vs-reference verdict per item

While doing so it asserts that the C restatement (oracle/greb_oracle.c) reproduces every one of
these outputs BIT FOR BIT; that is the oracle's parity pin.
"""
import hashlib
import json
import os
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from greb_climate_model_amd import abi, workload  # noqa: E402
from oracle import oracle as O  # noqa: E402

OUT = os.path.join(ROOT, "tests", "golden")
f32 = np.float32
quick = "--quick" in sys.argv


def stats(mon):
    """[months][5] mean/min/max + cos-lat-weighted mean, float64."""
    ny = mon.shape[2]
    lat = (np.arange(ny) + 0.5) * 180.0 / ny - 90.0
    w = np.cos(np.deg2rad(lat))[None, None, :, None]
    m64 = mon.astype(np.float64)
    return np.stack([m64.mean((2, 3)), m64.min((2, 3)), m64.max((2, 3)),
                     (m64 * w).sum((2, 3)) / (w.sum() * mon.shape[3])], axis=-1)


def main():
    O.build(ref=True)
    manifest = {
        "reference": "sieste/greb-climate-model src/greb.f90 (compiled in place, never copied)",
        "compiler": subprocess.run(["/opt/rocm/bin/amdflang", "--version"], capture_output=True,
                                   text=True).stdout.splitlines()[0],
        "flags": "-O2",
        "workload": "synthetic, tests/golden/basis_g96.npz expanded by greb_climate_model_amd/workload.py",
        "noise_floor_K_rms": {"note": "reference -O0/-O3-fma vs -O2, SURVEY.md C.2",
                              "Tsurf": 1.7e-5, "Tair": 1.6e-5, "Tocean": 1.4e-5, "q": 3.3e-9, "albedo": 2.0e-7},
        "items": {},
    }
    inp = workload.make_inputs()
    params = abi.default_params(ipx=95, ipy=38)  # the shipped namelist's diagnostic point (namelist:4-5)

    # ---------------------------------------------------------------- per routine
    orc = O.Oracle(inp, params)
    ref = O.RefLib(inp, orc)
    rng = np.random.default_rng(7)
    wz_air, wz_vapor = orc.field(5).copy(), orc.field(6).copy()
    rt = {"wz_air": wz_air, "wz_vapor": wz_vapor, "z_ocean": orc.field(7).copy(), "toclim": orc.field(8).copy()}
    for k, v in orc.grid().items():
        rt["grid_" + k] = np.asarray(v)
    all_equal = True
    for ityr in (1, 365, 730):
        n = lambda s: (s * rng.standard_normal((48, 96))).astype(f32)
        Ts = (inp.tclim[ityr - 1] + n(2.0)).astype(f32)
        Ta = (Ts + n(1.5)).astype(f32)
        To = (orc.field(8) + rng.random((48, 96)).astype(f32) * f32(3)).astype(f32)
        q = (inp.qclim[ityr - 1] * (f32(0.9) + f32(0.2) * rng.random((48, 96)).astype(f32))).astype(f32)
        cap0 = orc.field(4).copy()
        co2 = 298.0 + 100.0 * (ityr % 7)
        refo = {}
        refo["dif_Ta"] = ref.diffusion(Ta, wz_air)
        refo["dif_q"] = ref.diffusion(q, wz_vapor)
        refo["adv_Ta"] = ref.advection(ityr, Ta, wz_air)
        refo["adv_q"] = ref.advection(ityr, q, wz_vapor)
        refo["crc_Ta"] = ref.circulation(ityr, Ta, wz_air)
        refo["crc_q"] = ref.circulation(ityr, q, wz_vapor)
        refo["sw"], refo["albedo"] = ref.swradiation(ityr, Ts)
        refo["LWsurf"], refo["LWair_up"], refo["LWair_down"], refo["em"] = ref.lwradiation(ityr, Ts, Ta, q, co2)
        refo["Qlat"], refo["Qlat_air"], refo["dq_eva"], refo["dq_rain"] = ref.hydro(ityr, Ts, q)
        refo["dT_ocean"], refo["dTo"] = ref.deep_ocean(ityr, Ts, To)
        refo["cap_surf_new"] = ref.seaice(ityr, Ts, cap0)
        orco = {}
        orco["dif_Ta"] = orc.diffusion(Ta, wz_air); orco["dif_q"] = orc.diffusion(q, wz_vapor)
        orco["adv_Ta"] = orc.advection(Ta, wz_air, ityr=ityr); orco["adv_q"] = orc.advection(q, wz_vapor, ityr=ityr)
        orco["crc_Ta"] = orc.circulation(Ta, wz_air, ityr=ityr); orco["crc_q"] = orc.circulation(q, wz_vapor, ityr=ityr)
        orco["sw"], orco["albedo"] = orc.swradiation(ityr, Ts)
        orco["LWsurf"], orco["LWair_up"], orco["LWair_down"], orco["em"] = orc.lwradiation(ityr, Ts, Ta, q, co2)
        orco["Qlat"], orco["Qlat_air"], orco["dq_eva"], orco["dq_rain"] = orc.hydro(ityr, Ts, q)
        orco["dT_ocean"], orco["dTo"] = orc.deep_ocean(ityr, Ts, To)
        orc.field(4)[:] = cap0
        orco["cap_surf_new"] = orc.seaice(ityr, Ts)
        orc.field(4)[:] = cap0
        for k in refo:
            eq = bool(np.array_equal(refo[k], orco[k]))
            all_equal &= eq
            if not eq:
                print("MISMATCH routine", ityr, k, np.abs(refo[k] - orco[k]).max())
            rt[f"t{ityr}_out_{k}"] = refo[k]
        for k, v in (("Ts", Ts), ("Ta", Ta), ("To", To), ("q", q), ("cap_surf", cap0)):
            rt[f"t{ityr}_in_{k}"] = v
        rt[f"t{ityr}_co2"] = f32(co2)
    np.savez_compressed(os.path.join(OUT, "routine_g96.npz"), **rt)
    manifest["items"]["routine_g96"] = {"oracle_bit_identical": all_equal, "ityr": [1, 365, 730]}
    print("routine: oracle bit-identical to reference:", all_equal)
    assert all_equal
    orc.close()

    # ---------------------------------------------------------------- 1+2-yr run
    def whole(tag, tf, ts, co2, full_months=None):
        mon_ref, out, wall = O.run_reference_binary(inp, tf, ts, (co2,))
        o2 = O.Oracle(inp, params)
        t0 = time.time()
        yf = o2.flux_correction(tf)
        mon, yr = o2.run(ts, co2)
        wall_o = time.time() - t0
        state = o2.state5()
        o2.close()
        mon = mon.reshape(-1, 5, 48, 96)
        eq = bool(np.array_equal(mon, mon_ref))
        lines = O.parse_ref_stdout(out)
        yearly_ref = lines[:, 2:4].astype(f32)
        yearly_orc = np.concatenate([yf, yr]).astype(f32)
        yeq = bool(np.allclose(yearly_ref, yearly_orc, rtol=0, atol=2e-6 * 300))
        print(f"{tag}: reference {wall:.1f}s oracle {wall_o:.1f}s monthly bit-identical={eq} yearly match={yeq}")
        assert eq and yeq, (tag, np.abs(yearly_ref - yearly_orc).max())
        item = {"time_flux": tf, "time_scnr": ts, "co2_ppm": co2, "reference_wall_s": round(wall, 2),
                "oracle_wall_s": round(wall_o, 2), "reference_years_per_s": round((tf + ts) / wall, 3),
                "oracle_years_per_s": round((tf + ts) / wall_o, 3),
                "oracle_bit_identical": eq, "sha256": hashlib.sha256(mon_ref.tobytes()).hexdigest()}
        d = {"yearly": yearly_ref, "stats": stats(mon_ref), "final_state5": state}
        if full_months is None:
            d["monthly"] = mon_ref
        else:
            d["months"] = np.asarray(full_months)
            d["monthly_sel"] = mon_ref[[m - 1 for m in full_months]]
        return item, d

    item, d = whole("run_short", 1, 2, 680.0)
    np.savez_compressed(os.path.join(OUT, "run_short_g96.npz"), **d)
    manifest["items"]["run_short_g96"] = item

    if not quick:
        item, d = whole("run_default", 3, 50, 680.0, full_months=[1, 12, 300, 600])
        np.savez_compressed(os.path.join(OUT, "run_default_g96.npz"), **d)
        manifest["items"]["run_default_g96"] = item

    # ---------------------------------------------------------------- ensemble (config 4 in miniature)
    levels = [280.0, 400.0, 520.0, 640.0, 760.0, 880.0, 1000.0, 1120.0]
    dec, yrs = [], []
    for c in levels:
        mon_ref, out, wall = O.run_reference_binary(inp, 1, 3, (c,))
        dec.append(mon_ref[-1])
        yrs.append(O.parse_ref_stdout(out)[:, 2:4].astype(f32))
    np.savez_compressed(os.path.join(OUT, "ensemble_g96.npz"), co2=np.asarray(levels, f32),
                        december=np.stack(dec), yearly=np.stack(yrs))
    manifest["items"]["ensemble_g96"] = {"time_flux": 1, "time_scnr": 3, "levels": levels}

    manifest["input_sha256"] = {k: hashlib.sha256(np.ascontiguousarray(getattr(inp, k)).tobytes()).hexdigest()
                                for k in ("tclim", "qclim", "uclim", "vclim", "mldclim", "cldclim", "swetclim",
                                          "z_topo", "glacier", "sw_solar")}
    with open(os.path.join(OUT, "MANIFEST.json"), "w") as f:
        json.dump(manifest, f, indent=1)
    print("wrote MANIFEST.json")


if __name__ == "__main__":
    main()

"""Host-side sequencing of the upstream model variant's sensitivity experiments on the MI355X
engine.  Mirrors greb_model of src/greb.original.model.f90 (:139-233): what an experiment changes in the
BOUNDARY DATA, the CO2 series and the order of runs lives here; what it changes in the PROCESSES is the
engine's switch set (include/greb_engine.h GREB_X_*, greb_log_exp_switches).  Every number comes from the HIP
library; nothing here computes model physics.
"""
from __future__ import annotations

import copy

import numpy as np

from . import abi, engine, workload


def original_params(**over) -> abi.GrebParams:
    """The original's compile-time constants: as src/greb.f90's defaults except cp_land = cp_ocean/4.5
    (greb.original.model.f90:69; the namelist version has 926.222)."""
    p = engine.params_default()
    p.cp_land = float(np.float32(p.cp_ocean) / np.float32(4.5))
    for k, v in over.items():
        setattr(p, k, v)
    return p


def experiment_inputs(inp: workload.Inputs, log_exp: int, d_ocean: float = 50.0) -> workload.Inputs:
    """Boundary data of experiment `log_exp` (greb.original.model.f90:162-166); returns a modified copy."""
    out = copy.copy(inp)
    if log_exp == 1:
        out.z_topo = np.where(inp.z_topo > 1.0, np.float32(1.0), inp.z_topo).astype(np.float32)  # :162
    if log_exp <= 2:
        out.cldclim = np.full_like(inp.cldclim, 0.7)       # :163
    if log_exp <= 3:
        out.qclim = np.full_like(inp.qclim, 0.0052)        # :164
    if log_exp <= 9 or log_exp == 11:
        out.mldclim = np.full_like(inp.mldclim, d_ocean)   # :165-166
    return out


def co2_level(log_exp: int, year: float) -> float:
    """co2_level (:939-951) in the original's fp32 arithmetic: 680 ppm, or the A1B ramp for log_exp 12/13."""
    f, y = np.float32, np.float32(year)
    co2 = f(680.0)
    if log_exp in (12, 13):
        if y <= 2000:
            co2 = f(310.0) + f(60.0) / f(50.0) * (y - f(1950.0))
        if 2000 < y <= 2050:
            co2 = f(370.0) + f(150.0) / f(50.0) * (y - f(2000.0))
        if 2050 < y <= 2100:
            co2 = f(520.0) + f(180.0) / f(50.0) * (y - f(2050.0))
    return float(co2)


def run_original(inp: workload.Inputs, log_exp: int, time_flux: int, time_ctrl: int, time_scnr: int,
                 strict: bool = False, device: int = 0, multilaunch: bool = False):
    """flux correction at CO2_ctrl -> control run -> scenario run (:198-232).  Both runs start from the state
    the flux-correction phase ended in (its dummy arguments alias Ts_ini.., :201,:361); cap_surf carries
    through.  Returns (control, scenario) monthly means [years][12][5][ny][nx] (control None if time_ctrl = 0)."""
    co2_ctrl = 298.0 if log_exp in (12, 13) else 340.0                      # :178-179
    p = original_params(co2_flux=co2_ctrl)
    x = engine.log_exp_switches(log_exp)
    e = engine.Engine(experiment_inputs(inp, log_exp, p.d_ocean), p, strict=strict, device=device, multilaunch=multilaunch)
    try:
        e.set_experiment(x & ~abi.X_SST_PLUS1)
        e.flux_correction(time_flux)
        start = e.state()                                                   # Ts, Ta, To, q, cap_surf
        ctrl = None
        if time_ctrl > 0:
            ctrl, _ = e.run(time_ctrl, co2_ctrl)                            # :208-215
            ctrl = ctrl[0]
        restart = e.state()
        restart[:4] = start[:4]                                             # :219; cap_surf stays as the control run left it
        e.set_corrections(None, restart)
        e.set_experiment(x)                                                 # SST+1 applies to the scenario only (:224-226)
        if 14 <= log_exp <= 16:
            co2 = [co2_ctrl] * time_scnr                                    # :225
        else:
            co2 = [co2_level(log_exp, 1940.0 + n) for n in range(time_scnr)]  # :220-222
        scen, _ = e.run(time_scnr, np.asarray(co2, np.float32))
        return ctrl, scen[0]
    finally:
        e.close()

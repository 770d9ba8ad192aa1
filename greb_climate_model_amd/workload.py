"""Synthetic GREB workload: the ten input fields, bit-reproducibly, on any machine.

The reference reads ten raw little-endian fp32 direct-access files (src/greb.f90:1018-1027,
1073-1085); seven of them are not shipped (.MISSING_LARGE_BLOBS).  This module expands the
committed basis (tests/golden/basis_g96.npz, minted by tests/golden/make_basis.py) into the
full (nstep_yr, ny, nx) climatologies using fp32 multiply / add / clip only, so the reference
binary, the CPU oracle and the HIP engine all see byte-identical inputs here and on the GPU
box.  Arrays are C-ordered [t][lat][lon] == Fortran (lon, lat, t), longitude fastest.
"""
from __future__ import annotations

import os
from dataclasses import dataclass, field

import numpy as np

NSTEP_YR = 730
_HERE = os.path.dirname(os.path.abspath(__file__))
BASIS_PATH = os.path.join(os.path.dirname(_HERE), "tests", "golden", "basis_g96.npz")

# file name in input/ -> key (src/greb.f90:1018-1027)
INPUT_FILES = {
    "tsurf": "tclim", "vapor": "qclim", "topography": "z_topo", "soil.moisture": "swetclim",
    "solar.radiation": "sw_solar", "zonal.wind": "uclim", "meridional.wind": "vclim",
    "ocean.mld": "mldclim", "cloud.cover": "cldclim", "glacier.masks": "glacier",
}

f32 = np.float32


@dataclass
class Inputs:
    """The reference's input set (SURVEY.md 8b `greb_fields`)."""
    nx: int
    ny: int
    z_topo: np.ndarray      # [ny][nx]
    glacier: np.ndarray     # [ny][nx]
    sw_solar: np.ndarray    # [730][ny]
    tclim: np.ndarray       # [730][ny][nx]
    qclim: np.ndarray
    uclim: np.ndarray
    vclim: np.ndarray
    mldclim: np.ndarray
    cldclim: np.ndarray
    swetclim: np.ndarray
    meta: dict = field(default_factory=dict)

    def write_input_dir(self, path: str) -> None:
        """Write the ten files exactly as the reference opens them (src/greb.f90:1018-1027)."""
        os.makedirs(path, exist_ok=True)
        for fname, key in INPUT_FILES.items():
            np.ascontiguousarray(getattr(self, key), dtype="<f4").tofile(os.path.join(path, fname))


def _upsample_axis(a: np.ndarray, axis: int, n_out: int, periodic: bool) -> np.ndarray:
    """fp32 bilinear resample on cell-centred grids: x=(J+0.5)*n/N-0.5."""
    n_in = a.shape[axis]
    J = np.arange(n_out, dtype=np.float64)
    x = (J + 0.5) * n_in / n_out - 0.5
    i0 = np.floor(x).astype(np.int64)
    w = (x - i0).astype(f32)            # exact binary fractions for integer refinement ratios
    i1 = i0 + 1
    if periodic:
        i0 %= n_in
        i1 %= n_in
    else:
        i0 = np.clip(i0, 0, n_in - 1)
        i1 = np.clip(i1, 0, n_in - 1)
    a0 = np.take(a, i0, axis=axis)
    a1 = np.take(a, i1, axis=axis)
    shape = [1] * a.ndim
    shape[axis] = n_out
    w = w.reshape(shape)
    return (a0 * (f32(1.0) - w) + a1 * w).astype(f32)


def _upsample2d(a: np.ndarray, ny: int, nx: int) -> np.ndarray:
    a = _upsample_axis(a.astype(f32), a.ndim - 1, nx, periodic=True)    # longitude first
    return _upsample_axis(a, a.ndim - 2, ny, periodic=False)            # then latitude


def load_basis(path: str = BASIS_PATH) -> dict:
    with np.load(path, allow_pickle=False) as z:
        return {k: z[k].astype(f32) for k in z.files}


def make_inputs(nx: int = 96, ny: int = 48, basis: dict | None = None) -> Inputs:
    """Expand the basis into the full input set at 96x48 or a refinement of it (384x192)."""
    b = dict(load_basis() if basis is None else basis)
    season = b.pop("season")
    if (ny, nx) != (48, 96):
        if nx % 96 or ny % 48:
            raise ValueError("grid must be an integer refinement of 96x48")
        solar = b.pop("solar")
        b = {k: _upsample2d(v, ny, nx) for k, v in b.items()}
        b["solar"] = _upsample_axis(solar, 1, ny, periodic=False)
    s = [season[i].astype(f32)[:, None, None] for i in range(4)]

    def lin(k0, k1, si=0):
        return (b[k0][None] + b[k1][None] * s[si]).astype(f32)

    tclim = lin("T0", "T1")
    qclim = lin("q0", "q1")
    swet = np.clip(lin("s0", "s1", 1), f32(0.05), f32(1.0))
    uclim = lin("u0", "u1")
    vclim = ((b["v0"][None] + b["v1"][None] * s[2]).astype(f32) + b["v2"][None] * s[3]).astype(f32)
    mld = np.clip(lin("m0", "m1"), f32(15.0), f32(400.0))
    cld = np.clip(lin("c0", "c1"), f32(0.1), f32(0.95))
    return Inputs(nx=nx, ny=ny, z_topo=b["topography"], glacier=b["glacier"], sw_solar=b["solar"],
                  tclim=tclim, qclim=qclim, uclim=uclim, vclim=vclim, mldclim=mld, cldclim=cld,
                  swetclim=swet, meta={"basis": os.path.basename(BASIS_PATH)})


def routine_inputs_g384(inp: Inputs):
    """The (Tair, q, ityr) the 384x192 per-routine golden vectors were minted on (tests/golden/routine_g384.npz):
    climatology slices of step 400 displaced by a tenth of the seasonal difference, fp32 arithmetic only."""
    ityr = 400
    Ts = inp.tclim[ityr - 1]
    Ta = (Ts + f32(0.1) * (inp.tclim[99] - inp.tclim[499])).astype(f32)
    q = (inp.qclim[ityr - 1] * f32(0.95)).astype(f32)
    return Ta, q, ityr


def write_namelist(path: str, time_flux: int, time_scnr: int, co2_ppm=(680.0,), ipx: int = 95,
                   ipy: int = 38, output_file: str = "output/scenario", ens_id: str = "",
                   physics: dict | None = None, co2_flux: float | None = None,
                   year0: int | None = None) -> None:
    """A namelist file in the reference's four-group format (namelist:1-14, doc/namelist.md)."""
    lines = ["&PHYSICS_PAR"]
    for k, v in (physics or {}).items():
        lines.append(f"  {k} = {v!r}" if not isinstance(v, (list, tuple))
                     else f"  {k} = " + ", ".join(repr(float(x)) for x in v))
    lines += ["/", "&NUMERICS_PAR", f"  ipx = {ipx}", f"  ipy = {ipy}",
              f"  time_flux = {time_flux}", f"  time_scnr = {time_scnr}"]
    if year0 is not None:
        lines.append(f"  year0 = {year0}")
    lines += ["/", "&DIAGNOSTICS_PAR", f'  output_file = "{output_file}"']
    if ens_id:
        lines.append(f'  ens_id = "{ens_id}"')
    lines += ["/", "&CO2_PAR", "  co2_ppm = " + ", ".join(repr(float(c)) for c in co2_ppm)]
    if co2_flux is not None:
        lines.append(f"  co2_flux = {float(co2_flux)!r}")
    lines += ["/", ""]
    with open(path, "w") as f:
        f.write("\n".join(lines))


def read_greb(path: str, nx: int = 96, ny: int = 48, nvar: int = 5) -> np.ndarray:
    """Read an output/scenario file -> [month][var][ny][nx]; var order Tsurf, Tair, Tocean, q,
    albedo (src/greb.f90:978-982).  Same size assertion as R/functions.R:41."""
    a = np.fromfile(path, dtype="<f4")
    rec = nx * ny * nvar
    if a.size % rec:
        raise ValueError(f"{path}: size {a.size * 4} is not a multiple of {rec * 4} bytes")
    return a.reshape(-1, nvar, ny, nx)

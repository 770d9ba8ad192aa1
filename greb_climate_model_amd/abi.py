"""ctypes mirror of include/greb_engine.h (structs + defaults).  Pure host-side plumbing."""
from __future__ import annotations

import ctypes as C
import math

import numpy as np

NSTEP_YR = 730
NVAR_OUT = 5
F_STRICT = 1
F_MULTILAUNCH = 2
F_ROW_STRIPS = 4
F_NO_PERSISTENT = 8
F_PERSISTENT = 16
RUN_DEVICE_OUT = 1
# sensitivity-experiment switches, GREB_X_* of include/greb_engine.h
X_NO_ICE, X_NO_HYDRO, X_NO_DEEP_OCEAN, X_LW_LINEAR_VAPOR = 1, 2, 4, 8
X_NO_CIRCULATION, X_NO_VAPOR_TRANSPORT, X_VAPOR_DIFFUSION_ONLY, X_SST_PLUS1 = 16, 32, 64, 128

c_float_p = C.POINTER(C.c_float)

_PHYS = ["pi", "sig", "rho_ocean", "rho_land", "rho_air", "cp_ocean", "cp_land", "cp_air", "eps",
         "d_ocean", "d_land", "d_air", "ct_sens", "da_ice", "a_no_ice", "a_cloud",
         "Tl_ice1", "Tl_ice2", "To_ice1", "To_ice2",
         "co_turb", "kappa", "ce", "cq_latent", "cq_rain", "z_air", "z_vapor", "r_qviwv"]


class GrebParams(C.Structure):
    """struct greb_params: physics_par in declaration order (src/greb.f90:68-101,128-132)."""
    _fields_ = ([(n, C.c_float) for n in _PHYS] + [("p_emi", C.c_float * 10), ("co2_flux", C.c_float),
                ("ipx", C.c_int32), ("ipy", C.c_int32), ("year0", C.c_int32),
                ("dt", C.c_int32), ("dt_crcl", C.c_int32)])

    PHYSICS_NAMES = tuple(_PHYS)


class GrebFields(C.Structure):
    _fields_ = [(n, c_float_p) for n in ("z_topo", "glacier", "sw_solar", "tclim", "qclim", "uclim",
                                         "vclim", "mldclim", "cldclim", "swetclim")]


class GrebMemberOverrides(C.Structure):
    _fields_ = [(n, C.c_float) for n in ("da_ice", "a_no_ice", "a_cloud", "kappa")]


def default_params(**over) -> GrebParams:
    """Reference defaults (src/greb.f90:49-53,68-104), constants folded in fp32 like the compiler."""
    f = np.float32
    p = GrebParams()
    vals = dict(pi=3.1416, sig=5.6704e-8, rho_ocean=999.1, rho_land=2600., rho_air=1.2, cp_ocean=4186.,
                cp_land=926.222, cp_air=1005., eps=1., d_ocean=50., d_land=2., d_air=5000.,
                ct_sens=22.5, da_ice=0.25, a_no_ice=0.1, a_cloud=0.35,
                Tl_ice1=f(273.15) - f(10.), Tl_ice2=273.15, To_ice1=f(273.15) - f(7.),
                To_ice2=f(273.15) - f(1.7), co_turb=5.0, kappa=8e5, ce=2e-3, cq_latent=2.257e6,
                cq_rain=f(f(-0.1) / f(24.)) / f(3600.), z_air=8400., z_vapor=5000., r_qviwv=2.6736e3)
    for k, v in vals.items():
        setattr(p, k, float(f(v)))
    for i, v in enumerate((9.0721, 106.7252, 61.5562, 0.0179, 0.0028, 0.0570, 0.3462, 2.3406, 0.7032, 1.0662)):
        p.p_emi[i] = v
    p.co2_flux = 298.0
    p.ipx, p.ipy, p.year0, p.dt, p.dt_crcl = 1, 1, 1940, 12 * 3600, 1800
    for k, v in over.items():
        if k == "p_emi":
            for i, x in enumerate(v):
                p.p_emi[i] = x
        else:
            setattr(p, k, v)
    return p


def fptr(a: np.ndarray):
    """float* of a C-contiguous fp32 array (caller keeps `a` alive)."""
    assert a.dtype == np.float32 and a.flags.c_contiguous, (a.dtype, a.flags)
    return a.ctypes.data_as(c_float_p)


def make_fields(inp) -> tuple[GrebFields, list]:
    """GrebFields pointing at a workload.Inputs; returns (struct, keepalive list)."""
    keep = []
    f = GrebFields()
    for name in ("z_topo", "glacier", "sw_solar", "tclim", "qclim", "uclim", "vclim", "mldclim",
                 "cldclim", "swetclim"):
        a = np.ascontiguousarray(getattr(inp, name), dtype=np.float32)
        keep.append(a)
        setattr(f, name, fptr(a))
    return f, keep


def nan_overrides(n: int):
    arr = (GrebMemberOverrides * n)()
    for o in arr:
        o.da_ice = o.a_no_ice = o.a_cloud = o.kappa = math.nan
    return arr

"""Python mirror of the C ABI (include/greb_engine.h) over libgreb_hip.so.

Host-side plumbing only: every number is produced by the HIP library.  There is no CPU
fallback -- if the library is missing or there is no GPU, calls raise GrebError.
"""
from __future__ import annotations

import ctypes as C
import json
import os

import numpy as np

from . import abi, build, workload

_lib = None


class GrebError(RuntimeError):
    def __init__(self, code: int, msg: str):
        super().__init__(f"greb engine error {code}: {msg}")
        self.code = code


_lib_path = build.LIB


def use_tuning_build() -> None:
    """tools/ only: load libgreb_hip_tuning.so (-DGREB_TUNING, the build in which the GREB_DEBUG_* / tile-size
    environment knobs exist) instead of the release library.  Must be called before the first engine call."""
    global _lib_path
    if _lib is not None:
        raise GrebError(-100, "use_tuning_build() after the library was loaded")
    _lib_path = build.LIB_TUNING


def lib() -> C.CDLL:
    """Load libgreb_hip.so (never builds implicitly on the GPU box; fails loudly if absent)."""
    global _lib
    if _lib is None:
        if not os.path.exists(_lib_path):
            raise GrebError(-100, f"{_lib_path} not built; run `python -c 'import __graft_entry__ as g; g.build()'`")
        L = C.CDLL(_lib_path)
        L.greb_engine_last_error.restype = C.c_char_p
        L.greb_engine_last_error.argtypes = [C.c_void_p]
        L.greb_device_info.restype = C.c_char_p
        L.greb_params_default.restype = None
        _lib = L
    return _lib


EXPORTS = ["greb_params_default", "greb_engine_create", "greb_engine_flux_correction", "greb_engine_run",
           "greb_engine_get_corrections", "greb_engine_set_corrections", "greb_engine_get_state",
           "greb_engine_last_error", "greb_engine_destroy", "greb_device_info", "greb_diffusion_batched",
           "greb_advection_batched", "greb_circulation_batched", "greb_diffusion_batched_dev",
           "greb_engine_point_physics", "greb_log_exp_switches", "greb_engine_set_experiment",
           "greb_ensemble_moments_dev", "greb_ensemble_quantiles_dev", "greb_engine_set_state", "greb_release_caches", "greb_diffusion_launch_order",
           "greb_substep_launch_order", "greb_circulation_launch_plan", "greb_engine_describe"]


def _check(rc: int, h=None):
    if rc != 0:
        msg = lib().greb_engine_last_error(h)
        raise GrebError(rc, msg.decode() if msg else "")


def device_info(device: int = 0) -> dict:
    return json.loads(lib().greb_device_info(device).decode())


def log_exp_switches(log_exp: int) -> int:
    """Process switches of the upstream variant's experiment number (greb.original.model.f90:60)."""
    f = lib().greb_log_exp_switches
    f.restype = C.c_uint
    return int(f(int(log_exp)))


def params_default() -> abi.GrebParams:
    p = abi.GrebParams()
    lib().greb_params_default(C.byref(p))
    return p


class Engine:
    """greb_engine_* handle.  Mirrors the reference's run structure: flux_correction() is
    qflux_correction (src/greb.f90:311-364), run() is the scenario loop (:228-234)."""

    def __init__(self, inp: workload.Inputs, params: abi.GrebParams | None = None, n_members: int = 1,
                 overrides=None, device: int = 0, strict: bool = False, multilaunch: bool = False,
                 row_strips: bool = False, persistent: bool | None = None):
        """persistent (384-wide grids): True = the circulation call as one launch (GREB_F_PERSISTENT), False = one launch per
        sub-step (GREB_F_NO_PERSISTENT), None = the engine times both and keeps the faster."""
        L = lib()
        self.params = params or params_default()
        self.nx, self.ny, self.np, self.nm = inp.nx, inp.ny, inp.nx * inp.ny, n_members
        fields, self._keep = abi.make_fields(inp)
        ov = None
        if overrides is not None:
            ov = (abi.GrebMemberOverrides * n_members)()
            for i, o in enumerate(overrides):
                for k in ("da_ice", "a_no_ice", "a_cloud", "kappa"):
                    setattr(ov[i], k, float(o.get(k, float("nan"))))
        self.h = C.c_void_p()
        rc = L.greb_engine_create(C.byref(self.params), inp.nx, inp.ny, C.byref(fields), n_members, ov, device,
                                  (abi.F_STRICT if strict else 0) | (abi.F_MULTILAUNCH if multilaunch else 0) |
                                 (abi.F_ROW_STRIPS if row_strips else 0) |
                                  (0 if persistent is None else (abi.F_PERSISTENT if persistent else abi.F_NO_PERSISTENT)),
                                  C.byref(self.h))
        if rc != 0:
            msg = L.greb_engine_last_error(self.h).decode()
            if self.h:
                L.greb_engine_destroy(self.h)
                self.h = None
            raise GrebError(rc, msg)

    def describe(self) -> dict:
        """greb_engine_describe: grid, members, arithmetic, which circulation form runs (and the trial's timings)."""
        f = lib().greb_engine_describe
        f.restype = C.c_char_p
        return json.loads(f(self.h).decode())

    def close(self):
        if getattr(self, "h", None):
            lib().greb_engine_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def flux_correction(self, years: int) -> np.ndarray:
        yearly = np.zeros((self.nm, max(years, 1), 2), np.float32)
        _check(lib().greb_engine_flux_correction(self.h, int(years), abi.fptr(yearly)), self.h)
        return yearly[:, :years]

    def run(self, years: int, co2_ppm, monthly_dev_ptr: int | None = None, out: np.ndarray | None = None):
        """co2_ppm: scalar, [years] or [n_members][years].  Returns (monthly, yearly); with
        monthly_dev_ptr (a device address) the monthly means stay on the GPU and monthly is None.
        out: caller-owned host buffer for the monthly means (float32, C-contiguous, n_members*years*12*5*np
        elements -- e.g. the numpy view of a pinned torch tensor, which makes the delivery a true DMA)."""
        co2 = np.ascontiguousarray(np.broadcast_to(np.asarray(co2_ppm, np.float32), (self.nm, years)))
        yearly = np.zeros((self.nm, years, 2), np.float32)
        if monthly_dev_ptr is None:
            shape = (self.nm, years, 12, 5, self.ny, self.nx)
            if out is None:
                monthly = np.empty(shape, np.float32)
            else:
                if out.dtype != np.float32 or not out.flags.c_contiguous or out.size != int(np.prod(shape)):
                    raise GrebError(-1, "run: `out` must be C-contiguous float32 with n_members*years*12*5*ny*nx elements")
                monthly = out.reshape(shape)
            _check(lib().greb_engine_run(self.h, int(years), abi.fptr(co2), abi.fptr(monthly), abi.fptr(yearly), 0), self.h)
            return monthly, yearly
        _check(lib().greb_engine_run(self.h, int(years), abi.fptr(co2), C.c_void_p(monthly_dev_ptr), abi.fptr(yearly),
                                     abi.RUN_DEVICE_OUT), self.h)
        return None, yearly

    def state(self, member: int = 0) -> np.ndarray:
        s = np.empty((5, self.ny, self.nx), np.float32)
        _check(lib().greb_engine_get_state(self.h, member, abi.fptr(s)), self.h)
        return s

    def get_corrections(self, member: int = 0):
        corr = np.empty((3, abi.NSTEP_YR, self.ny, self.nx), np.float32)
        st = np.empty((5, self.ny, self.nx), np.float32)
        _check(lib().greb_engine_get_corrections(self.h, member, abi.fptr(corr), abi.fptr(st)), self.h)
        return corr, st

    def set_corrections(self, corr, state5, member: int = -1):
        """Either argument may be None (left unchanged); member -1 = every member."""
        corr = None if corr is None else np.ascontiguousarray(corr, np.float32)
        state5 = None if state5 is None else np.ascontiguousarray(state5, np.float32)
        _check(lib().greb_engine_set_corrections(self.h, member, None if corr is None else abi.fptr(corr),
                                                 None if state5 is None else abi.fptr(state5)), self.h)

    def set_experiment(self, switches: int):
        """Sensitivity-experiment switches (abi.X_*; log_exp_switches() maps the original's log_exp)."""
        _check(lib().greb_engine_set_experiment(self.h, C.c_uint(int(switches))), self.h)

    def point_physics(self, ityr: int, co2: float, in5) -> np.ndarray:
        in5 = np.ascontiguousarray(in5, np.float32)
        out = np.empty((15, self.ny, self.nx), np.float32)
        _check(lib().greb_engine_point_physics(self.h, int(ityr), C.c_float(co2), abi.fptr(in5), abi.fptr(out)), self.h)
        return out


POINT_FIELDS = ("albedo", "sw", "LWsurf", "LWair_down", "em", "Q_sens", "Qlat", "Qlat_air", "dq_eva", "dq_rain",
                "dT_ocean", "dTo", "cap_surf_new")


def _batched(fn_name, params, arrays, strict, device):
    arrs = [np.ascontiguousarray(a, np.float32) for a in arrays]
    shape = arrs[0].shape
    if arrs[0].ndim == 2:
        arrs = [a[None] for a in arrs]
    b, ny, nx = arrs[0].shape
    out = np.empty((b, ny, nx), np.float32)
    p = params or params_default()
    fn = getattr(lib(), fn_name)
    _check(fn(C.byref(p), nx, ny, b, *[abi.fptr(a) for a in arrs], abi.fptr(out), int(bool(strict)), device))
    return out.reshape(shape)


def diffusion(T1, wz, params=None, strict=False, device=0):
    """Batched mirror of diffusion(T1,dX,h_scl,wz), src/greb.f90:556-723."""
    return _batched("greb_diffusion_batched", params, (T1, wz), strict, device)


def advection(T1, wz, u, v, params=None, strict=False, device=0):
    """Batched mirror of advection(T1,dX,h_scl,wz), src/greb.f90:726-915 (u, v: raw wind slice)."""
    return _batched("greb_advection_batched", params, (T1, wz, u, v), strict, device)


def circulation(X, wz, u, v, params=None, strict=False, device=0):
    """Batched mirror of circulation(X_in,dX,h_scl,wz), src/greb.f90:528-553."""
    return _batched("greb_circulation_batched", params, (X, wz, u, v), strict, device)


def diffusion_launch_order(params, nx, ny, batch):
    """Host-only diagnostic: the task list of the 384-wide diffusion sweep, arrays (field, k0, k1, up); empty when the
    grid does not take that kernel (include/greb_engine.h: greb_diffusion_launch_order)."""
    params = params if params is not None else params_default()
    f = lib().greb_diffusion_launch_order
    n = f(C.byref(params), nx, ny, batch, None, None, None, None, 0)
    if n < 0:
        _check(n)
    out = [np.zeros(n, np.int32) for _ in range(4)]
    if n:
        ptr = [a.ctypes.data_as(C.POINTER(C.c_int)) for a in out]
        assert f(C.byref(params), nx, ny, batch, *ptr, n) == n
    return out


def substep_launch_order(params, nx, ny, n_members, kappa=None):
    """Host-only diagnostic: the task list of the engine's row-strip circulation sub-step, arrays (field, k0, k1) with
    field = 2 * member + tracer (include/greb_engine.h: greb_substep_launch_order)."""
    params = params if params is not None else params_default()
    kap = None if kappa is None else np.ascontiguousarray(kappa, np.float32)
    kp = None if kap is None else kap.ctypes.data_as(C.POINTER(C.c_float))
    f = lib().greb_substep_launch_order
    n = f(C.byref(params), nx, ny, n_members, kp, None, None, None, 0)
    if n < 0:
        _check(n)
    out = [np.zeros(n, np.int32) for _ in range(3)]
    if n:
        ptr = [a.ctypes.data_as(C.POINTER(C.c_int)) for a in out]
        assert f(C.byref(params), nx, ny, n_members, kp, *ptr, n) == n
    return out


def circulation_launch_plan(params, nx, ny, n_members, kappa=None, slots=2048):
    """Host-only diagnostic: the tasks of the one-launch circulation call for `slots` wavefront slots: arrays
    (field, k0, k1, chain, dep[n][4]) (include/greb_engine.h: greb_circulation_launch_plan)."""
    params = params if params is not None else params_default()
    kap = None if kappa is None else np.ascontiguousarray(kappa, np.float32)
    kp = None if kap is None else kap.ctypes.data_as(C.POINTER(C.c_float))
    f = lib().greb_circulation_launch_plan
    n = f(C.byref(params), nx, ny, n_members, kp, slots, None, None, None, None, None, 0)
    if n < 0:
        _check(n)
    out = [np.zeros(n, np.int32) for _ in range(4)] + [np.zeros((n, 4), np.int32)]
    if n:
        ptr = [a.ctypes.data_as(C.POINTER(C.c_int)) for a in out]
        assert f(C.byref(params), nx, ny, n_members, kp, slots, *ptr, n) == n
    return out


def diffusion_dev(params, nx, ny, batch, T1_ptr, wz_ptr, dX_ptr, strict=False, sweeps=1, stream=0):
    """Device-pointer diffusion sweeps for the roofline bench (no sync)."""
    _check(lib().greb_diffusion_batched_dev(C.byref(params), nx, ny, batch, C.c_void_p(T1_ptr), C.c_void_p(wz_ptr),
                                            C.c_void_p(dX_ptr), int(bool(strict)), int(sweeps), C.c_void_p(stream)))

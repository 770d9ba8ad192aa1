"""Build libgreb_hip.so (the C-ABI engine) in-tree with hipcc for gfx950.

hipcc cross-compiles without a GPU, so this runs in the build container; the resulting .so is
git-ignored but travels to the GPU box with the snapshot.
"""
from __future__ import annotations

import os
import shutil
import subprocess

PKG = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(PKG)
CSRC = os.path.join(PKG, "csrc")
LIB = os.path.join(PKG, "libgreb_hip.so")
# the same sources with -DGREB_TUNING: the timing-experiment knobs of tools/ (GREB_DEBUG_SKIP, GREB_DEBUG_NSUB, ...)
# exist only in this variant; the release library above never reads the environment
LIB_TUNING = os.path.join(PKG, "libgreb_hip_tuning.so")
SOURCES = ["greb_engine.cpp", "greb_kernels.hip", "greb_member.hip", "greb_ensemble.hip", "greb_rows.hip", "greb_step_rows.hip", "greb_circ_rows.hip"]
HEADERS = ["greb_device.h", "greb_kernels.h", "greb_stencil.h", "greb_chain6.h", "greb_rows.h", "greb_step_strip.h", "greb_step_order.h", "greb_pair.h", "greb_physics_step.h", os.path.join(ROOT, "include", "greb_engine.h")]
# -O2: measured 1.4 % faster than -O3 on the fused member kernel (3 790 vs 3 735 yr/s), equal elsewhere
HIPCC_FLAGS = ["--offload-arch=gfx950", "-O2", "-std=c++17", "-fPIC", "-shared", "-Wno-unused-value",
               "-I" + os.path.join(ROOT, "include")]


def hipcc() -> str:
    return shutil.which("hipcc") or "/opt/rocm/bin/hipcc"


# per-source extra flags.  greb_rows.hip: the SLP vectoriser packs the scalar stencil of the row-strip kernel into
# v_pk_* instructions whose halves it then has to shuffle together (319 v_mov against 113, 156 VGPRs against 120: one
# wave per SIMD less); a packed fp32 instruction occupies the pipe as long as its two halves would, so nothing is gained.
EXTRA_FLAGS = {"greb_rows.hip": ["-fno-slp-vectorize"], "greb_step_rows.hip": ["-fno-slp-vectorize"],
               "greb_circ_rows.hip": ["-fno-slp-vectorize"]}
OBJ_DIR = os.path.join(PKG, "csrc", "_obj")


def _deps(src: str) -> list[str]:
    return [os.path.join(CSRC, src)] + [h if os.path.isabs(h) else os.path.join(CSRC, h) for h in HEADERS]


def needs_build(lib: str = LIB) -> bool:
    if not os.path.exists(lib):
        return True
    t = os.path.getmtime(lib)
    return any(os.path.getmtime(d) > t for s in SOURCES for d in _deps(s)) or os.path.getmtime(__file__) > t


def build_lib(force: bool = False, verbose: bool = False, tuning: bool = False) -> str:
    """One object per source (each with its own flags, rebuilt when the source, a header or this script is newer),
    then one link.  No relocatable device code is needed: every device function lives in a header."""
    lib = LIB_TUNING if tuning else LIB
    if not force and not needs_build(lib):
        return lib
    os.makedirs(OBJ_DIR, exist_ok=True)
    flags = [f for f in HIPCC_FLAGS if f != "-shared"] + (["-DGREB_TUNING"] if tuning else [])
    objs = []
    for src in SOURCES:
        obj = os.path.join(OBJ_DIR, os.path.splitext(src)[0] + ("_tuning.o" if tuning else ".o"))
        objs.append(obj)
        newest = max(max(os.path.getmtime(d) for d in _deps(src)), os.path.getmtime(__file__))
        if not force and os.path.exists(obj) and os.path.getmtime(obj) > newest:
            continue
        cmd = [hipcc(), *flags, *EXTRA_FLAGS.get(src, []), "-c", os.path.join(CSRC, src), "-o", obj]
        if verbose:
            print(" ".join(cmd))
        subprocess.run(cmd, check=True, cwd=CSRC)
    cmd = [hipcc(), "--offload-arch=gfx950", "-shared", "-fPIC", "-o", lib, *objs]
    if verbose:
        print(" ".join(cmd))
    subprocess.run(cmd, check=True, cwd=CSRC)
    return lib


LIB_ASAN = os.path.join(PKG, "libgreb_hip_asan.so")


def build_lib_asan(verbose: bool = False) -> str:
    """The library with HOST-side AddressSanitizer + UBSan (-fno-gpu-sanitize: device code as in the release build; GPU
    sanitizers are not available on this pool and nothing here runs on a GPU): for tests/test_sanitizers_cpu.py, which
    drives the host-only entry points (launch orders, launch plan, argument checks) through it.  Never loaded by the
    product; stays in the build container (.gpurunignore)."""
    from concurrent.futures import ThreadPoolExecutor
    if os.path.exists(LIB_ASAN):
        t = os.path.getmtime(LIB_ASAN)
        if not (any(os.path.getmtime(d) > t for s in SOURCES for d in _deps(s)) or os.path.getmtime(__file__) > t):
            return LIB_ASAN
    os.makedirs(OBJ_DIR, exist_ok=True)
    san = ["-fsanitize=address,undefined", "-fno-gpu-sanitize", "-fno-omit-frame-pointer", "-shared-libsan", "-g"]
    flags = [f for f in HIPCC_FLAGS if f != "-shared"] + san

    def one(src):
        obj = os.path.join(OBJ_DIR, os.path.splitext(src)[0] + "_asan.o")
        cmd = [hipcc(), *flags, *EXTRA_FLAGS.get(src, []), "-c", os.path.join(CSRC, src), "-o", obj]
        if verbose:
            print(" ".join(cmd))
        subprocess.run(cmd, check=True, cwd=CSRC, capture_output=not verbose)
        return obj

    with ThreadPoolExecutor(max_workers=4) as ex:
        objs = list(ex.map(one, SOURCES))
    subprocess.run([hipcc(), "--offload-arch=gfx950", "-shared", "-fPIC", *san, "-o", LIB_ASAN, *objs], check=True, cwd=CSRC,
                   capture_output=not verbose)
    return LIB_ASAN


def asan_runtime() -> str | None:
    """clang's shared ASan runtime, to be LD_PRELOADed into an uninstrumented python that loads LIB_ASAN."""
    import glob
    hits = sorted(glob.glob("/opt/rocm/lib/llvm/lib/clang/*/lib/linux/libclang_rt.asan-x86_64.so"))
    return hits[-1] if hits else None


def build_host(verbose: bool = False) -> str | None:
    """The thin Fortran hosts over iso_c_binding: greb_host (src/greb.f90's shell) and greb_host_original (the
    upstream variant's shell with the log_exp experiments) -> greb_climate_model_amd/.  Returns greb_host's
    path, or None when no Fortran compiler is present."""
    fc = shutil.which("amdflang") or ("/opt/rocm/bin/amdflang" if os.path.exists("/opt/rocm/bin/amdflang") else None)
    api = os.path.join(PKG, "host", "greb_c_api.f90")
    if fc is None or not os.path.exists(api):
        return None
    moddir = os.path.join(PKG, "host", "_mod")
    os.makedirs(moddir, exist_ok=True)
    first = None
    for name in ("greb_host", "greb_host_original"):
        src = os.path.join(PKG, "host", name + ".f90")
        out = os.path.join(PKG, name)
        first = first or out
        if os.path.exists(out) and os.path.getmtime(out) > max(os.path.getmtime(src), os.path.getmtime(api), os.path.getmtime(LIB)):
            continue
        cmd = [fc, "-O2", "-module-dir", moddir, "-o", out, api, src, "-L" + PKG, "-lgreb_hip", "-Wl,-rpath," + PKG,
               "-Wl,-rpath,$ORIGIN"]
        if verbose:
            print(" ".join(cmd))
        subprocess.run(cmd, check=True)
    return first


def build_c_example(verbose: bool = False) -> str | None:
    """examples/greb_run.c (the ABI from plain C, gcc) -> greb_climate_model_amd/greb_run_c."""
    cc = shutil.which("gcc") or shutil.which("cc")
    src = os.path.join(ROOT, "examples", "greb_run.c")
    if cc is None or not os.path.exists(src):
        return None
    out = os.path.join(PKG, "greb_run_c")
    if os.path.exists(out) and os.path.getmtime(out) > max(os.path.getmtime(src), os.path.getmtime(LIB)):
        return out
    cmd = [cc, "-std=c11", "-O2", "-Wall", "-Wextra", src, "-I" + os.path.join(ROOT, "include"), "-L" + PKG, "-lgreb_hip",
           "-Wl,-rpath," + PKG, "-Wl,-rpath,$ORIGIN", "-o", out]
    if verbose:
        print(" ".join(cmd))
    subprocess.run(cmd, check=True)
    return out


if __name__ == "__main__":
    print(build_lib(force=True, verbose=True))
    print(build_host(verbose=True))
    print(build_c_example(verbose=True))

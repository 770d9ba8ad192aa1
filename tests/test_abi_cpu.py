"""CPU: the C-ABI library builds, loads and exports every symbol include/greb_engine.h declares;
host-side logic that needs no GPU.  No compute call is made here."""
import ctypes as C
import os
import re

import numpy as np
import pytest

from greb_climate_model_amd import abi, build, engine

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def lib():
    build.build_lib()
    return engine.lib()


def test_header_symbols_exported(lib):
    hdr = open(os.path.join(ROOT, "include", "greb_engine.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    names = set(re.findall(r"\b(greb_[a-z0-9_]+)\s*\(", hdr))
    assert names, "no prototypes parsed"
    assert names == set(engine.EXPORTS), names ^ set(engine.EXPORTS)
    for n in names:
        assert hasattr(lib, n), n


def test_params_struct_layout_and_defaults(lib):
    p = engine.params_default()
    q = abi.default_params()
    assert C.sizeof(abi.GrebParams) == (28 + 10 + 1) * 4 + 5 * 4
    for n in abi.GrebParams.PHYSICS_NAMES + ("co2_flux", "ipx", "ipy", "year0", "dt", "dt_crcl"):
        assert getattr(p, n) == getattr(q, n), n
    assert list(p.p_emi) == list(q.p_emi)
    # folded constants (src/greb.f90:86-89,94)
    assert np.float32(p.To_ice2) == np.float32(273.15) - np.float32(1.7)
    assert np.float32(p.cq_rain) == np.float32(np.float32(-0.1) / np.float32(24.0)) / np.float32(3600.0)


def test_no_gpu_is_a_loud_error_not_a_fallback(lib, inputs):
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    with pytest.raises(engine.GrebError) as ei:
        engine.Engine(inputs)
    assert ei.value.code == -2 and "no CPU path" in str(ei.value)
    with pytest.raises(engine.GrebError) as ei:
        engine.diffusion(inputs.tclim[0], inputs.tclim[0])
    assert ei.value.code == -2


def test_bad_arguments_rejected(lib, inputs):
    p = engine.params_default()
    out = C.c_void_p()
    f, keep = abi.make_fields(inputs)
    assert lib.greb_engine_create(C.byref(p), 95, 48, C.byref(f), 1, None, 0, 0, C.byref(out)) == -1
    assert lib.greb_engine_create(C.byref(p), 96, 48, C.byref(f), 0, None, 0, 0, C.byref(out)) == -1
    assert lib.greb_engine_run(None, 1, None, None, None, 0) == -1
    assert lib.greb_engine_destroy(None) == 0


def test_product_does_not_touch_oracle():
    """The product path must not import, link or load anything under oracle/."""
    pkg = os.path.join(ROOT, "greb_climate_model_amd")
    pats = [r"(from|import)\s+oracle", r"liboracle", r"oracle/", r"#include\s+\".*oracle", r"greb_oracle"]
    for dirpath, _, files in os.walk(pkg):
        for fn in files:
            if fn.endswith((".py", ".cpp", ".hip", ".h", ".f90")):
                txt = open(os.path.join(dirpath, fn)).read()
                for pat in pats:
                    assert not re.search(pat, txt), (fn, pat)
    import subprocess
    out = subprocess.run(["ldd", build.LIB], capture_output=True, text=True).stdout
    assert "oracle" not in out


def test_release_library_never_reads_the_environment(lib):
    """Timing-experiment knobs (GREB_DEBUG_SKIP, GREB_DEBUG_NSUB, tile sizes) are compiled in only under
    -DGREB_TUNING (libgreb_hip_tuning.so, tools/): the release library does not import getenv at all, so a stray
    variable cannot change the physics."""
    import subprocess
    und = subprocess.run(["nm", "-D", "--undefined-only", build.LIB], capture_output=True, text=True, check=True).stdout
    assert "getenv" not in und
    raw = open(build.LIB, "rb").read()
    for knob in (b"GREB_DEBUG_SKIP", b"GREB_DEBUG_NSUB", b"GREB_BAND_LIMIT_KB", b"GREB_NO_STREAM", b"GREB_STREAM_WGS",
                 b"GREB_PAIR_ROWS"):
        assert knob not in raw, knob

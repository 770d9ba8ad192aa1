"""CPU sanitizer legs (the reference's own debug build is `debug=1` -> bounds and FP checks,
/root/reference/Makefile:10).  GPU sanitizers do not exist on this pool; what CAN be checked is everything that runs on
the host:
  * the oracle's C restatement under ASan + UBSan (gcc), against the reference-minted golden vectors;
  * the engine library's host side under ASan + UBSan (hipcc -fsanitize=address,undefined -fno-gpu-sanitize): the
    launch-order builders of the row-strip kernels, the one-launch circulation plan with its dependency table, the
    ABI's argument checks -- the code that sizes every device allocation and every grid.
Each leg is a child pytest with the sanitizer runtime preloaded; a report fails the child (exit code != 0)."""
import os
import shutil
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(env_extra, args, timeout):
    env = dict(os.environ)
    env.update(env_extra)
    env["ASAN_OPTIONS"] = "detect_leaks=0:abort_on_error=0:exitcode=86"  # (CPython itself leaks by design)
    env["UBSAN_OPTIONS"] = "halt_on_error=1:print_stacktrace=1:exitcode=87"
    r = subprocess.run([sys.executable, "-m", "pytest", "-x", "-q", "-p", "no:cacheprovider", *args], cwd=ROOT, env=env,
                       capture_output=True, text=True, timeout=timeout)
    tail = (r.stdout + r.stderr)[-3000:]
    assert r.returncode == 0, tail
    assert "passed" in r.stdout and "ERROR: AddressSanitizer" not in tail and "runtime error:" not in tail, tail
    return r.stdout


def test_oracle_restatement_under_asan_ubsan():
    gcc = shutil.which("gcc")
    if not gcc:
        pytest.skip("no gcc")
    libasan = subprocess.run([gcc, "-print-file-name=libasan.so"], capture_output=True, text=True).stdout.strip()
    if not os.path.isabs(libasan) or not os.path.exists(libasan):
        pytest.skip("no libasan")
    subprocess.run(["make", "-s", "-C", os.path.join(ROOT, "oracle"), "asan"], check=True)
    so = os.path.join(ROOT, "oracle", "liboracle_greb_asan.so")
    # every routine at three times of year + a 1+2-yr run at 96x48 and the stencils at 384x192 (incl. the 1 800-sweep rows):
    # every array bound of the restatement is walked (~80 s); the long runs are left to the plain build
    out = _run({"GREB_ORACLE_SO": so, "LD_PRELOAD": libasan},
               ["tests/test_oracle_golden.py", "-k", "routines_bit_exact or run_short or grid_tables or toy_known"], 1500)
    assert " passed" in out


def test_engine_host_side_under_asan_ubsan():
    from greb_climate_model_amd import build
    rt = build.asan_runtime()
    if rt is None or not shutil.which("hipcc") and not os.path.exists("/opt/rocm/bin/hipcc"):
        pytest.skip("no clang ASan runtime / hipcc")
    lib = build.build_lib_asan()
    boot = ("import sys, pytest; from greb_climate_model_amd import engine; engine._lib_path = %r; "
            "sys.exit(pytest.main(['-x', '-q', '-p', 'no:cacheprovider', 'tests/test_rows_order_cpu.py', 'tests/test_abi_cpu.py', "
            "'-k', 'not release_library and not product_does_not']))" % lib)
    env = dict(os.environ)
    env.update({"LD_PRELOAD": rt, "ASAN_OPTIONS": "detect_leaks=0:exitcode=86", "UBSAN_OPTIONS": "halt_on_error=1:print_stacktrace=1:exitcode=87"})
    r = subprocess.run([sys.executable, "-c", boot], cwd=ROOT, env=env, capture_output=True, text=True, timeout=1500)
    tail = (r.stdout + r.stderr)[-3000:]
    assert r.returncode == 0 and " passed" in r.stdout, tail
    assert "ERROR: AddressSanitizer" not in tail and "runtime error:" not in tail, tail

"""The counter records bench.py quotes (profiles/r04_*.json) were collected on the kernels this tree builds: each record
carries the sha256 of the profiled kernel's gfx950 machine code (greb_climate_model_amd/codesha.py); a kernel edit after
the counter passes makes this fail until tools/verify_round.sh profiles has been run again."""
import json
import os

import pytest

from greb_climate_model_amd import build, codesha

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _bench_files():
    import importlib.util
    spec = importlib.util.spec_from_file_location("bench_mod", os.path.join(ROOT, "bench.py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return [m.TRAFFIC_FILE, m.G384_DIF_FILE, m.G384_STEP_FILE]


def test_code_hash_is_of_the_kernel_and_nothing_else():
    lib = build.build_lib()
    a = codesha.kernel_functions(lib, "diffusion_stream_kernelILb0ELi96ELi48E")
    assert len(a) == 1 and len(next(iter(a.values()))) > 4096  # one kernel, real code
    assert codesha.code_sha(lib, "diffusion_stream_kernelILb0ELi96ELi48E") != codesha.code_sha(lib, "diffusion_stream_kernelILb1ELi96ELi48E")
    assert codesha.code_sha(lib, "no_such_kernel") is None
    rec = {"code": codesha.record(lib, "dif_rows_kernelILb0E")}
    assert codesha.source_check(rec, "x.json", lib)["matches_loaded_library"] is True
    rec["code"]["sha256"] = "0" * 64
    assert codesha.source_check(rec, "x.json", lib)["matches_loaded_library"] is False
    assert codesha.source_check(None, "x.json", lib)["matches_loaded_library"] is None


@pytest.mark.parametrize("name", _bench_files())
def test_committed_counter_records_belong_to_the_built_kernels(name):
    path = os.path.join(ROOT, "profiles", name)
    assert os.path.exists(path), f"{name}: bench.py quotes it; run tools/verify_round.sh profiles"
    rec = json.load(open(path))
    assert "code" in rec and rec["code"]["sha256"], name
    got = codesha.code_sha(build.build_lib(), rec["code"]["kernel_symbol_contains"])
    assert got == rec["code"]["sha256"], (f"{name} was collected on other code of {rec['code']['kernel_symbol_contains']}: "
                                          "re-run the counter passes (tools/verify_round.sh profiles)")

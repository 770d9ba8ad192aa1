"""CPU: the oracle's log_exp sensitivity-experiment switches against the golden vectors minted
from the upstream model variant compiled in place (tests/golden/make_golden_logexp.py).  Bit-exact."""
import hashlib
import json
import os

import numpy as np
import pytest

from conftest import GOLDEN, load_golden

PINNED = (5, 6, 8, 9, 10, 11, 12, 13, 14, 15)


def original_params():
    from greb_climate_model_amd import abi
    return abi.default_params(cp_land=float(np.float32(4186.0) / np.float32(4.5)))  # greb.original.model.f90:69


def test_manifest_lists_the_pin():
    m = json.load(open(os.path.join(GOLDEN, "MANIFEST.json")))
    it = m["items"]["logexp_g96"]
    assert tuple(it["log_exp"]) == PINNED and all(it["oracle_bit_identical"][str(k)] for k in PINNED)
    assert it["unpinned"] == [1, 2, 3, 4, 7, 16]  # the original reads an unassigned dX_crcl there


@pytest.mark.parametrize("log_exp", PINNED)
def test_oracle_reproduces_original_variant(inputs, oracle_lib, log_exp):
    g = load_golden("logexp_g96.npz")
    o = oracle_lib.Oracle(inputs, original_params())
    ctrl, scen = o.run_original(log_exp, 1, 1, 2)
    o.close()
    scen = scen.reshape(24, 5, 48, 96)
    k = f"le{log_exp:02d}"
    assert hashlib.sha256(np.ascontiguousarray(scen).tobytes()).digest() == g[k + "_sha256"].tobytes()
    assert np.array_equal(scen[-1], g[k + "_scen_last"])
    assert np.array_equal(ctrl.reshape(12, 5, 48, 96)[-1], g[k + "_ctrl_last"])


def test_switches_change_the_climate(inputs):
    """The experiments are not no-ops: every pinned one differs from the complete model (10) somewhere."""
    g = load_golden("logexp_g96.npz")
    base = g["le10_scen_stats"]
    for le in PINNED:
        if le != 10:
            assert np.abs(g[f"le{le:02d}_scen_stats"] - base).max() > 1e-3, le


def test_co2_ramp_matches_a1b_formula(inputs, oracle_lib):
    """co2_level of the original (greb.original.model.f90:939-951) for log_exp 12/13; 680 ppm otherwise."""
    o = oracle_lib.Oracle(inputs, original_params())
    assert o.co2_level(10, 1990.0) == 680.0
    assert o.co2_level(12, 1950.0) == 310.0 and o.co2_level(13, 2000.0) == 370.0
    assert o.co2_level(12, 2050.0) == 520.0 and o.co2_level(12, 2100.0) == 700.0 and o.co2_level(12, 2101.0) == 680.0
    assert abs(o.co2_level(12, 1940.0) - 298.0) < 1e-4
    o.close()


def test_host_side_experiment_helpers_match_the_oracle(inputs, oracle_lib):
    """greb_climate_model_amd/original.py (host plumbing of the engine, no GPU needed): the A1B CO2 series and the
    boundary-data changes of every experiment equal the oracle's restatement of greb.original.model.f90."""
    from greb_climate_model_amd import original
    o = oracle_lib.Oracle(inputs, original_params())
    for le in (10, 12, 13):
        for year in (1940.0, 1950.0, 1999.0, 2000.0, 2001.0, 2050.0, 2075.0, 2100.0, 2101.0):
            assert original.co2_level(le, year) == o.co2_level(le, year), (le, year)
    o.close()
    for le in (1, 2, 3, 5, 9, 10, 11, 14):
        mod = original.experiment_inputs(inputs, le)
        oo = oracle_lib.Oracle(inputs, original_params())
        oo.set_log_exp(le)
        # the oracle exposes its (modified) derived fields: initial q = qclim(last), wz_air from z_topo, cap_surf from mldclim(1)
        assert np.array_equal(oo.field(3), mod.qclim[-1]), le
        assert np.array_equal(oo.field(5), np.exp(-mod.z_topo / np.float32(8400.0)).astype(np.float32)) or \
            np.abs(oo.field(5) - np.exp(-mod.z_topo.astype(np.float64) / 8400.0)).max() < 1e-6, le
        if le <= 2:
            assert float(mod.cldclim.min()) == float(mod.cldclim.max()) == float(np.float32(0.7))
        if le <= 9 or le == 11:
            assert float(mod.mldclim.min()) == float(mod.mldclim.max()) == 50.0
        else:
            assert np.array_equal(mod.mldclim, inputs.mldclim)
        oo.close()
    assert original.experiment_inputs(inputs, 10).tclim is inputs.tclim  # untouched fields are shared, not copied

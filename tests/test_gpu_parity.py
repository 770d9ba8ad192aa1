"""GPU parity tests: the HIP path (through the C ABI) against the oracle and the golden vectors
minted from the compiled reference.

Bars:
  * STRICT stencils (diffusion / advection / circulation): BIT-EXACT -- they contain only
    + - * / in the reference's order, IEEE division, no FMA.
  * FAST stencils: increments differ by re-association only; tolerance 2e-6 relative to the
    largest increment of the field (fp32 eps = 6e-8, ~40 operations).
  * point physics: OCML expf/logf vs glibc -> <= 4 ulp on the affected fluxes, exact elsewhere.
  * whole runs, monthly means vs the reference Fortran: north-star tolerance < 1e-4 K RMS for
    Tsurf/Tair/Tocean; companions q < 2e-8, albedo < 1e-6 (the reference's own
    compiler-flag noise floor is 1.7e-5 K).
"""
import numpy as np
import pytest

import os

from conftest import GOLDEN, fast_global_mean_is_the_better_one, load_golden, rms, yearly_close

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def eng_mod():
    import torch
    assert torch.cuda.is_available(), "GPU tests need a GPU"
    from greb_climate_model_amd import engine
    engine.lib()  # fails loudly if libgreb_hip.so is missing
    return engine


def _fields(inputs, oracle_lib, params, ityr, seed=3):
    o = oracle_lib.Oracle(inputs, params)
    rng = np.random.default_rng(seed)
    f = np.float32
    Ta = (inputs.tclim[ityr - 1] + (1.5 * rng.standard_normal((48, 96))).astype(f)).astype(f)
    q = (inputs.qclim[ityr - 1] * (f(0.9) + f(0.2) * rng.random((48, 96)).astype(f))).astype(f)
    return o, Ta, q, o.field(5).copy(), o.field(6).copy(), inputs.uclim[ityr - 1], inputs.vclim[ityr - 1]


# ------------------------------------------------------------------------------------ stencils
@pytest.mark.parametrize("ityr", [1, 365, 730])
def test_strict_stencils_bit_exact_vs_golden(eng_mod, params, routine_golden, inputs, ityr):
    g = routine_golden
    Ta, q = g[f"t{ityr}_in_Ta"], g[f"t{ityr}_in_q"]
    wa, wv = g["wz_air"], g["wz_vapor"]
    u, v = inputs.uclim[ityr - 1], inputs.vclim[ityr - 1]
    X = np.stack([Ta, q]); W = np.stack([wa, wv]); U = np.stack([u, u]); V = np.stack([v, v])
    d = eng_mod.diffusion(X, W, params, strict=True)
    assert np.array_equal(d[0], g[f"t{ityr}_out_dif_Ta"]) and np.array_equal(d[1], g[f"t{ityr}_out_dif_q"])
    a = eng_mod.advection(X, W, U, V, params, strict=True)
    assert np.array_equal(a[0], g[f"t{ityr}_out_adv_Ta"]) and np.array_equal(a[1], g[f"t{ityr}_out_adv_q"])
    c = eng_mod.circulation(X, W, U, V, params, strict=True)
    assert np.array_equal(c[0], g[f"t{ityr}_out_crc_Ta"]) and np.array_equal(c[1], g[f"t{ityr}_out_crc_q"])


@pytest.mark.parametrize("ityr", [1, 365, 730])
def test_fast_stencils_within_reassociation_tolerance(eng_mod, params, routine_golden, inputs, ityr):
    g = routine_golden
    Ta, q = g[f"t{ityr}_in_Ta"], g[f"t{ityr}_in_q"]
    X = np.stack([Ta, q]); W = np.stack([g["wz_air"], g["wz_vapor"]])
    u, v = inputs.uclim[ityr - 1], inputs.vclim[ityr - 1]
    U = np.stack([u, u]); V = np.stack([v, v])
    for name, got in (("dif", eng_mod.diffusion(X, W, params)), ("adv", eng_mod.advection(X, W, U, V, params)),
                      ("crc", eng_mod.circulation(X, W, U, V, params))):
        for i, tr in enumerate(("Ta", "q")):
            ref = g[f"t{ityr}_out_{name}_{tr}"]
            # rows 1-10/39-48 return fl(fl(T+d)-T): quantised to ulp(T); allow one ulp of the state there
            tol = 2e-6 * np.abs(ref).max() + (np.spacing(np.abs(X[i]).max()) if name != "crc" else 24 * np.spacing(np.abs(X[i]).max()))
            assert np.abs(got[i].astype(np.float64) - ref).max() <= tol, (name, tr)


def test_stencil_edge_cases_strict(eng_mod, params, inputs, oracle_lib):
    """Cases the smooth fixtures never reach: the clamp (src/greb.f90:715,907) on a tracer with
    zeros and spikes, u<0 in sub-cycled rows (the :881 index bug), rough weights, a constant field."""
    o, Ta, q, wa, wv, u, v = _fields(inputs, oracle_lib, params, 100)
    rng = np.random.default_rng(11)
    f = np.float32
    spiky = (q * (rng.random((48, 96)) < 0.7)).astype(f)            # 30 % exact zeros
    spiky[0, ::7] = f(0.05); spiky[47, 3::5] = f(0.08); spiky[5, 90:] = f(0.03)   # polar spikes -> clamp
    rough_w = (f(0.05) + rng.random((48, 96)).astype(f) * f(0.95)).astype(f)
    west = (-np.abs(u) - f(3)).astype(f)                             # u < 0 everywhere
    const = np.full((48, 96), f(281.5))
    cases = [(spiky, wv, u, v), (spiky, rough_w, west, v), (Ta, rough_w, west, (-v).astype(f)), (const, wa, u, v),
             (Ta, wa, np.zeros_like(u), np.zeros_like(v))]
    for X, W, U, V in cases:
        assert np.array_equal(eng_mod.diffusion(X, W, params, strict=True), o.diffusion(X, W))
        assert np.array_equal(eng_mod.advection(X, W, U, V, params, strict=True), o.advection(X, W, u=U, v=V))
        assert np.array_equal(eng_mod.circulation(X, W, U, V, params, strict=True), o.circulation(X, W, u=U, v=V))
        # the FAST arithmetic must take the same clamp decisions except at exact ties
        df = eng_mod.diffusion(X, W, params)
        dr = o.diffusion(X, W)
        assert np.abs(df.astype(np.float64) - dr).max() <= 4e-6 * max(np.abs(dr).max(), 1e-30) + np.spacing(np.abs(X).max())
        # ... and so must the fused engine's FAST circulation, whose polar rows are the four-chains-in-one-wave form
        # (greb_member.hip: quad_chain_substep) with the clamp decided by the min test of greb_chain6.h: 24 sub-steps
        cf = eng_mod.circulation(X, W, U, V, params)
        cr = o.circulation(X, W, u=U, v=V)
        err = np.abs(cf.astype(np.float64) - cr)
        assert err.max() <= 1e-5 * max(np.abs(cr).max(), 1e-30) + 24 * np.spacing(np.abs(X).max()), err.max()
    o.close()


def test_batch_independence(eng_mod, params, inputs, oracle_lib):
    """Every batch item is an independent field: a batch of 37 ragged-count fields equals 37 single calls."""
    o, Ta, q, wa, wv, u, v = _fields(inputs, oracle_lib, params, 200)
    rng = np.random.default_rng(5)
    X = np.stack([(Ta + np.float32(i)).astype(np.float32) for i in range(37)])
    W = np.stack([wa if i % 2 else wv for i in range(37)])
    d = eng_mod.diffusion(X, W, params, strict=True)
    for i in (0, 1, 17, 36):
        assert np.array_equal(d[i], o.diffusion(X[i], W[i]))
    o.close()


def test_diffusion_g384_strict(eng_mod, params, oracle_lib):
    """384x192: every row sub-cycled, up to 225 sweeps in row 2, time2=1/ccx2=0 in the polar rows
    (SURVEY.md App. B).  Oracle at the same grid."""
    from greb_climate_model_amd import workload
    inp = workload.make_inputs(384, 192)
    o = oracle_lib.Oracle(inp, params)
    g = o.grid()
    assert g["dif_time2"][1] == 225 and g["dif_time2"][0] == 1 and g["dif_ccx2"][0] == 0.0
    X = np.stack([inp.tclim[10], inp.qclim[400]])
    W = np.stack([o.field(5), o.field(6)])
    d = eng_mod.diffusion(X, W, params, strict=True)
    assert np.array_equal(d[0], o.diffusion(X[0], W[0])) and np.array_equal(d[1], o.diffusion(X[1], W[1]))
    a = eng_mod.advection(X, W, np.stack([inp.uclim[10]] * 2), np.stack([inp.vclim[10]] * 2), params, strict=True)
    assert np.array_equal(a[0], o.advection(X[0], W[0], ityr=11))
    o.close()


def test_row_strip_diffusion_ragged_batches_and_other_tables(eng_mod, oracle_lib, inputs384):
    """The 384-wide row-strip diffusion kernel (greb_rows.hip) away from the benchmark shape: batches that do not fill
    the launch order's groups of eight fields (1, 3, 11 fields), and another diffusivity (kappa = 6.5e5: other sub-cycle
    tables -- 1 800 sweeps in the polar rows, 277 in the next -- hence other strips).  STRICT bit-exact against the oracle,
    FAST within the re-association tolerance, every field independent of its batch neighbours."""
    from greb_climate_model_amd import abi
    nx, ny, inp = 384, 192, inputs384
    for kappa, batches in ((None, (1, 3, 11)), (6.5e5, (5,))):
        p = abi.default_params()
        if kappa is not None:
            p.kappa = kappa
        o = oracle_lib.Oracle(inp, p)
        wa, wv = o.field(5).copy(), o.field(6).copy()
        for nb in batches:
            X = np.stack([(inp.tclim[7 * i % 730] + np.float32(0.25 * i)).astype(np.float32) if i % 2 == 0 else inp.qclim[31 * i % 730] for i in range(nb)])
            W = np.stack([wa if i % 2 == 0 else wv for i in range(nb)])
            ds = eng_mod.diffusion(X, W, p, strict=True)
            df = eng_mod.diffusion(X, W, p)
            for i in range(nb):
                ref = o.diffusion(X[i], W[i])
                assert np.array_equal(ds[i], ref), (nx, ny, nb, i)
                err = np.abs(df[i].astype(np.float64) - ref)
                assert err.max() <= 4e-6 * max(np.abs(ref).max(), 1e-30) + 2 * np.spacing(np.abs(X[i]).max()), (nx, ny, nb, i, err.max())
        o.close()


# ------------------------------------------------------------------------------------ point physics
@pytest.mark.parametrize("ityr", [1, 365, 730])
def test_point_physics_vs_golden(eng_mod, params, routine_golden, inputs, ityr):
    g = routine_golden
    e = eng_mod.Engine(inputs, params)
    in5 = np.stack([g[f"t{ityr}_in_{k}"] for k in ("Ts", "Ta", "To", "q", "cap_surf")])
    out = e.point_physics(ityr, float(g[f"t{ityr}_co2"]), in5)
    got = dict(zip(eng_mod.POINT_FIELDS, out))
    exact = ("albedo", "sw", "LWsurf", "dq_rain", "Qlat_air", "dT_ocean", "dTo", "cap_surf_new")
    for k in exact:  # no transcendental on the path: bit-exact
        assert np.array_equal(got[k], g[f"t{ityr}_out_{k}"]), k
    for k, ulps in (("em", 16), ("LWair_down", 32), ("Qlat", 64), ("dq_eva", 64)):
        ref = g[f"t{ityr}_out_{k}"]
        # Qlat = (q-qs)*...: cancellation amplifies the 1-ulp expf difference in qs
        tol = ulps * np.spacing(np.abs(ref).max())
        assert np.abs(got[k].astype(np.float64) - ref).max() <= tol, (k, np.abs(got[k] - ref).max(), tol)
    e.close()


# ------------------------------------------------------------------------------------ whole runs
TOL = {"Tsurf": 1e-4, "Tair": 1e-4, "Tocean": 1e-4, "q": 2e-8, "albedo": 1e-6}


def _check_run(mon, ref, label):
    for i, (name, tol) in enumerate(TOL.items()):
        r = rms(mon[:, i], ref[:, i])
        print(f"{label:>8s} {name:7s} rms {r:.3e} max {np.abs(mon[:, i].astype(np.float64) - ref[:, i]).max():.3e} (tol {tol:.0e})")
        assert r < tol, (label, name, r)


@pytest.mark.parametrize("strict", [True, False])
def test_run_short_vs_reference(eng_mod, params, inputs, strict):
    """1+2-yr default namelist (2xCO2) vs the reference Fortran's output/scenario."""
    g = load_golden("run_short_g96.npz")
    e = eng_mod.Engine(inputs, params, strict=strict)
    yf = e.flux_correction(1)
    mon, yr = e.run(2, 680.0)
    _check_run(mon[0].reshape(24, 5, 48, 96), g["monthly"], "strict" if strict else "fast")
    yearly = np.concatenate([yf[0], yr[0]])
    yearly_close(yearly, g["yearly"], strict)
    if not strict:
        fast_global_mean_is_the_better_one(yr[0][1][0], g["yearly"][2][0], mon[0, 1], "96x48 year 2")
    st = e.state(0)
    for i in range(4):
        assert rms(st[i], g["final_state5"][i]) < (1e-3 if i < 3 else 1e-7)
    assert np.isfinite(mon).all()
    e.close()


def test_ensemble_co2_sweep_vs_reference(eng_mod, params, inputs):
    """BASELINE config 4 in miniature: 8 CO2 levels as 8 members of ONE engine (shared flux
    correction, src/greb.f90:221), 1+3 yr, last December vs 8 separate reference runs."""
    g = load_golden("ensemble_g96.npz")
    e = eng_mod.Engine(inputs, params, n_members=8)
    yf = e.flux_correction(1)
    co2 = np.repeat(g["co2"][:, None], 3, axis=1)
    mon, yr = e.run(3, co2)
    for m in range(8):
        _check_run(mon[m, 2, 11][None], g["december"][m][None], f"co2={g['co2'][m]:.0f}")
        yearly_close(np.concatenate([yf[m], yr[m]]), g["yearly"][m])
    # members are independent: warmer with more CO2
    assert np.all(np.diff(yr[:, -1, 0]) > 0)
    e.close()


def test_resume_and_corrections_roundtrip(eng_mod, params, inputs):
    """run(2) == run(1)+run(1); corrections + state exported from one engine restart another
     bit-identically."""
    e = eng_mod.Engine(inputs, params)
    e.flux_correction(1)
    corr, st = e.get_corrections(0)
    a1, _ = e.run(1, 680.0)
    a2, _ = e.run(1, 680.0)
    e2 = eng_mod.Engine(inputs, params)
    e2.set_corrections(corr, st)
    b, _ = e2.run(2, 680.0)
    assert np.array_equal(b[0, 0], a1[0, 0]) and np.array_equal(b[0, 1], a2[0, 0])
    e.close(); e2.close()


def test_perturbed_physics_members(eng_mod, params, inputs, oracle_lib):
    """BASELINE config 5 in miniature: per-member albedo/diffusivity overrides give each member
    its own flux correction and sub-cycle tables; check two members against the oracle."""
    from greb_climate_model_amd import abi
    ov = [{}, {"kappa": 8.8e5, "a_cloud": 0.33}, {"da_ice": 0.27, "a_no_ice": 0.09}]
    e = eng_mod.Engine(inputs, params, n_members=3, overrides=ov, strict=True)
    e.flux_correction(1)
    mon, _ = e.run(1, 680.0)
    for m in (1, 2):
        p = abi.default_params(ipx=95, ipy=38, **ov[m])
        o = oracle_lib.Oracle(inputs, p)
        o.flux_correction(1)
        ref, _ = o.run(1, 680.0)
        _check_run(mon[m].reshape(12, 5, 48, 96), ref.reshape(12, 5, 48, 96), f"member{m}")
        o.close()
    assert rms(mon[1], mon[0]) > 1e-3  # the perturbation does something
    e.close()


# ------------------------------------------------------------------------------------ any-grid engine
@pytest.mark.parametrize("strict", [True, False])
def test_multilaunch_engine_g96_vs_reference(eng_mod, params, inputs, strict):
    """The any-grid (multi-launch) engine forced onto 96x48: same 1+2-yr run, same tolerances."""
    g = load_golden("run_short_g96.npz")
    e = eng_mod.Engine(inputs, params, strict=strict, multilaunch=True)
    yf = e.flux_correction(1)
    mon, yr = e.run(2, 680.0)
    _check_run(mon[0].reshape(24, 5, 48, 96), g["monthly"], "ml-" + ("strict" if strict else "fast"))
    yearly_close(np.concatenate([yf[0], yr[0]]), g["yearly"], strict)
    e.close()


def _g384_reductions_close(mon, g, label, months=None):
    """All months of a 384x192 run against the reference's per-month reductions (the full output is 35 MB and not
    committed): zonal means per latitude row and eight polar rows in full -- the rows where the reference's
    sub-cycling semantics differ most from 96x48 (225-sweep chains, the NINT(Inf) rows)."""
    n = mon.shape[0] if months is None else months
    zon = mon[:n].astype(np.float64).mean(axis=3)
    for i, (name, tol) in enumerate(TOL.items()):
        dz = np.abs(zon[:, i] - g["zonal"][:n, i]).max()
        r = rms(mon[:n, i][:, list(g["rows"])], g["polar_rows"][:n, i])
        print(f"{label:>24s} {name:7s} zonal-mean max diff {dz:.2e}  polar-rows rms {r:.2e} (tol {tol:.0e})")
        assert dz < tol and r < tol, (label, name, dz, r)  # worst measured: 0.46 x / 0.22 x tol (profiles/r02_gpu_parity_numbers.txt)


@pytest.mark.parametrize("kappa", [None, 7.2e5])
def test_strict_stencils_bit_exact_g384_vs_reference(eng_mod, oracle_lib, inputs384, kappa):
    """384x192 stencils against the REFERENCE's own diffusion / advection / circulation compiled at that grid
    (tests/golden/routine_g384.npz, oracle/Makefile ref384): STRICT bit-exact, FAST within the re-association
    tolerance; kappa = 7.2e5 puts 1 800 dependent sweeps into each polar row."""
    from greb_climate_model_amd import abi, workload
    g = load_golden("routine_g384.npz")
    p = abi.default_params()
    tag = ""
    if kappa is not None:
        p.kappa, tag = kappa, "_k72"
    Ta, q, ityr = workload.routine_inputs_g384(inputs384)
    o = oracle_lib.Oracle(inputs384, p)  # only for the weights wz_air / wz_vapor (src/greb.f90:201-202)
    wa, wv = o.field(5).copy(), o.field(6).copy()
    o.close()
    u, v = inputs384.uclim[ityr - 1], inputs384.vclim[ityr - 1]
    X = np.stack([Ta, q]); W = np.stack([wa, wv]); U = np.stack([u, u]); V = np.stack([v, v])
    d = eng_mod.diffusion(X, W, p, strict=True)
    assert np.array_equal(d[0], g["dif_Ta" + tag])
    c = eng_mod.circulation(X, W, U, V, p, strict=True)
    assert np.array_equal(c[0], g["crc_Ta" + tag])
    if kappa is None:
        a = eng_mod.advection(X, W, U, V, p, strict=True)
        assert np.array_equal(d[1], g["dif_q"]) and np.array_equal(a[0], g["adv_Ta"]) and np.array_equal(a[1], g["adv_q"])
        assert np.array_equal(c[1], g["crc_q"])
    # FAST: every row of this grid is sub-cycled, so an increment is fl(fl(T + d) - T) (:718) -- quantised to ulp(T) =
    # 3.05e-5 at 250-300 K; re-associated sweeps may land one ulp(T) away: 2 ulp(T) for one call, 8 after 24 sub-steps
    ulp_T = float(np.spacing(np.float32(np.abs(Ta).max())))
    for name, got, ref, n_ulp in (("dif", eng_mod.diffusion(X, W, p)[0], g["dif_Ta" + tag], 2),
                                  ("crc", eng_mod.circulation(X, W, U, V, p)[0], g["crc_Ta" + tag], 8)):
        err = np.abs(got.astype(np.float64) - ref)
        assert err.max() <= n_ulp * ulp_T and np.sqrt((err ** 2).mean()) < 0.5 * ulp_T, (name, err.max(), ulp_T)


def test_chain_clamp_g384(eng_mod, oracle_lib, inputs384):
    """The clamp (src/greb.f90:715,907) INSIDE the register-resident 384-point chains: a vapour field with exact zeros
    and polar spikes makes increments reach -T during the 225 / 82 / 40 ... dependent sweeps of rows 2-12 and 181-191.
    STRICT bit-exact against the oracle (itself bit-identical to the reference at this grid, routine_g384.npz); FAST --
    whose sweep is the hand-scheduled body of greb_chain6.h and takes the checked path exactly where the min test
    fires -- must take the same clamp decisions except at exact ties."""
    from greb_climate_model_amd import abi, workload
    p = abi.default_params()
    Ta, q, ityr = workload.routine_inputs_g384(inputs384)
    o = oracle_lib.Oracle(inputs384, p)
    wa, wv = o.field(5).copy(), o.field(6).copy()
    u, v = inputs384.uclim[ityr - 1], inputs384.vclim[ityr - 1]
    rng = np.random.default_rng(384)
    f = np.float32
    spiky = (q * (rng.random(q.shape) < 0.7)).astype(f)                       # 30 % exact zeros
    spiky[1, ::7] = f(0.05); spiky[2, 3::5] = f(0.08); spiky[189, 300:] = f(0.03); spiky[190, ::11] = f(0.06)
    holes = Ta.copy(); holes[1:4, 100:140] = f(0); holes[188:191, ::9] = f(0)  # zeros in the longest chains
    west = (-np.abs(u) - f(3)).astype(f)
    n_clamped = 0
    for X, W, U in ((spiky, wv, u), (spiky, wv, west), (holes, wa, u)):
        dr, ar, cr = o.diffusion(X, W), o.advection(X, W, u=U, v=v), o.circulation(X, W, u=U, v=v)
        X2, W2, U2, V2 = X[None], W[None], U[None], v[None]
        assert np.array_equal(eng_mod.diffusion(X2, W2, p, strict=True)[0], dr)
        assert np.array_equal(eng_mod.advection(X2, W2, U2, V2, p, strict=True)[0], ar)
        assert np.array_equal(eng_mod.circulation(X2, W2, U2, V2, p, strict=True)[0], cr)
        n_clamped += int(np.count_nonzero((X > 0) & (dr / np.maximum(W, f(1e-30)) <= -0.89 * X)))
        # FAST: one ulp of the state per sweep decision at most, as at 96x48 (test_stencil_edge_cases_strict)
        scale = float(np.spacing(np.abs(X).max()))
        for got, ref, n in ((eng_mod.diffusion(X2, W2, p)[0], dr, 2), (eng_mod.advection(X2, W2, U2, V2, p)[0], ar, 2),
                            (eng_mod.circulation(X2, W2, U2, V2, p)[0], cr, 24)):
            err = np.abs(got.astype(np.float64) - ref)
            assert err.max() <= 4e-6 * max(np.abs(ref).max(), 1e-30) + n * scale, (err.max(), scale)
    o.close()
    assert n_clamped > 0  # the cases do drive points to the clamp


def test_chain_loop_flavours_g384(eng_mod, oracle_lib, inputs384):
    """The three FAST loops of a 384-point diffusion chain (greb_chain6.h) against the oracle's sweeps WITH their clamp:
    rows whose values lie within a factor 64 (and whose weights make the sweep a convex combination) run without any
    clamp test -- the proof is in chain_stays_positive, this is its experiment: the harshest admissible input, 1 and 60
    alternating along the 225- / 82-sweep rows --; a wider positive range carries the minimum along; zeros take the
    stop-before loop.  Every flavour within the FAST tolerance of the sequentially clamped reference result."""
    from greb_climate_model_amd import abi, workload
    p = abi.default_params()
    Ta, q, ityr = workload.routine_inputs_g384(inputs384)
    o = oracle_lib.Oracle(inputs384, p)
    wa = o.field(5).copy()
    f = np.float32
    rng = np.random.default_rng(64)
    polar = [0, 1, 2, 3, 4, 187, 188, 189, 190, 191]
    narrow = Ta.copy(); narrow[polar] = (1 + 59 * (np.arange(384) % 2)).astype(f)                 # 1, 60, 1, 60, ...
    narrow[2] = (1 + 49 * rng.random(384)).astype(f)
    wide = Ta.copy(); wide[polar] = (10.0 ** rng.uniform(-3, 1, (len(polar), 384))).astype(f)     # four decades, all > 0
    zeros = narrow.copy(); zeros[1, 5] = f(0); zeros[190, 77] = f(0)
    for name, X in (("narrow", narrow), ("wide", wide), ("zeros", zeros)):
        ref = o.diffusion(X, wa)
        got = eng_mod.diffusion(X[None], wa[None], p)[0]
        err = np.abs(got.astype(np.float64) - ref)[polar]  # at the scale of the polar rows themselves
        tol = 4 * float(np.spacing(np.abs(X[polar]).max()))  # 225 sweeps later: 4 ulp of the rows' largest value (measured 1-2)
        print(f"chain flavour {name}: polar rows max |FAST - oracle| = {err.max():.3e} (tolerance {tol:.3e})")
        assert err.max() <= tol, (name, err.max(), tol)
        if name == "narrow":  # (a total decay of 89 % would be the trace of a clamp here: the rows relax to their mean, ~30)
            assert not np.any((X[polar] > 0) & (ref[polar] / np.maximum(wa[polar], f(1e-30)) <= -0.89 * X[polar]))
        assert np.array_equal(eng_mod.diffusion(X[None], wa[None], p, strict=True)[0], ref)
    o.close()


def test_row_strip_substep_equals_band_kernel_strict(eng_mod, inputs384):
    """The row-strip circulation sub-step (greb_step_rows.hip; GREB_F_ROW_STRIPS) in STRICT arithmetic against the band
    kernel in STRICT arithmetic: the same expression trees through completely different data movement (wavefront-private
    LDS rings fed by LDS-DMA, 6 longitudes per lane, wave rotates for the zonal halo) -- one flux-correction month and a
    scenario month of three members with different physics (own kappa: own sub-cycle tables) must agree BIT FOR BIT."""
    from greb_climate_model_amd import abi
    g = load_golden("g384_physpar.npz")
    ov = [dict(zip(("da_ice", "a_no_ice", "a_cloud", "kappa"), map(float, g["overrides"][m]))) for m in (0, 1, 4)]
    out = []
    # band kernel | strips, one launch per sub-step | strips, one launch per circulation call (greb_circ_rows.hip: the 24
    # sub-steps in the kernel, rows handed from strip to strip through memory flags; chain rows resident in registers)
    for strips, persistent in ((False, False), (True, False), (True, True)):
        e = eng_mod.Engine(inputs384, abi.default_params(ipx=380, ipy=152), n_members=3, overrides=ov, strict=True,
                           row_strips=strips, persistent=persistent)
        e.flux_correction(1)
        mon, yr = e.run(1, 680.0)
        st = [e.state(m) for m in range(3)]
        e.close()
        out.append((mon, yr, st))
    for other in out[1:]:
        assert np.array_equal(out[0][0], other[0]) and np.array_equal(out[0][1], other[1])
        for a, b in zip(out[0][2], other[2]):
            assert np.array_equal(a, b)
    assert np.isfinite(out[1][0]).all()


@pytest.mark.parametrize("members", [1, 5, 40])
def test_one_launch_circulation_equals_one_launch_per_substep_fast(eng_mod, inputs384, members):
    """FAST arithmetic, the product path: the circulation call as ONE launch (src/greb.f90:546-550 as the reference runs
    it: 24 sub-steps inside one call) against one launch per sub-step.  Row by row the two run the same instruction
    sequences on the same operands -- what differs is who hands which row to whom and when (kernel boundaries there;
    agent-scope stores, flags and L1-bypassing LDS-DMA here; chain rows that never leave their registers) -- so a year of
    both, flux corrections included, must agree BIT FOR BIT: any row read before its owner had written it, or from a stale
    cache line, would show.  1 member: every task has a SIMD of its own; 5: own diffusivities incl. the 1 800-sweep
    member, chain tasks next to each other; 40: tasks share SIMDs."""
    from greb_climate_model_amd import abi
    g = load_golden("g384_physpar.npz")
    ov = None
    if members > 1:
        ov = [dict(zip(("da_ice", "a_no_ice", "a_cloud", "kappa"), map(float, g["overrides"][m % 5]))) for m in range(members)]
    if members == 40:
        for o in ov:
            o["kappa"] = max(o["kappa"], 7.4e5)  # (the 1 800-sweep rows are in the 5-member case; here: a short run)
    out = []
    # one launch per sub-step | one launch per call | the default: the engine times both in its first eight model steps
    # (switching between them mid-year) and keeps the faster
    for persistent in (False, True, None):
        e = eng_mod.Engine(inputs384, abi.default_params(ipx=380, ipy=152), n_members=members, overrides=ov, persistent=persistent)
        yf = e.flux_correction(1)
        mon, yr = e.run(1, 680.0)
        st = [e.state(m) for m in range(members)]
        d = e.describe()
        e.close()
        out.append((mon, yr, yf, st))
        forms = {c["members_run"]: c["form"] for c in d.get("circulation", [])}
        print(f"members {members} persistent={persistent}: {d}")
        if persistent is False:
            assert not forms
        else:
            assert forms and all(f.startswith("one launch per") for f in forms.values()), d  # decided, every members_run
            if persistent:
                assert set(forms.values()) == {"one launch per call"}, d
    assert np.isfinite(out[1][0]).all()
    for other in out[1:]:
        assert np.array_equal(out[0][0], other[0]) and np.array_equal(out[0][1], other[1]) and np.array_equal(out[0][2], other[2])
        for a, b in zip(out[0][3], other[3]):
            assert np.array_equal(a, b)


def test_row_strip_substep_with_experiment_switches(eng_mod, inputs384):
    """The process switches of the upstream variant (GREB_X_*) on the 384-wide row-strip sub-step: vapour diffused but
    not advected (the strips give the vapour fields zero wind), no vapour transport, no circulation at all.  FAST on the
    strips against STRICT on the band kernel with the same switches, one flux-correction year and one scenario year; and
    each switch must change the climate."""
    from greb_climate_model_amd import abi
    p = abi.default_params(ipx=380, ipy=152)
    base = None
    for x in (0, abi.X_VAPOR_DIFFUSION_ONLY, abi.X_NO_VAPOR_TRANSPORT, abi.X_NO_CIRCULATION):
        out = []
        for strict in (True, False):
            e = eng_mod.Engine(inputs384, p, strict=strict)
            e.set_experiment(x)
            e.flux_correction(1)
            mon, _ = e.run(1, 680.0)
            e.close()
            out.append(mon[0, 0])
        _check_run(out[1], out[0], f"g384 switches {x}")
        if x == 0:
            base = out[1]
        else:
            assert rms(out[1][:, 3], base[:, 3]) > 1e-6, x  # humidity responds to every one of these switches


@pytest.mark.parametrize("mode", ["strict", "fast", "fast40", "strict_strips"])
def test_engine_g384_vs_reference(eng_mod, inputs384, mode):
    """BASELINE config 3 in miniature against the REFERENCE compiled at 384x192 (g384_short.npz): 1+2 yr, 2xCO2.
    strict: 2 members on the band kernel (greb_kernels.hip: sweep_kernel<fused>); fast / fast40: 2 / 40 members on the
    row-strip sub-step (greb_step_rows.hip); strict_strips: the row strips in STRICT arithmetic (GREB_F_ROW_STRIPS).
    Months 1, 12, 24 in full, every month by zonal means and polar rows."""
    from greb_climate_model_amd import abi
    g = load_golden("g384_short.npz")
    p = abi.default_params(ipx=380, ipy=152)
    nm = 40 if mode == "fast40" else 2
    e = eng_mod.Engine(inputs384, p, n_members=nm, strict=mode.startswith("strict"), row_strips=mode == "strict_strips")
    yf = e.flux_correction(1)
    co2 = np.full((nm, 2), 340.0, np.float32); co2[-1] = 680.0
    mon, yr = e.run(2, co2)
    e.close()
    last = mon[-1].reshape(24, 5, 192, 384)
    _check_run(last[[0, 11, 23]], g["monthly_sel"], f"g384 {mode}")
    _g384_reductions_close(last, g, f"g384 {mode}")
    yearly_close(np.concatenate([yf[-1], yr[-1]]), g["yearly"], mode.startswith("strict"), 384 * 192)
    if not mode.startswith("strict"):
        fast_global_mean_is_the_better_one(yr[-1][1][0], g["yearly"][2][0], mon[-1, 1], f"384x192 {mode} year 2")
    assert rms(mon[0, 1, 11, 0], mon[-1, 1, 11, 0]) > 1e-2  # the members differ (CO2)
    if nm > 2:
        assert np.array_equal(mon[0], mon[1])  # replicas agree bit for bit


@pytest.mark.parametrize("mode", ["strict", "fast", "fast40", "fast60"])
def test_config5_perturbed_members_g384_vs_reference(eng_mod, inputs384, mode):
    """BASELINE config 5 in miniature: perturbed-physics members at 384x192 -- per-member RowTables (own kappa),
    own flux corrections (shared_corr = false), the multi-launch engine with the row-strip sub-step (fast*) -- against
    five separate runs of the reference with the same &PHYSICS_PAR (g384_physpar.npz), 1+1 yr.  Member 4 has
    kappa = 7.2e5: 1 800 dependent sweeps in each polar row (inherited semantics, src/greb.f90:652-654)."""
    from greb_climate_model_amd import abi
    g = load_golden("g384_physpar.npz")
    ov = g["overrides"]
    # 40 / 60 members: the launch order of the row strips is cut coarser as the member count grows (step_rows_tasks)
    reps = {"fast40": 8, "fast60": 12}.get(mode, 1)
    overrides = [dict(zip(("da_ice", "a_no_ice", "a_cloud", "kappa"), map(float, ov[m % 5]))) for m in range(5 * reps)]
    e = eng_mod.Engine(inputs384, abi.default_params(ipx=380, ipy=152), n_members=5 * reps, overrides=overrides,
                       strict=mode == "strict")
    yf = e.flux_correction(1)
    mon, yr = e.run(1, 680.0)
    e.close()
    for m in range(5):
        mm = mon[5 * (reps - 1) + m, 0]
        _check_run(mm[11][None], g["december"][m][None], f"cfg5 {mode} m{m}")
        _g384_reductions_close(mm, {k: g[k][m] if k != "rows" else g[k] for k in ("zonal", "polar_rows", "rows")}, f"cfg5 {mode} m{m}")
        yearly_close(np.concatenate([yf[5 * (reps - 1) + m], yr[5 * (reps - 1) + m]]), g["yearly"][m], mode == "strict", 384 * 192)
    assert rms(mon[0, 0, 11, 0], mon[1, 0, 11, 0]) > 1e-3  # different physics, different climate
    if reps > 1:
        assert np.array_equal(mon[0], mon[5])  # replicas agree bit for bit


def test_config5_two_engines_beside_each_other_vs_reference(eng_mod, inputs384):
    """The execution path bench.py's `g384.config5` object and tools/run_config.py 5 take: the ensemble split by what
    bounds a member (ensemble.latency_groups: the kappa = 7.2e5 member with its 1 800-sweep polar rows apart from the
    other four) into TWO engines driven by two host threads on one device (ensemble.run_beside), the months put back
    in member order -- against the five reference runs of g384_physpar.npz, member by member.  Members are independent
    runs in the reference (src/greb.f90:153,1064-1068), so who shares an engine, a launch or the device with whom must
    not show in anybody's result: the same five members through ONE engine must agree bit for bit."""
    from greb_climate_model_amd import abi, ensemble
    g = load_golden("g384_physpar.npz")
    ov = g["overrides"]
    keys = ("da_ice", "a_no_ice", "a_cloud", "kappa")
    overrides = [dict(zip(keys, map(float, ov[m]))) for m in range(5)]
    p = abi.default_params(ipx=380, ipy=152)
    groups = ensemble.latency_groups([o["kappa"] for o in overrides], 384, 192)
    assert [list(map(int, x)) for x in groups] == [[0, 1, 2, 3], [4]], groups  # :652-654: dtdff2 0 -> 1 below 7.27e5
    engines = [eng_mod.Engine(inputs384, p, n_members=len(x), overrides=[overrides[i] for i in x]) for x in groups]
    yfs = ensemble.run_beside([lambda e=e: e.flux_correction(1) for e in engines])
    outs = ensemble.run_beside([lambda e=e: e.run(1, 680.0) for e in engines])
    for e in engines:
        e.close()
    mon = np.empty((5,) + outs[0][0].shape[1:], np.float32)
    yf = np.empty((5,) + yfs[0].shape[1:], np.float32); yr = np.empty((5,) + outs[0][1].shape[1:], np.float32)
    for x, a, (m_, y_) in zip(groups, yfs, outs):
        mon[x] = m_; yf[x] = a; yr[x] = y_
    for m in range(5):
        mm = mon[m, 0]
        _check_run(mm[11][None], g["december"][m][None], f"cfg5 2eng m{m}")
        _g384_reductions_close(mm, {k: g[k][m] if k != "rows" else g[k] for k in ("zonal", "polar_rows", "rows")}, f"cfg5 2eng m{m}")
        yearly_close(np.concatenate([yf[m], yr[m]]), g["yearly"][m], False, 384 * 192)
    one = eng_mod.Engine(inputs384, p, n_members=5, overrides=overrides)
    yf1 = one.flux_correction(1)
    mon1, yr1 = one.run(1, 680.0)
    one.close()
    assert np.array_equal(mon1, mon) and np.array_equal(yf1, yf) and np.array_equal(yr1, yr)


def test_table_caches_are_bounded_and_stay_correct(eng_mod, params, inputs, inputs384, oracle_lib):
    """A long-lived host that sweeps kappa: the device copies of row tables and launch orders behind the batched entry
    points are LRU caches of 32 entries (an entry is retired only once the device is idle).  Forty diffusivities through
    both grids, then the first again: bit-identical to its first answer and to the oracle."""
    from greb_climate_model_amd import abi
    f = np.float32
    X96, W96 = inputs.tclim[0][None], np.full((1, 48, 96), f(0.7))
    X384, W384 = inputs384.tclim[0][None], np.full((1, 192, 384), f(0.7))
    first = None
    for i in list(range(40)) + [0]:
        p = abi.default_params(ipx=95, ipy=38)
        p.kappa = f(8e5) * (f(1) + f(0.004) * f(i))
        a = eng_mod.diffusion(X96, W96, p, strict=True)
        b = eng_mod.diffusion(X384, W384, p, strict=True)
        if first is None:
            first = (a.copy(), b.copy())
            o = oracle_lib.Oracle(inputs, p)
            assert np.array_equal(a[0], o.diffusion(X96[0], W96[0]))
            o.close()
    assert np.array_equal(a, first[0]) and np.array_equal(b, first[1])


# ------------------------------------------------------------------------------------ error behaviour
def test_bad_arguments_are_errors_with_messages(eng_mod, params, inputs):
    """Nothing throws or exits across the ABI: bad shapes / indices / call arguments come back as GREB_E_INVALID
    with a message (include/greb_engine.h error convention); zero flux-correction years are legal (A.9-10)."""
    import ctypes as C
    from greb_climate_model_amd import abi, workload
    L = eng_mod.lib()
    fields, keep = abi.make_fields(inputs)
    h = C.c_void_p()
    for nx, ny, nm in ((94, 48, 1), (96, 3, 1), (96, 48, 0), (8, 48, 1)):
        assert L.greb_engine_create(C.byref(params), nx, ny, C.byref(fields), nm, None, 0, 0, C.byref(h)) == -1
        assert b"bad argument" in L.greb_engine_last_error(None)
    p2 = abi.default_params(ipx=97, ipy=1)
    assert L.greb_engine_create(C.byref(p2), 96, 48, C.byref(fields), 1, None, 0, 0, C.byref(h)) == -1
    assert L.greb_engine_create(C.byref(params), 96, 48, C.byref(fields), 1, None, 99, 0, C.byref(h)) != 0  # no such device
    e = eng_mod.Engine(inputs, params, n_members=2)
    with pytest.raises(eng_mod.GrebError):
        e.state(2)
    with pytest.raises(eng_mod.GrebError):
        e.set_experiment(1 << 9)
    with pytest.raises(eng_mod.GrebError):
        e.point_physics(731, 680.0, np.zeros((5, 48, 96), np.float32))
    assert L.greb_engine_run(e.h, 0, None, None, None, 0) == -1
    assert e.flux_correction(0).shape == (2, 0, 2)  # time_flux = 0: corrections stay zero, the run still works
    mon, _ = e.run(1, 680.0)
    assert np.isfinite(mon).all()
    e.close()
    with pytest.raises(eng_mod.GrebError):  # batched routines: same validation
        eng_mod.diffusion(np.zeros((48, 94), np.float32), np.ones((48, 94), np.float32), params)


def test_co2_series_vs_reference(eng_mod, params, inputs):
    """A CO2 concentration that changes from year to year (src/greb.f90:918-926 picks co2_ppm by model year): 1+3 yr
    with 400, 520, 520 ppm against the reference Fortran run with the namelist series `400, 520` (which it pads)."""
    g = load_golden("co2series_g96.npz")
    e = eng_mod.Engine(inputs, params)
    yf = e.flux_correction(1)
    mon, yr = e.run(3, g["co2"])
    e.close()
    mon = mon[0].reshape(36, 5, 48, 96)
    for j, month in enumerate((11, 23, 35)):
        for i, tol in enumerate((1e-4, 1e-4, 1e-4, 2e-8, 1e-6)):
            assert rms(mon[month, i], g["decembers"][j, i]) < tol, (month, i)
    assert np.abs(mon.astype(np.float64).mean((2, 3)) - g["stats"][:, :, 0]).max() < 1e-4
    yearly_close(np.concatenate([yf[0], yr[0]]), g["yearly"])
    # two calls of one and two years continue the same series (the model year advances across calls)
    e = eng_mod.Engine(inputs, params)
    e.flux_correction(1)
    a, _ = e.run(1, g["co2"][:1])
    b, _ = e.run(2, g["co2"][1:])
    e.close()
    assert np.array_equal(np.concatenate([a[0], b[0]]).reshape(36, 5, 48, 96), mon)


@pytest.mark.parametrize("strict", [False, True])
def test_nondefault_physics_par_vs_reference(eng_mod, inputs, strict):
    """Engine-wide parameter overrides (the namelist group physics_par): kappa = 6e5 (six instead of eight sweeps
    in the polar rows), a_cloud, da_ice, ct_sens changed; 1+1 yr against the reference Fortran's own output."""
    from greb_climate_model_amd import abi
    g = load_golden("physpar_g96.npz")
    phys = {str(k): float(v) for k, v in zip(g["names"], g["values"])}
    e = eng_mod.Engine(inputs, abi.default_params(ipx=95, ipy=38, **phys), strict=strict)
    yf = e.flux_correction(1)
    mon, yr = e.run(1, 680.0)
    e.close()
    _check_run(mon[0].reshape(12, 5, 48, 96), g["monthly"], f"physics_par strict={strict}")
    yearly_close(np.concatenate([yf[0], yr[0]]), g["yearly"], strict)


@pytest.mark.parametrize("strict", [False, True])
def test_run_without_flux_correction_vs_reference(eng_mod, params, inputs, strict):
    """time_flux = 0: the engine's corrections stay zero and the scenario starts from the initial state, like the
    reference run with that namelist (the model drifts several K in a year -- a sensitive comparison)."""
    g = load_golden("noflux_g96.npz")
    e = eng_mod.Engine(inputs, params, strict=strict)
    assert e.flux_correction(0).shape[1] == 0
    mon, yr = e.run(1, 680.0)
    e.close()
    mon = mon[0].reshape(12, 5, 48, 96)
    for j, month in enumerate((0, 5, 11)):
        for i, tol in enumerate((1e-4, 1e-4, 1e-4, 2e-8, 1e-6)):
            assert rms(mon[month, i], g["months"][j, i]) < tol, (month, i)
    assert np.abs(mon.astype(np.float64).mean((2, 3)) - g["stats"][:, :, 0]).max() < 1e-4
    yearly_close(yr[0], g["yearly"], strict)


def test_engine_g192_vs_reference(eng_mod, oracle_lib):
    """A grid that is neither of the two BASELINE ones (runtime grid sizes): 192x96, bilinear-upsampled
    inputs, every row sub-cycled, 10 rows iterating (up to 129 sweeps).  Against the REFERENCE compiled at that grid
    (tests/golden/routine_g192.npz, g192_short.npz; oracle/Makefile ref192): batched STRICT stencils bit-exact against
    the reference's own subroutines, FAST within the re-association tolerance, and a 1+1-yr run in both arithmetic
    modes against the reference program's output (any-grid multi-launch engine)."""
    import json
    from greb_climate_model_amd import abi, workload
    g, gs = load_golden("routine_g192.npz"), load_golden("g192_short.npz")
    ityr = json.load(open(os.path.join(GOLDEN, "MANIFEST.json")))["items"]["routine_g192"]["ityr"]
    inp = workload.make_inputs(192, 96)
    p = abi.default_params(ipx=190, ipy=75)
    o = oracle_lib.Oracle(inp, p)  # only for the weights wz_air / wz_vapor (src/greb.f90:201-202)
    wa, wv = o.field(5).copy(), o.field(6).copy()
    o.close()
    T, q = inp.tclim[ityr - 1], inp.qclim[ityr - 1]
    u, v = inp.uclim[ityr - 1], inp.vclim[ityr - 1]
    X, W, U, V = np.stack([T, q]), np.stack([wa, wv]), np.stack([u, u]), np.stack([v, v])
    for name, fn in (("dif", lambda s: eng_mod.diffusion(X, W, p, strict=s)), ("adv", lambda s: eng_mod.advection(X, W, U, V, p, strict=s)),
                     ("crc", lambda s: eng_mod.circulation(X, W, U, V, p, strict=s))):
        ref = np.stack([g[name + "_Ta"], g[name + "_q"]])
        assert np.array_equal(fn(True), ref), name
        fast = fn(False)
        for i in range(2):  # increments are fl(fl(T + d) - T): quantised to ulp(T); 2 ulp per call, 8 after 24 sub-steps
            ulp = float(np.spacing(np.float32(np.abs(X[i]).max())))
            assert np.abs(fast[i].astype(np.float64) - ref[i]).max() <= (8 if name == "crc" else 2) * ulp + 4e-6 * np.abs(ref[i]).max(), (name, i)
    ref = gs["monthly"]
    # STRICT on the latitude bands; FAST on the row strips (the default there since round 4: the 192-wide row laid twice
    # around the wavefront, greb_rows.h -- one launch per circulation call and one per sub-step); STRICT on the strips too
    runs = {}
    for label, kw in (("strict bands", dict(strict=True)), ("fast strips, one launch per call", dict(persistent=True)),
                      ("fast strips, one launch per sub-step", dict(persistent=False)),
                      ("strict strips, one launch per call", dict(strict=True, row_strips=True, persistent=True)),
                      ("strict strips, one launch per sub-step", dict(strict=True, row_strips=True, persistent=False))):
        e = eng_mod.Engine(inp, p, **kw)
        d = e.describe()
        assert d["engine"] == "row strips", d  # this grid takes the strips (STRICT without row_strips keeps its bands at run time)
        yf = e.flux_correction(1)
        mon, yr = e.run(1, 680.0)
        d = e.describe()
        e.close()
        forms = [c["form"] for c in d.get("circulation", [])]
        if "strips" in label:
            assert forms == ["one launch per call" if "per call" in label else "one launch per sub-step"] or (not forms and "sub-step" in label), (label, d)
        strict = label.startswith("strict")
        _check_run(mon[0, 0], ref, f"g192 {label}")
        yearly_close(np.concatenate([yf[0], yr[0]]), gs["yearly"], strict, 192 * 96)
        runs[label] = (mon, yr, yf)
    for a, b in (("strict bands", "strict strips, one launch per call"), ("strict bands", "strict strips, one launch per sub-step"),
                 ("fast strips, one launch per call", "fast strips, one launch per sub-step")):
        for x, y in zip(runs[a], runs[b]):
            assert np.array_equal(x, y), (a, b)  # the same arithmetic through different data movement: bit for bit


def test_engine_g192_members_are_independent(eng_mod):
    """192x96 on the native strips with several members in one engine (tasks share SIMDs from 10 members on): a member's
    result must not depend on who else is in the engine -- member k of a 24-member CO2 sweep equals the same member run
    alone, bit for bit (members are separate processes in the reference, src/greb.f90:153,1064-1068) --, replicas agree,
    and more CO2 is warmer."""
    from greb_climate_model_amd import abi, ensemble, workload
    inp = workload.make_inputs(192, 96)
    p = abi.default_params(ipx=190, ipy=75)
    levels = ensemble.co2_sweep(24).astype(np.float32)
    levels[23] = levels[5]  # a replica
    e = eng_mod.Engine(inp, p, n_members=24)
    e.flux_correction(1)
    mon, yr = e.run(1, levels[:, None])
    e.close()
    assert np.isfinite(mon).all() and np.array_equal(mon[23], mon[5])
    t = yr[:23, -1, 0]
    assert (np.diff(t) > 0).all(), t
    one = eng_mod.Engine(inp, p, n_members=1)
    one.flux_correction(1)
    m1, y1 = one.run(1, levels[7:8, None])
    one.close()
    assert np.array_equal(m1[0], mon[7]) and np.array_equal(y1[0], yr[7])


@pytest.mark.parametrize("nx,ny", [(384, 96), (192, 192), (192, 48), (384, 48)])
def test_row_strips_on_other_384_and_192_wide_grids(eng_mod, nx, ny):
    """The row-strip kernels take any grid 384 or 192 longitudes wide (src/greb.f90:36 is all that fixes the grid in the
    reference): other latitude counts put the iterating rows, the chain tasks and the strip cuts elsewhere.  No reference
    build exists at these grids, so this is a consistency pin, not a parity pin: STRICT on the strips -- one launch per
    circulation call -- against STRICT on the latitude bands (which the three reference-pinned grids hold to the reference),
    bit for bit over a flux-correction year and a scenario year with two members; FAST stays within the whole-run
    tolerances of the STRICT result."""
    from greb_climate_model_amd import abi, workload
    inp = workload.make_inputs(nx, ny)
    p = abi.default_params(ipx=nx - 3, ipy=max(2, (3 * ny) // 4))
    co2 = np.array([[340.0], [680.0]], np.float32)
    out = {}
    for label, kw in (("bands", dict(strict=True)), ("strips", dict(strict=True, row_strips=True, persistent=True)), ("fast", dict())):
        e = eng_mod.Engine(inp, p, n_members=2, **kw)
        assert e.describe()["engine"] == "row strips"
        yf = e.flux_correction(1)
        mon, yr = e.run(1, co2)
        e.close()
        out[label] = (mon, yr, yf)
    assert np.isfinite(out["fast"][0]).all()
    for a, b in zip(out["bands"], out["strips"]):
        assert np.array_equal(a, b)
    _check_run(out["fast"][0][1, 0], out["bands"][0][1, 0].astype(np.float64), f"{nx}x{ny} fast vs strict")
    assert rms(out["fast"][0][0, 0, 11, 0], out["fast"][0][1, 0, 11, 0]) > 1e-2  # the members differ (CO2)

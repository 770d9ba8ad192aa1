"""GPU: on-device ensemble statistics (greb_ensemble_moments_dev / greb_ensemble_quantiles_dev)
against numpy in float64.  The reference has no counterpart (its ensembles are separate processes analysed in R):
the checker is the textbook definition, so this row is "parity unpinned" against the reference by construction."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("m,n", [(1, 64), (7, 4608), (64, 4610), (512, 5 * 4608), (300, 1001)])
def test_moments_vs_numpy(m, n):
    import torch
    from greb_climate_model_amd import ensemble
    rng = np.random.default_rng(m * 1000 + n)
    x = (280.0 + 15.0 * rng.standard_normal((m, n))).astype(np.float32)
    x[0, :3] = [1e-3, -5.0, 400.0]
    xd = torch.from_numpy(x).cuda()
    s = ensemble.ensemble_summary(xd)  # single process: no all-reduce
    x64 = x.astype(np.float64)
    assert s["n"] == m
    assert np.allclose(s["mean"].cpu().numpy(), x64.mean(0), rtol=1e-6, atol=0)
    assert np.allclose(s["var"].cpu().numpy(), x64.var(0), rtol=2e-4, atol=1e-4)  # E[x^2]-E[x]^2 in fp64, cast to fp32
    assert np.array_equal(s["min"].cpu().numpy(), x.min(0)) and np.array_equal(s["max"].cpu().numpy(), x.max(0))


@pytest.mark.parametrize("m,n", [(1, 10), (2, 100), (8, 4608), (100, 777), (512, 2 * 4608), (1000, 300)])
def test_quantiles_vs_numpy(m, n):
    import torch
    from greb_climate_model_amd import ensemble
    rng = np.random.default_rng(m + n)
    x = (280.0 + 15.0 * rng.standard_normal((m, n))).astype(np.float32)
    x[:, 0] = 5.0  # ties
    probs = [0.0, 0.05, 0.25, 0.5, 0.75, 0.95, 1.0]
    got = ensemble.ensemble_quantiles(torch.from_numpy(x).cuda(), probs).cpu().numpy()
    want = np.quantile(x.astype(np.float64), probs, axis=0)
    assert got.shape == want.shape
    assert np.abs(got - want).max() < 1e-4  # fp32 interpolation of values ~300
    assert np.array_equal(got[0], x.min(0)) and np.array_equal(got[-1], x.max(0))


def test_statistics_of_a_real_ensemble(inputs, params):
    """The engine's own output: 64 members of a CO2 sweep, last December; mean / spread / median are finite,
    ordered, and the median member is between the coldest and warmest."""
    import torch
    from greb_climate_model_amd import engine, ensemble
    M = 64
    e = engine.Engine(inputs, params, n_members=M)
    e.flux_correction(1)
    dev = torch.empty((M, 1, 12, 5, 48 * 96), dtype=torch.float32, device="cuda")
    e.run(1, ensemble.co2_sweep(M)[:, None], monthly_dev_ptr=dev.data_ptr())
    e.close()
    dec = dev[:, 0, 11].contiguous()  # [M, 5, np]
    s = ensemble.ensemble_summary(dec)
    q = ensemble.ensemble_quantiles(dec, [0.05, 0.5, 0.95])
    assert all(bool(torch.isfinite(v).all()) for v in (s["mean"], s["var"], q))
    assert bool((s["min"] <= q[0]).all() and (q[0] <= q[1]).all() and (q[1] <= q[2]).all() and (q[2] <= s["max"]).all())
    assert float((s["max"][0] - s["min"][0]).mean()) > 0.5  # Tsurf spread of a 280..1120 ppm sweep after one year
    ref = dec.double()
    assert torch.allclose(s["mean"].double(), ref.mean(0), rtol=1e-6)

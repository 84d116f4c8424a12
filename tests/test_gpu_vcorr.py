"""GPU parity tests of the vector 2-point correlation (tgp_vcorr, treegp_amd.utils) against the NumPy
restatement of treegp/utils.py:5-105 in oracle/gp_oracle.py."""
import warnings

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _field(n, seed, scale=1.0):
    rng = np.random.default_rng(seed)
    x, y = rng.uniform(0, scale, n), rng.uniform(0, scale, n)
    dx = np.sin(3 * x / scale) + 0.1 * rng.standard_normal(n)
    dy = np.cos(2 * y / scale) * np.sin(x / scale) + 0.1 * rng.standard_normal(n)
    return x, y, dx, dy


@pytest.mark.parametrize("n,kw", [(3000, {}), (1500, dict(rmin=0.01, rmax=0.8, dlogr=0.2)), (257, dict(rmin=1e-3, rmax=3.0, dlogr=0.5)),
                                   (2, {}), (1, {})])
def test_vcorr_matches_oracle(n, kw):
    from treegp_amd import utils
    from oracle import gp_oracle as O
    x, y, dx, dy = _field(n, n)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        got = utils.vcorr(x, y, dx, dy, **kw)
    ref = O.vcorr(x, y, dx, dy, **kw)
    from treegp_amd import ops
    edges, bins = utils._log_edges(kw.get("rmin", 5.0 / 3600.0), kw.get("rmax", 1.5), kw.get("dlogr", 0.05))
    counts = ops.vcorr_sums(x, y, dx, dy, edges)[0]
    np.testing.assert_array_equal(counts, ref[5])                             # bin numbers: exact
    for g, r in zip(got, ref[:5]):
        assert g.shape == r.shape
        np.testing.assert_array_equal(np.isnan(g), np.isnan(r))
        ok = ~np.isnan(r)
        np.testing.assert_allclose(g[ok], r[ok], rtol=0, atol=1e-12 * max(1.0, np.abs(r[ok]).max(initial=0.0)))


def test_vcorr_duplicates_and_edges():
    """coincident points (log 0 = -inf is outside the range) and separations exactly on bin edges"""
    from treegp_amd import ops, utils
    from oracle import gp_oracle as O
    edges, bins = utils._log_edges(0.01, 1.0, 0.25)
    r = np.exp(edges)                                    # separations on every edge, incl. the last (inclusive)
    x = np.concatenate([[0.0, 0.0], r, 10.0 + r])
    y = np.zeros_like(x); y[len(r) + 2:] = 5.0
    rng = np.random.default_rng(0)
    dx, dy = rng.standard_normal(len(x)), rng.standard_normal(len(x))
    acc = ops.vcorr_sums(x, y, dx, dy, edges)
    ref = O.vcorr(x, y, dx, dy, rmin=0.01, rmax=1.0, dlogr=0.25)
    np.testing.assert_array_equal(acc[0], ref[5])
    ok = ref[5] > 0
    np.testing.assert_allclose(acc[2][ok] / acc[0][ok], ref[1][ok], rtol=0, atol=1e-12)


def test_comp_eb_runs_and_is_consistent():
    """what the reference's own test exercises (tests/test_hyp_search.py:133-139: both entry points run), plus
    E + B = xi+ and agreement of the two binning conventions where bins are populated"""
    import treegp_amd
    from treegp_amd import utils
    x, y, dx, dy = _field(4000, 11, scale=2.0)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        e, b, logr = treegp_amd.comp_eb(x, y, dx, dy)
        e2, b2, logr2 = treegp_amd.comp_eb_treecorr(x, y, dx, dy)
        lr, xip, xim, xix, xiz2 = utils.vcorr(x, y, dx, dy)
    assert e.shape == b.shape == logr.shape == e2.shape == (140,)
    ok = np.isfinite(e)
    np.testing.assert_allclose((e + b)[ok], xip[ok], rtol=0, atol=1e-12)
    np.testing.assert_allclose(np.diff(logr2), 0.05, rtol=1e-12)
    np.testing.assert_allclose(logr[ok], logr2[ok], atol=0.025)            # mean log r lies inside its bin
    # subsampling branch (utils.py:28-35): about maxpts points survive, results stay finite where populated
    np.random.seed(3)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        out = utils.vcorr(x, y, dx, dy, maxpts=1000)
    assert np.isfinite(out[1][np.isfinite(out[0])]).all()

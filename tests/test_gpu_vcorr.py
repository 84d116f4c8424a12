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


def _cmp_golden(got, g, tag):
    for v, nm in zip(got, ("logr", "xiplus", "ximinus", "xicross", "xiz2")):
        ref = g[tag + "_" + nm]
        assert v.shape == ref.shape
        np.testing.assert_array_equal(np.isnan(v), np.isnan(ref), err_msg=nm)
        ok = ~np.isnan(ref)
        np.testing.assert_allclose(v[ok], ref[ok], rtol=0, atol=1e-12 * max(1.0, np.abs(ref[ok]).max(initial=0.0)), err_msg=nm)


def test_vcorr_against_reference_golden(golden):
    """tgp_vcorr through treegp_amd.utils against the REFERENCE's own vcorr / comp_eb outputs (g9_vcorr.npz,
    produced by importing treegp/utils.py unmodified): vector fields, the default 140-bin grid with empty bins,
    and the subsampling branch on the legacy global random stream."""
    from treegp_amd import utils
    g = golden("g9_vcorr.npz")
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        kw = dict(rmin=float(g["a_rmin"]), rmax=float(g["a_rmax"]), dlogr=float(g["a_dlogr"]))
        _cmp_golden(utils.vcorr(g["a_x"], g["a_y"], g["a_dx"], g["a_dy"], **kw), g, "a")
        xie, xib, logr = utils.comp_eb(g["a_x"], g["a_y"], g["a_dx"], g["a_dy"], **kw)
        np.testing.assert_allclose(xie, g["a_xie"], rtol=0, atol=1e-12)
        np.testing.assert_allclose(xib, g["a_xib"], rtol=0, atol=1e-12)
        np.testing.assert_allclose(logr, g["a_logr"], rtol=0, atol=1e-12)
        _cmp_golden(utils.vcorr(g["b_x"], g["b_y"], g["b_dx"], g["b_dy"]), g, "b")
        np.random.seed(int(g["d_seed"]))
        _cmp_golden(utils.vcorr(g["d_x"], g["d_y"], g["d_dx"], g["d_dy"], rmin=float(g["d_rmin"]), rmax=float(g["d_rmax"]),
                                dlogr=float(g["d_dlogr"]), maxpts=int(g["d_maxpts"])), g, "d")


def test_kk_log_against_reference_golden(golden):
    """tgp_kk_log against the reference's exact pair binner run on a scalar field (g9 case c: dy = 0, bins equal to
    the KK log grid): xi = xi+, meanlogr = logr; the same data through tgp_vcorr as well."""
    from treegp_amd import ops, utils
    g = golden("g9_vcorr.npz")
    x, y, k = g["c_x"], g["c_y"], g["c_k"]
    mn, mx, nb = float(g["c_min_sep"]), float(g["c_max_sep"]), int(g["c_nbins"])
    xi, wt, meanr, meanlogr, npairs = ops.kk_log(x, y, k, None, mn, mx, nb)
    np.testing.assert_allclose(xi, g["c_xiplus"], rtol=0, atol=1e-12 * np.abs(g["c_xiplus"]).max())
    np.testing.assert_allclose(meanlogr, g["c_logr"], rtol=0, atol=1e-12)
    np.testing.assert_array_equal(wt, npairs)
    assert np.all(np.exp(meanlogr) <= meanr * (1 + 1e-12))                   # geometric mean <= arithmetic mean, per bin
    _cmp_golden(utils.vcorr(x, y, k, np.zeros_like(k), rmin=mn, rmax=mx, dlogr=float(g["c_dlogr"])), g, "c")
    # per-point weights against the reference's binner (g9 case f): xi = <w w k k> / <w w>, weight = <w w> npairs
    xiw, wtw, _, _, npw = ops.kk_log(x, y, k, g["f_w"], mn, mx, nb)
    np.testing.assert_array_equal(npw, npairs)
    np.testing.assert_allclose(xiw, g["f_xiplus_wk"] / g["f_xiplus_w"], rtol=0, atol=1e-12 * np.abs(xiw).max())
    np.testing.assert_allclose(wtw, g["f_xiplus_w"] * npairs, rtol=1e-12)
    # uniform weights leave xi unchanged and scale the weight by w^2
    xi2, wt2 = ops.kk_log(x, y, k, np.full(len(x), 3.0), mn, mx, nb)[:2]
    np.testing.assert_allclose(xi2, xi, rtol=1e-12, atol=1e-15)
    np.testing.assert_allclose(wt2, 9.0 * npairs, rtol=1e-12)

"""GPU parity tests of the mean-function builder: tgp_binned_stat_2d against
scipy.stats.binned_statistic_2d (the call treegp/meanify.py:76-107 makes) and the `meanify` class against
oracle.meanify_grid, including the FITS round trip into GPInterpolation(average_fits=...)."""
import os
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))


def _scipy(u, v, val, ue, ve, stat):
    from scipy.stats import binned_statistic_2d
    return binned_statistic_2d(u, v, val, bins=[ue, ve], statistic=stat)[0]


def _check(u, v, val, ue, ve, err=None):
    from treegp_amd import ops
    cnt_ref = _scipy(u, v, val, ue, ve, "count")
    avg, wrms, cnt = ops.binned_stat_2d(u, v, val, ue, ve, "mean")
    np.testing.assert_array_equal(cnt, cnt_ref)                               # bin numbers: exact
    ref = _scipy(u, v, val, ue, ve, "mean")
    np.testing.assert_array_equal(np.isnan(avg), np.isnan(ref))
    np.testing.assert_allclose(avg, ref, rtol=1e-13, atol=1e-13 * np.nanmax(np.abs(ref)), equal_nan=True)
    assert np.all(wrms == 0.0)
    med, _, cnt2 = ops.binned_stat_2d(u, v, val, ue, ve, "median")
    np.testing.assert_array_equal(cnt2, cnt_ref)
    np.testing.assert_array_equal(med, _scipy(u, v, val, ue, ve, "median"))   # order statistics: bit-exact
    if err is not None:
        w = 1.0 / err ** 2
        s_wpp = _scipy(u, v, w * val * val, ue, ve, "sum")
        s_wp = _scipy(u, v, w * val, ue, ve, "sum")
        s_w = _scipy(u, v, w, ue, ve, "sum")
        with np.errstate(invalid="ignore", divide="ignore"):
            a_ref = s_wp / s_w
            r_ref = np.sqrt((1.0 / s_w) * (s_wpp - 2.0 * a_ref * s_wp + a_ref * a_ref * s_w))
        a, r, sw = ops.binned_stat_2d(u, v, val, ue, ve, "weighted", err=err)
        np.testing.assert_allclose(sw, s_w, rtol=1e-13)
        np.testing.assert_allclose(a, a_ref, rtol=1e-12, equal_nan=True)
        ok = np.isfinite(r_ref) & (r_ref > 1e-3 * np.nanmax(r_ref))
        np.testing.assert_allclose(r[ok], r_ref[ok], rtol=1e-9)


def test_binned_stat_random_field():
    rng = np.random.default_rng(1)
    n = 150000
    u, v = rng.uniform(0, 2048, n), rng.uniform(0, 2048, n)
    val = 0.02 + 5e-8 * (u - 1024) ** 2 + 0.03 * rng.standard_normal(n)
    err = 0.01 + 0.02 * rng.uniform(size=n)
    _check(u, v, val, np.linspace(u.min(), u.max(), 51), np.linspace(v.min(), v.max(), 51), err)


def test_binned_stat_edges_outliers_empty_bins():
    rng = np.random.default_rng(2)
    ue = np.linspace(-1.0, 3.0, 9)
    ve = np.array([0.0, 0.1, 0.5, 0.50001, 2.0, 7.5])               # uneven, one nearly empty bin
    n = 20000
    u = rng.uniform(-1.5, 3.5, n)                                    # some points outside on both sides
    v = rng.uniform(-0.5, 8.0, n)
    # points exactly on inner edges, on the first edge, on the last edge (goes to the last bin) and just past it
    u[:9] = ue; v[:9] = 1.0
    u[9:15] = 0.3; v[9:15] = ve
    u[15] = 3.0; v[15] = 7.5
    u[16] = np.nextafter(3.0, 4.0); v[16] = 1.0
    u[17] = 3.0 + 4e-7; v[17] = 1.0                                  # rounds to the last edge at scipy's `decimal`
    u[18] = 1.0; v[18] = np.nan
    val = rng.standard_normal(n)
    val[100:140] = 0.25                                              # ties around the median
    _check(u, v, val, ue, ve, err=0.5 + rng.uniform(size=n))
    # a region with no points at all -> nan bins
    keep = ~((u > 0) & (u < 1.5))
    _check(u[keep], v[keep], val[keep], ue, ve)


def test_binned_stat_small_and_single_bin():
    rng = np.random.default_rng(3)
    for n in (1, 2, 3, 257):
        u, v, val = rng.uniform(0, 1, n), rng.uniform(0, 1, n), rng.standard_normal(n)
        _check(u, v, val, np.array([0.0, 1.0]), np.array([0.0, 1.0]))
        _check(u, v, val, np.linspace(0, 1, 4), np.linspace(0, 1, 3))


def test_binned_stat_many_bins_global_atomics_and_big_bin():
    """more bin slots than the LDS histogram holds, and one bin far larger than the others (median of a
    big segment)"""
    rng = np.random.default_rng(4)
    n = 1 << 20
    u, v = rng.uniform(0, 1, n), rng.uniform(0, 1, n)
    u[: n // 4] = 0.5 + 1e-4 * rng.uniform(size=n // 4)
    v[: n // 4] = 0.5 + 1e-4 * rng.uniform(size=n // 4)
    val = rng.standard_normal(n)
    _check(u, v, val, np.linspace(0, 1, 201), np.linspace(0, 1, 151))
    # few bins, many points: LDS-private histogram path
    _check(u, v, val, np.linspace(0, 1, 11), np.linspace(0, 1, 9), err=0.5 + rng.uniform(size=n))


@pytest.mark.parametrize("stat", ["mean", "median", "weighted"])
def test_meanify_class_matches_oracle(stat, tmp_path):
    import treegp_amd
    from treegp_amd.fits_io import read_bintable_row
    from oracle import gp_oracle as O
    from test_oracle_golden import meanify_fixture_coords
    fields = meanify_fixture_coords(nfields=60)
    rng = np.random.default_rng(7)
    m = treegp_amd.meanify(bin_spacing=40.0, statistics=stat)
    P, E = [], []
    for c in fields:
        p = 0.02 + 5e-8 * (c[:, 0] - 1024) ** 2 + 5e-8 * (c[:, 1] - 1024) ** 2 + 0.03 * rng.standard_normal(len(c))
        e = 0.01 + 0.02 * rng.uniform(size=len(c))
        P.append(p); E.append(e)
        m.add_field(c, p, params_err=e if stat == "weighted" else None)
    m.meanify()
    ref = O.meanify_grid(np.concatenate(fields, axis=0), np.concatenate(P), np.concatenate(E), 40.0, stat)
    np.testing.assert_array_equal(m.coords0, ref["coords0"])
    np.testing.assert_array_equal(m._u0, ref["u0"])
    np.testing.assert_array_equal(m._v0, ref["v0"])
    if stat == "median":
        np.testing.assert_array_equal(m.params0, ref["params0"])
    else:
        np.testing.assert_allclose(m.params0, ref["params0"], rtol=1e-12)
    np.testing.assert_allclose(m.wrms0, ref["wrms0"], rtol=1e-9, atol=1e-15)
    np.testing.assert_allclose(m._average, ref["average"], rtol=1e-12, equal_nan=True)
    # the file GPInterpolation reads back (gp_interp.py:97-102)
    path = os.path.join(str(tmp_path), "mean_gp.fits")
    m.save_results(name_output=path)
    back = read_bintable_row(path)
    assert list(back) == ["COORDS0", "PARAMS0", "WRMS0", "_AVERAGE", "_WRMS", "_U0", "_V0"]
    np.testing.assert_array_equal(back["COORDS0"], m.coords0)
    np.testing.assert_array_equal(back["PARAMS0"], m.params0)
    np.testing.assert_array_equal(back["_AVERAGE"], m._average)
    gp = treegp_amd.GPInterpolation(kernel="0.03**2 * RBF(300.)", optimizer="none", average_fits=path)
    np.testing.assert_array_equal(gp._X0, m.coords0)
    np.testing.assert_array_equal(gp._y0, m.params0)
    X = fields[0][:200]
    gp.initialize(X, P[0][:200], y_err=E[0][:200])
    np.testing.assert_allclose(gp._spatial_average, O.knn_mean(m.coords0, m.params0, X, 4), rtol=1e-13)


def test_meanify_argument_errors():
    import treegp_amd
    with pytest.raises(ValueError):
        treegp_amd.meanify(statistics="mode")
    m = treegp_amd.meanify(statistics="weighted")
    with pytest.raises(ValueError):
        m.add_field(np.zeros((3, 2)), np.zeros(3))
    with pytest.raises(ValueError):
        treegp_amd.meanify().add_field(np.zeros((3, 1)), np.zeros(3))


@pytest.mark.parametrize("stat", ["mean", "median"])
def test_meanify_class_against_reference_golden(stat, golden):
    """g13: the treegp_amd.meanify class (tgp_binned_stat_2d) against the values the reference's own meanify produced for the
    same fields: grid, bin centres, the nan filter of empty bins, means to 1e-13, medians bit for bit."""
    import treegp_amd
    g = golden("g13_meanify.npz")
    nf = int(g["nfields"])
    for tag, lim in (("auto", {}), ("lim", dict(lu_min=100.0, lu_max=1900.0, lv_min=0.0, lv_max=2048.0))):
        key = stat + "_" + tag
        m = treegp_amd.meanify(bin_spacing=120.0, statistics=stat)
        for i in range(nf):
            m.add_field(g["coords%d" % i], g["params%d" % i])
        m.meanify(**lim)
        np.testing.assert_array_equal(m._xedge, g[key + "_xedge"])
        np.testing.assert_array_equal(m._yedge, g[key + "_yedge"])
        np.testing.assert_array_equal(m._u0, g[key + "_u0"])
        np.testing.assert_array_equal(m._v0, g[key + "_v0"])
        np.testing.assert_array_equal(m.coords0, g[key + "_coords0"])
        np.testing.assert_array_equal(np.isnan(m._average), np.isnan(g[key + "_average"]))
        if stat == "median":
            np.testing.assert_array_equal(m.params0, g[key + "_params0"])
        else:
            np.testing.assert_allclose(m.params0, g[key + "_params0"], rtol=1e-13)
        np.testing.assert_array_equal(m.wrms0, g[key + "_wrms0"])

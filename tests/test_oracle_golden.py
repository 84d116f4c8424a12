"""Pins oracle/gp_oracle.py against the golden vectors produced by the reference
(tests/golden/make_golden.py).  CPU only."""
import numpy as np

from oracle import gp_oracle as O


def _inv3(invL):
    return dict(a=invL[0, 0], b=invL[0, 1], c=invL[1, 1])


def test_kernel_tables(golden):
    g = golden("g4_kernels.npz")
    X, Y, invL = g["X"], g["Y"], g["invLam"]
    cases = {
        "rbf": dict(kind="gauss", amp=4.0, a=1 / 0.45 ** 2, b=0.0, c=1 / 0.45 ** 2),
        "arbf": dict(kind="gauss", amp=0.25, **_inv3(g["arbf_invLam"])),
        "vk": dict(kind="vk", amp=2.25, ell=3.0),
        "avk": dict(kind="avk", amp=0.7 ** 2, **_inv3(g["avk_invLam"])),
        "vk_noamp": dict(kind="vk", amp=1.0, ell=0.02),
    }
    for tag, p in cases.items():
        np.testing.assert_allclose(O.kernel_matrix(X=X, **p), g[tag + "_self"], rtol=2e-12, atol=1e-14, err_msg=tag)
        np.testing.assert_allclose(O.kernel_matrix(X=Y, Y=X, **p), g[tag + "_cross"], rtol=2e-12, atol=1e-14, err_msg=tag)
    X1 = g["X1"]
    for tag, p in {"rbf1d": dict(kind="gauss", amp=4.0, a=0.25, b=0.0, c=0.0),
                   "vk1d": dict(kind="vk", amp=4.0, ell=2.0),
                   "arbf1d": dict(kind="gauss", amp=1.0, a=0.25, b=0.0, c=0.0)}.items():
        np.testing.assert_allclose(O.kernel_matrix(X=X1, **p), g[tag + "_self"], rtol=2e-12, atol=1e-14)
        np.testing.assert_allclose(O.kernel_matrix(X=X1[:10] + 0.5, Y=X1, **p), g[tag + "_cross"], rtol=2e-12, atol=1e-14)


def test_config1_rbf1d(golden):
    g = golden("g1_c1_rbf1d.npz")
    p = dict(kind="gauss", amp=1.0, a=0.25, b=0.0, c=0.0)
    K = O.kernel_matrix(X=g["X"], **p)
    mean = np.mean(g["y"])
    np.testing.assert_allclose(mean, g["mean"], rtol=1e-15)
    alpha, logdet = O.gp_solve(K, g["y"] - mean, g["y_err"])
    np.testing.assert_allclose(alpha, g["alpha"], rtol=1e-9, atol=1e-9 * np.abs(g["alpha"]).max())
    yp = O.gp_predict(O.kernel_matrix(X=g["Xs"], Y=g["X"], **p), alpha) + mean
    np.testing.assert_allclose(yp, g["y_pred"], rtol=1e-10, atol=1e-10)
    np.testing.assert_allclose(O.log_likelihood(K, g["y"] - mean, g["y_err"]), g["logL"], rtol=1e-12)


def test_aniso2d_with_cov(golden):
    g = golden("g2_aniso2d.npz")
    p = dict(kind="gauss", amp=float(g["amp"]), **_inv3(g["invLam"]))
    yerr = np.sqrt(g["y_err"] ** 2 + float(g["white_noise"]) ** 2)
    np.testing.assert_allclose(yerr, g["y_err_eff"], rtol=1e-15)
    mean = np.mean(g["y"])
    K = O.kernel_matrix(X=g["X"], **p)
    alpha, _ = O.gp_solve(K, g["y"] - mean, yerr)
    HT = O.kernel_matrix(X=g["Xs"], Y=g["X"], **p)
    np.testing.assert_allclose(O.gp_predict(HT, alpha) + mean, g["y_pred"], rtol=1e-10, atol=1e-10)
    cov = O.gp_predict_cov(K, yerr, HT[:256], O.kernel_matrix(X=g["Xs"][:256], **p))
    np.testing.assert_allclose(cov, g["cov256"], rtol=1e-9, atol=1e-11)


def test_vonkarman_gp(golden):
    g = golden("g3_vonkarman.npz")
    mean = np.mean(g["y"])
    for tag, p in (("vk", dict(kind="vk", amp=1.69, ell=0.4)),
                   ("avk", dict(kind="avk", amp=1.69, **_inv3(g["avk_invLam"])))):
        K = O.kernel_matrix(X=g["X"], **p)
        alpha, _ = O.gp_solve(K, g["y"] - mean, g["y_err"])
        HT = O.kernel_matrix(X=g["Xs"], Y=g["X"], **p)
        np.testing.assert_allclose(O.gp_predict(HT, alpha) + mean, g[tag + "_y_pred"], rtol=1e-10, atol=1e-10)
        cov = O.gp_predict_cov(K, g["y_err"], HT[:200], O.kernel_matrix(X=g["Xs"][:200], **p))
        np.testing.assert_allclose(cov, g[tag + "_cov200"], rtol=1e-9, atol=1e-11)
        np.testing.assert_allclose(O.log_likelihood(K, g["y"] - mean, g["y_err"]), g[tag + "_logL"], rtol=1e-12)


def test_meanify_case(golden):
    g = golden("g6_meanify.npz")
    sa = O.knn_mean(g["X0"], g["y0"], g["X"], k=4)
    np.testing.assert_allclose(sa, g["spatial_average"], rtol=1e-13)
    np.testing.assert_allclose(O.knn_mean(g["X0"], g["y0"], g["Xs"], k=4), g["spatial_average_Xs"], rtol=1e-13)
    mean = np.mean(g["y"] - sa)
    np.testing.assert_allclose(mean, g["mean"], rtol=1e-13)


def test_host_scalars(golden):
    g = golden("g7_host_scalars.npz")
    for nb in (15, 20, 21):
        m = O.twod_mask(nb)
        assert np.array_equal(m, g["mask_%d" % nb])
        assert int(m.sum()) == int(g["npix_%d" % nb])
        assert O.n_bootstrap(int(m.sum())) == int(g["nboot_%d" % nb])
    assert np.array_equal(O.bootstrap_indices(10, 4), g["boot_n10"])
    assert np.array_equal(O.bootstrap_indices(1000, 3), g["boot_n1000"])
    assert int(g["nboot_21"]) == 444 and int(g["npix_21"]) == 221


def test_kk_twod_vs_naive_loop():
    """Exact pair binner against a pure-Python double loop (small n)."""
    rng = np.random.default_rng(0)
    n, nb, mx = 60, 7, 0.4
    x, y = rng.uniform(0, 1, n), rng.uniform(0, 1, n)
    k = rng.standard_normal(n); w = rng.uniform(0.5, 2, n)
    xi, wt, npairs = O.kk_twod(x, y, k, w, 0.0, mx, nb)
    bs = 2 * mx / nb
    s_wkk = np.zeros(nb * nb); s_w = np.zeros(nb * nb); s_n = np.zeros(nb * nb)
    for i in range(n):
        for j in range(i + 1, n):
            dx, dy = x[j] - x[i], y[j] - y[i]
            if dx == 0 and dy == 0:
                continue
            if max(abs(dx), abs(dy)) >= mx:
                continue
            for s in (1, -1):
                ix, iy = int((s * dx + mx) / bs), int((s * dy + mx) / bs)
                if 0 <= ix < nb and 0 <= iy < nb:
                    s_wkk[iy * nb + ix] += w[i] * w[j] * k[i] * k[j]
                    s_w[iy * nb + ix] += w[i] * w[j]
                    s_n[iy * nb + ix] += 1
    assert np.array_equal(npairs, s_n)
    np.testing.assert_allclose(wt, s_w, rtol=1e-13)
    np.testing.assert_allclose(xi[s_w > 0], (s_wkk / np.where(s_w > 0, s_w, 1))[s_w > 0], rtol=1e-12, atol=1e-15)
    # point symmetry the reference relies on (two_pcf.py:306-310)
    np.testing.assert_allclose(xi.reshape(nb, nb), xi.reshape(nb, nb)[::-1, ::-1], rtol=1e-12, atol=1e-15)


def _cmp_vcorr(got, g, tag, tol=1e-13):
    names = ("logr", "xiplus", "ximinus", "xicross", "xiz2")
    for v, nm in zip(got, names):
        ref = g[tag + "_" + nm]
        assert v.shape == ref.shape
        np.testing.assert_array_equal(np.isnan(v), np.isnan(ref), err_msg=nm)
        ok = ~np.isnan(ref)
        np.testing.assert_allclose(v[ok], ref[ok], rtol=0, atol=tol * max(1.0, np.abs(ref[ok]).max(initial=0.0)), err_msg=nm)


def test_vcorr_against_reference_binner(golden):
    """oracle.vcorr vs the reference's own treegp/utils.py vcorr (g9, generated by importing it)."""
    g = golden("g9_vcorr.npz")
    with np.errstate(invalid="ignore", divide="ignore"):
        _cmp_vcorr(O.vcorr(g["a_x"], g["a_y"], g["a_dx"], g["a_dy"], rmin=float(g["a_rmin"]), rmax=float(g["a_rmax"]),
                           dlogr=float(g["a_dlogr"])), g, "a")
        _cmp_vcorr(O.vcorr(g["b_x"], g["b_y"], g["b_dx"], g["b_dy"]), g, "b")
        # the subsampling branch, utils.py:28-35 (legacy global stream)
        n = len(g["d_x"])
        np.random.seed(int(g["d_seed"]))
        use = np.random.random(n) <= float(int(g["d_maxpts"])) / n
        _cmp_vcorr(O.vcorr(g["d_x"][use], g["d_y"][use], g["d_dx"][use], g["d_dy"][use], rmin=float(g["d_rmin"]),
                           rmax=float(g["d_rmax"]), dlogr=float(g["d_dlogr"])), g, "d")


def test_kk_log_against_reference_binner(golden):
    """oracle.kk_log (unit weights) vs the reference's vcorr run on a scalar field (dy = 0) with bins chosen to
    coincide: xi = xi+, meanlogr = logr.  Pins the Log-bin pair binner's bin assignment and sums."""
    g = golden("g9_vcorr.npz")
    x, y, k = g["c_x"], g["c_y"], g["c_k"]
    mn, mx, nb = float(g["c_min_sep"]), float(g["c_max_sep"]), int(g["c_nbins"])
    xi, wt, meanr, meanlogr, npairs = O.kk_log(x, y, k, None, mn, mx, nb)
    np.testing.assert_allclose(xi, g["c_xiplus"], rtol=0, atol=1e-13 * np.abs(g["c_xiplus"]).max())
    np.testing.assert_allclose(meanlogr, g["c_logr"], rtol=0, atol=1e-13)
    assert npairs.sum() > 1e5 and np.array_equal(wt, npairs)
    # and the same through oracle.vcorr
    _cmp_vcorr(O.vcorr(x, y, k, np.zeros_like(k), rmin=mn, rmax=mx, dlogr=float(g["c_dlogr"])), g, "c")
    # weighted sums are the same expression with w_i w_j folded in: uniform weights c leave xi unchanged
    xi2 = O.kk_log(x, y, k, np.full(len(x), 3.0), mn, mx, nb)[0]
    np.testing.assert_allclose(xi2, xi, rtol=1e-13, atol=1e-16)
    # per-point weights, pinned by the reference's binner too (g9 case f): weighted xi = <w w k k> / <w w>, weight = <w w> npairs
    w = g["f_w"]
    xiw, wtw, _, _, npw = O.kk_log(x, y, k, w, mn, mx, nb)
    np.testing.assert_array_equal(npw, npairs)
    np.testing.assert_allclose(xiw, g["f_xiplus_wk"] / g["f_xiplus_w"], rtol=0, atol=1e-12 * np.abs(xiw).max())
    np.testing.assert_allclose(wtw, g["f_xiplus_w"] * npairs, rtol=1e-12)


def test_sklearn_kernel_trees_host_side(golden):
    """g10: for kernel trees outside the parametrised device kernels the product evaluates ``kernel.__call__`` on the host and
    hands K to the device; here the same host evaluation + the oracle's Cholesky reproduce the reference's alpha,
    predictions, covariance and log-likelihood (pins the host half of the dense path without a GPU)."""
    import treegp_amd as treegp
    g = golden("g10_sklearn_kernels.npz")
    X, y, y_err, Xs = g["X"], g["y"], g["y_err"], g["Xs"]
    mean = np.mean(y)
    for tag in ("sumwhite", "matern", "rq"):         # "sum2" holds an AnisotropicRBF leaf, whose __call__ is the GPU's (S1)
        k = treegp.eval_kernel(str(g[tag + "_kernel"]))
        K = k(X)
        alpha, _ = O.gp_solve(K, y - mean, y_err)
        np.testing.assert_allclose(alpha, g[tag + "_alpha"], rtol=0, atol=1e-9 * np.abs(g[tag + "_alpha"]).max())
        HT = k(Xs, Y=X)
        np.testing.assert_allclose(O.gp_predict(HT, alpha) + mean, g[tag + "_y_pred"], rtol=1e-10, atol=1e-10)
        cov = O.gp_predict_cov(K, y_err, HT[:128], k(Xs[:128]))
        np.testing.assert_allclose(cov, g[tag + "_cov128"], rtol=1e-9, atol=1e-11)
        np.testing.assert_allclose(O.log_likelihood(K, y - mean, y_err), g[tag + "_logL"], rtol=1e-12)


def meanify_fixture_coords(nfields=300, ndata=500):
    """The star positions behind the reference's tests/inputs/mean_gp_stat_mean.fits: legacy
    np.random.seed(42) stream of its tests/test_meanify.py:43-55 -- per field 500 uniform x, 500 uniform
    y, then the 500 normals that np.random.multivariate_normal draws (values not needed here)."""
    np.random.seed(42)
    coords = []
    for _ in range(nfields):
        x = np.random.uniform(0, 2048, size=ndata)
        y = np.random.uniform(0, 2048, size=ndata)
        np.random.standard_normal(ndata)
        coords.append(np.array([x, y]).T)
    return coords


def test_meanify_grid_geometry_against_reference_fixture(golden):
    """COORDS0 of the reference's own meanify output (bin_spacing=40, 50 x 50 bins, all populated) pins the
    edges, the bin-centre formula and the flattening order of oracle.meanify_grid."""
    g = golden("g6_meanify.npz")
    coords = np.concatenate(meanify_fixture_coords(), axis=0)
    # the reference's smooth mean function (test_meanify.py:27) stands in for the GRF draws
    params = 0.02 + 5e-8 * (coords[:, 0] - 1024) ** 2 + 5e-8 * (coords[:, 1] - 1024) ** 2
    for stat in ("mean", "median"):
        m = O.meanify_grid(coords, params, bin_spacing=40.0, statistics=stat)
        np.testing.assert_array_equal(m["coords0"], g["X0"])
        assert m["average"].shape == (50, 50)
        # the fixture's PARAMS0 are GRF realisations around that function (reference tolerance 2e-1, :67)
        np.testing.assert_allclose(m["params0"], g["y0"], atol=2e-1)
    err = 0.01 + 0.02 * np.random.default_rng(0).uniform(size=len(params))
    w = O.meanify_grid(coords, params, params_err=err, bin_spacing=40.0, statistics="weighted")
    np.testing.assert_array_equal(w["coords0"], g["X0"])
    assert np.all(w["wrms0"] >= 0) and np.all(np.isfinite(w["params0"]))


def test_meanify_values_against_reference(golden):
    """g13: treegp/meanify.py run by the reference itself (mean and median, default and explicit limits, a hole that leaves
    empty bins): oracle.meanify_grid reproduces its grid, bin centres, filtered coordinates and values."""
    g = golden("g13_meanify.npz")
    nf = int(g["nfields"])
    coords = np.concatenate([g["coords%d" % i] for i in range(nf)], axis=0)
    params = np.concatenate([g["params%d" % i] for i in range(nf)])
    for stat in ("mean", "median"):
        for tag, lim in (("auto", {}), ("lim", dict(lu_min=100.0, lu_max=1900.0, lv_min=0.0, lv_max=2048.0))):
            key = stat + "_" + tag
            m = O.meanify_grid(coords, params, bin_spacing=120.0, statistics=stat, **lim)
            np.testing.assert_array_equal(m["xedge"], g[key + "_xedge"])
            np.testing.assert_array_equal(m["yedge"], g[key + "_yedge"])
            np.testing.assert_array_equal(m["u0"], g[key + "_u0"])
            np.testing.assert_array_equal(m["coords0"], g[key + "_coords0"])
            np.testing.assert_allclose(m["average"], g[key + "_average"], rtol=1e-14, equal_nan=True)
            np.testing.assert_allclose(m["params0"], g[key + "_params0"], rtol=1e-14)
            assert np.isnan(g[key + "_average"]).any() and len(m["params0"]) < m["average"].size


def test_loglik_gradient_against_reference_kernel_derivative(golden):
    """g14: 1/2 tr((alpha alpha^T - K^-1) dK/dtheta) with the reference's own dK/dtheta.  The oracle's restatement, driven by the
    host chain rule of the product (kernels.spec_jacobian: d(log amp, a, b, c)/dtheta), reproduces it -- pins both without a GPU."""
    import treegp_amd as treegp
    from treegp_amd.kernels import kernel_to_spec, spec_jacobian
    g = golden("g14_loglik_grad.npz")
    for tag in ("arbf2d", "arbf1d", "rbf2d"):
        k = treegp.eval_kernel(str(g[tag + "_kernel"]))
        np.testing.assert_allclose(k.theta, g[tag + "_theta"], rtol=1e-13)
        spec, J = kernel_to_spec(k), spec_jacobian(k)
        assert J.shape == (len(k.theta), 4)
        X = g[tag + "_X"]
        nd = X.shape[1]
        invLam = np.array([[spec.a, spec.b], [spec.b, spec.c]])[:nd, :nd]
        basis = [np.array([[1.0, 0], [0, 0]])[:nd, :nd], np.array([[0, 1.0], [1.0, 0]])[:nd, :nd], np.array([[0, 0], [0, 1.0]])[:nd, :nd]]
        g_amp, g_abc = O.loglik_grad_invlam(X, g[tag + "_y"], g[tag + "_y_err"], spec.amp, invLam, basis)
        grad = J.dot(np.concatenate([[g_amp], g_abc]))
        ref = g[tag + "_grad"]
        np.testing.assert_allclose(grad, ref, rtol=1e-7, atol=1e-7 * np.abs(ref).max(), err_msg=tag)


def test_kernel_derivative_host_arithmetic_against_reference(golden):
    """g14 kg_*: kernel(X, eval_gradient=True) of the reference.  The product's classes take K from the device and form dK/dtheta on
    the host; here the same host arithmetic is fed the reference's K (no GPU) and has to reproduce the reference's dK."""
    import treegp_amd as treegp
    g = golden("g14_loglik_grad.npz")
    X = g["kg_X"]
    k = treegp.eval_kernel(str(g["kg_arbf_bare_kernel"]))
    np.testing.assert_allclose(k._gradient(X, g["kg_arbf_bare_K"]), g["kg_arbf_bare_dK"], rtol=1e-13, atol=1e-15)
    k = treegp.eval_kernel(str(g["kg_arbf2d_kernel"]))                    # Product(ConstantKernel, AnisotropicRBF)
    unit = g["kg_arbf2d_K"] / k.k1.constant_value
    np.testing.assert_allclose(k.k1.constant_value * k.k2._gradient(X, unit), g["kg_arbf2d_dK"][:, :, 1:], rtol=1e-13, atol=1e-15)
    np.testing.assert_allclose(g["kg_arbf2d_dK"][:, :, 0], g["kg_arbf2d_K"], rtol=1e-14)       # d/d log sigma^2
    k = treegp.eval_kernel(str(g["kg_arbf1d_kernel"]))
    unit = g["kg_arbf1d_K"] / k.k1.constant_value
    np.testing.assert_allclose(k.k1.constant_value * k.k2._gradient(X[:, :1], unit), g["kg_arbf1d_dK"][:, :, 1:], rtol=1e-13, atol=1e-15)

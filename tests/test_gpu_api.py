"""GPU tests through the treegp-compatible Python API.  They read like the reference's own tests
(tests/test_kernels.py, test_gp_interp.py, test_hyp_search.py, test_meanify.py) plus exact
comparisons with the golden vectors the reference produced (tests/golden/*.npz)."""
import numpy as np
import pytest
from scipy import special

import treegp_amd as treegp
from treegp_amd.synthetic import correlation_length_matrix as get_correlation_length_matrix

pytestmark = pytest.mark.gpu

REL = 1e-10          # north-star tolerance on predicted values (relative to the field scale)


def _close_rel(a, b, rel=REL):
    np.testing.assert_allclose(a, b, rtol=0, atol=rel * max(np.abs(b).max(), 1e-300))


# ---------------------------------------------------------------- golden vectors ---------------
def test_golden_g1_config1(golden):
    g = golden("g1_c1_rbf1d.npz")
    gp = treegp.GPInterpolation(kernel=str(g["kernel"]), optimizer="none", normalize=True, white_noise=0.0)
    gp.initialize(g["X"], g["y"], y_err=g["y_err"])
    _close_rel(gp.predict(g["Xs"]), g["y_pred"])
    _close_rel(gp._alpha, g["alpha"], 1e-9)
    np.testing.assert_allclose(gp.return_log_likelihood(), g["logL"], rtol=1e-11)


def test_golden_g2_aniso_cov(golden):
    g = golden("g2_aniso2d.npz")
    gp = treegp.GPInterpolation(kernel=str(g["kernel"]), optimizer="none", normalize=True, white_noise=0.01)
    gp.initialize(g["X"], g["y"], y_err=g["y_err"])
    np.testing.assert_allclose(gp._y_err, g["y_err_eff"], rtol=1e-15)
    yp, cov = gp.predict(g["Xs"][:256], return_cov=True)
    _close_rel(yp, g["y_pred"][:256])
    np.testing.assert_allclose(cov, g["cov256"], rtol=0, atol=1e-9 * np.abs(g["cov256"]).max())
    _close_rel(gp.predict(g["Xs"]), g["y_pred"])            # alpha cache path


def test_golden_g3_vonkarman(golden):
    g = golden("g3_vonkarman.npz")
    for tag in ("vk", "avk"):
        gp = treegp.GPInterpolation(kernel=str(g[tag + "_kernel"]), optimizer="none", normalize=True)
        gp.initialize(g["X"], g["y"], y_err=g["y_err"])
        _close_rel(gp.predict(g["Xs"]), g[tag + "_y_pred"])
        yp, cov = gp.predict(g["Xs"][:200], return_cov=True)
        np.testing.assert_allclose(cov, g[tag + "_cov200"], rtol=0, atol=1e-9 * np.abs(g[tag + "_cov200"]).max())
        np.testing.assert_allclose(gp.return_log_likelihood(), g[tag + "_logL"], rtol=1e-11)


def test_golden_g4_kernel_tables(golden):
    g = golden("g4_kernels.npz")
    for tag in ("rbf", "arbf", "vk", "avk", "vk_noamp"):
        k = treegp.eval_kernel(str(g[tag + "_str"]))
        if tag == "rbf":
            from treegp_amd import ops
            spec = treegp.kernel_to_spec(k)      # sklearn's own RBF.__call__ is host code; check the device path
            Kself, Kx = ops.kernel_matrix(spec, g["X"]), ops.kernel_matrix(spec, g["Y"], g["X"])
        else:
            Kself, Kx = k(g["X"]), k(g["Y"], Y=g["X"])
        np.testing.assert_allclose(Kself, g[tag + "_self"], rtol=2e-12, atol=1e-12, err_msg=tag)   # test_kernels.py atol
        np.testing.assert_allclose(Kx, g[tag + "_cross"], rtol=2e-12, atol=1e-12, err_msg=tag)
    for tag in ("vk1d", "arbf1d"):
        k = treegp.eval_kernel(str(g[tag + "_str"]))
        np.testing.assert_allclose(k(g["X1"]), g[tag + "_self"], rtol=2e-12, atol=1e-12)
        np.testing.assert_allclose(k(g["X1"][:10] + 0.5, Y=g["X1"]), g[tag + "_cross"], rtol=2e-12, atol=1e-12)


def test_golden_g5_loglike(golden):
    g = golden("g5_loglike.npz")
    gp = treegp.GPInterpolation(kernel=str(g["kernel"]), optimizer="none", normalize=True)
    gp.initialize(g["X"], g["y"], y_err=g["y_err"])
    for t, ll in zip(g["thetas"], g["logL"]):
        np.testing.assert_allclose(gp.return_log_likelihood(theta=t), ll, rtol=1e-11)
    gp2 = treegp.GPInterpolation(kernel="1.0**2 * AnisotropicRBF(scale_length=[50., 50.])", optimizer="none", normalize=False)
    gp2.initialize(g["X"], g["y"], y_err=np.zeros(len(g["y"])))
    assert gp2.return_log_likelihood() == -np.inf == float(g["logL_singular"])      # log_likelihood.py:38-39


def test_golden_g6_meanify(golden, tmp_path):
    from treegp_amd.fits_io import write_bintable_row
    g = golden("g6_meanify.npz")
    p = str(tmp_path / "mean.fits")
    write_bintable_row(p, {"COORDS0": g["X0"], "PARAMS0": g["y0"]})
    gp = treegp.GPInterpolation(kernel=str(g["kernel"]), optimizer="none", normalize=True, n_neighbors=4, average_fits=p)
    gp.initialize(g["X"], g["y"], y_err=g["y_err"])
    np.testing.assert_allclose(gp._spatial_average, g["spatial_average"], rtol=1e-13)
    np.testing.assert_allclose(gp._mean, g["mean"], rtol=1e-12)
    _close_rel(gp.predict(g["Xs"]), g["y_pred"])


def test_golden_g8_reference_test_problems(golden):
    g = golden("g8_reftests.npz")
    x = g["x"]
    for tag in ("rbf", "vk"):
        kern = str(g[tag + "_kernel"])
        gp = treegp.GPInterpolation(kernel=kern, optimizer="none", white_noise=0.0)
        gp.initialize(x, g[tag + "_y"], y_err=0.1 * np.ones(len(x)))
        yp, cov = gp.predict(x, return_cov=True)
        _close_rel(yp, g[tag + "_y_pred"])
        np.testing.assert_allclose(cov, g[tag + "_cov"], rtol=0, atol=1e-9 * np.abs(g[tag + "_cov"]).max())
        gpb = treegp.GPInterpolation(kernel=kern, optimizer="none", normalize=False, white_noise=0.0)
        gpb.initialize(x, g[tag + "_y"], y_err=0.1 * np.ones(len(x)))
        ypb, covb = gpb.predict(g[tag + "_new_x"], return_cov=True)
        np.testing.assert_allclose(ypb, g[tag + "_y_far"], rtol=0, atol=1e-12)
        np.testing.assert_allclose(covb, g[tag + "_cov_far"], rtol=0, atol=1e-9 * np.abs(g[tag + "_cov_far"]).max())


# ---------------------------------------------------------------- tests/test_kernels.py --------
def test_anisotropic_kernels_closed_form():
    corr_length = [1.0, 30.0, 30.0, 30.0, 30.0]
    g1 = [0, 0.4, 0.4, -0.4, -0.4]
    g2 = [0, 0.4, -0.4, 0.4, -0.4]
    kernel_amp = [1e-4, 1e-3, 1e-2, 1.0, 1.0]
    dist = np.linspace(0, 10, 100)
    coord = np.array([dist, dist]).T
    d21 = np.linspace(-10, 10, 21)
    XX, YY = np.meshgrid(d21, d21)
    x, y = XX.reshape(-1), YY.reshape(-1)
    coord_corr = np.array([x, y]).T
    lim0 = special.gamma(5.0 / 6.0) / (2 * (np.pi ** (5.0 / 6.0)))
    for i in range(5):
        inv_L = np.linalg.inv(get_correlation_length_matrix(corr_length[i], g1[i], g2[i]))
        q = inv_L[0, 0] * x * x + 2 * inv_L[0, 1] * x * y + inv_L[1, 1] * y * y
        ker = kernel_amp[i] ** 2 * treegp.AnisotropicRBF(invLam=inv_L)
        np.testing.assert_allclose(ker(coord_corr, Y=np.zeros_like(coord_corr))[:, 0],
                                   kernel_amp[i] ** 2 * np.exp(-0.5 * q), atol=1e-12)
        dd = coord[:, None, :] - coord[None, :, :]
        qq = np.einsum("ijk,kl,ijl->ij", dd, inv_L, dd)
        Kt = kernel_amp[i] ** 2 * np.exp(-0.5 * qq)
        np.testing.assert_allclose(ker(coord), Kt, atol=1e-12)
        kvk = kernel_amp[i] ** 2 * treegp.AnisotropicVonKarman(invLam=inv_L)
        z = np.ones_like(q)
        nz = q != 0
        z[nz] = q[nz] ** (5.0 / 12.0) * special.kv(5.0 / 6.0, 2 * np.pi * np.sqrt(q[nz])) / lim0
        np.testing.assert_allclose(kvk(coord_corr, Y=np.zeros_like(coord_corr))[:, 0], kernel_amp[i] ** 2 * z, atol=1e-12)


def test_vonkarman_kernel_closed_form():
    dist = np.linspace(0.01, 10, 100)
    coord_corr = np.array([dist, np.zeros_like(dist)]).T
    div = 5.0 / 6.0
    lim0 = (2 * (np.pi ** div)) / special.gamma(div)
    for corr in [1.0, 10.0, 100.0, 1000.0]:
        for amp in [1e-4, 1e-3, 1e-2, 1.0]:
            kernel = "%.10f * VonKarman(length_scale=%f)" % ((amp ** 2, corr))
            interp = treegp.GPInterpolation(kernel=kernel, normalize=False, white_noise=0.0)
            ker = interp.kernel_template
            got = ker(coord_corr, Y=np.zeros_like(coord_corr))[:, 0]
            want = amp ** 2 * lim0 * ((dist / corr) ** (5.0 / 6.0)) * special.kv(-5.0 / 6.0, 2 * np.pi * dist / corr)
            np.testing.assert_allclose(got, want, atol=1e-12)


def test_anisotropic_limit():
    np.random.seed(42)
    gp1 = treegp.GPInterpolation(kernel="RBF(0.45)")
    gp2 = treegp.GPInterpolation(kernel="AnisotropicRBF(scale_length=[0.45, 0.45])")
    X = np.random.rand(1000, 2)
    np.testing.assert_allclose(gp1.kernel_template(X), gp2.kernel_template(X))
    from treegp_amd import ops
    np.testing.assert_allclose(ops.kernel_matrix(treegp.kernel_to_spec(gp1.kernel_template), X), gp2.kernel_template(X), rtol=1e-12)
    k3 = treegp.eval_kernel("VonKarman(0.45)")
    k4 = treegp.eval_kernel("AnisotropicVonKarman(scale_length=[0.45, 0.45])")
    np.testing.assert_allclose(k3(X), k4(X), rtol=1e-11, atol=1e-14)


# ---------------------------------------------------------------- tests/test_gp_interp.py ------
def _grf(kernel_skl, noise, npoints, ndim, seed=42):
    """Gaussian random field drawn the way tests/treegp_test_helper.py:47-104 draws it (legacy
    NumPy global stream, uniform coordinates in [-10, 10], optional white noise)."""
    np.random.seed(seed)
    if ndim == 1:
        x = np.random.uniform(-10, 10, npoints).reshape((npoints, 1))
    else:
        x1 = np.random.uniform(-10, 10, npoints)
        x2 = np.random.uniform(-10, 10, npoints)
        x = np.array([x1, x2]).T
    K = kernel_skl(x)
    y = np.random.multivariate_normal(np.zeros(npoints), K)
    if noise is not None:
        y += np.random.normal(scale=noise, size=npoints)
        return x, y, np.ones_like(y) * noise
    return x, y, None


def _check_interp(kernel, x, y, y_err, white_noise, sigma, new_x, noise):
    npoints = len(y)
    gp = treegp.GPInterpolation(kernel=kernel, optimizer="none", white_noise=white_noise)
    gp.initialize(x, y, y_err=y_err)
    y_predict, y_cov = gp.predict(x, return_cov=True)
    y_std = np.sqrt(np.diag(y_cov))
    pull = y - y_predict
    if noise is not None:
        pull /= np.sqrt(y_err ** 2 + y_std ** 2)
    else:
        np.testing.assert_allclose(y, y_predict, atol=3.0 * white_noise)
        np.testing.assert_allclose(np.zeros_like(y_std), y_std, atol=3.0 * white_noise)
    np.testing.assert_allclose(0.0, np.mean(pull), atol=3.0 * np.std(pull) / np.sqrt(npoints))
    assert np.std(pull) <= 1.0
    gp = treegp.GPInterpolation(kernel=kernel, optimizer="none", normalize=False, white_noise=white_noise)
    gp.initialize(x, y, y_err=y_err)
    y_predict, y_cov = gp.predict(new_x, return_cov=True)
    y_std = np.sqrt(np.diag(y_cov))
    np.testing.assert_allclose(np.zeros_like(y_predict), y_predict, atol=1e-5)
    np.testing.assert_allclose(sigma * np.ones_like(y_std), y_std, atol=1e-5)


def test_gp_interp_1d():
    npoints = 40
    noise = [None, 0.1]
    white_noise = [1e-5, 0.0]
    sigma = [1.0, 2.0]
    l = [2.0, 2.0]
    for ker in ["RBF", "VonKarman"]:
        for i in range(2):
            kernel = "%f**2 * %s(%f)" % ((sigma[i], ker, l[i]))
            x, y, y_err = _grf(treegp.eval_kernel(kernel), noise[i], npoints, 1)
            new_x = np.linspace(np.max(x) + 6.0 * l[i], np.max(x) + 7.0 * l[i], npoints).reshape((npoints, 1))
            _check_interp(kernel, x, y, y_err, white_noise[i], sigma[i], new_x, noise[i])


def test_gp_interp_2d():
    npoints = 200
    noise = [None, 0.1]
    white_noise = [1e-5, 0.0]
    size = [2.0, 4.0]
    g1 = [0.0, 0.2]
    g2 = [0.0, 0.2]
    for ker in ["AnisotropicRBF", "AnisotropicVonKarman"]:
        for i in range(2):
            invL = np.linalg.inv(get_correlation_length_matrix(size[i], g1[i], g2[i]))
            kernel = "%f**2*%s" % ((1.0, ker)) + "(invLam={0!r})".format(invL)
            x, y, y_err = _grf(treegp.eval_kernel(kernel), noise[i], npoints, 2)
            far = np.max(x) + 6.0 * size[i]
            new_x = np.full((npoints, 2), far)
            _check_interp(kernel, x, y, y_err, white_noise[i], 1.0, new_x, noise[i])


# ---------------------------------------------------------------- tests/test_hyp_search.py -----
def test_hyperparameter_search_loglikelihood():
    for ker, sig, ell in (("RBF", 1.0, 0.5), ("RBF", 2.0, 0.8), ("VonKarman", 1.0, 8.0), ("VonKarman", 2.0, 10.0)):
        kernel = "%f**2 * %s(%f)" % (sig, ker, ell)
        kernel_skl = treegp.eval_kernel(kernel)
        x, y, y_err = _grf(kernel_skl, 0.01, 100, 1)
        gp = treegp.GPInterpolation(kernel=kernel, optimizer="log-likelihood", normalize=True)
        gp.initialize(x, y, y_err=y_err)
        gp.solve()
        np.testing.assert_allclose(kernel_skl.theta, gp.kernel.theta, atol=7e-1)          # test_hyp_search.py:43
        np.testing.assert_allclose(gp.return_log_likelihood(), gp._optimizer._logL, atol=1e-10)   # :49-50
        assert gp._alpha is None
    invL = np.linalg.inv(get_correlation_length_matrix(0.5, 0.2, 0.2))
    kernel = "%f**2*%s" % (2.0, "AnisotropicRBF") + "(invLam={0!r})".format(invL)
    kernel_skl = treegp.eval_kernel(kernel)
    x, y, y_err = _grf(kernel_skl, 0.01, 600, 2)
    gp = treegp.GPInterpolation(kernel=kernel, optimizer="log-likelihood", normalize=True)
    gp.initialize(x, y, y_err=y_err)
    gp.solve()
    np.testing.assert_allclose(kernel_skl.theta, gp.kernel.theta, atol=5e-1)               # :141


def test_ml_fit_concurrent_differences_follow_the_sequential_path():
    """The ntheta+1 likelihood evaluations of one finite-difference gradient run concurrently (one context each):
    same formula and step as SciPy's own numerical gradient, so the fit visits the same iterates."""
    import time
    from treegp_amd.log_likelihood import log_likelihood
    invL = np.linalg.inv(get_correlation_length_matrix(0.5, 0.2, 0.2))
    kernel = treegp.eval_kernel("%f**2*%s" % (2.0, "AnisotropicRBF") + "(invLam={0!r})".format(invL))
    x, y, y_err = _grf(kernel, 0.01, 600, 2)
    start = kernel.clone_with_theta(kernel.theta + np.array([0.3, -0.2, 0.25, 0.1]))
    fits, walls = [], []
    for parallel in (False, True):
        ll = log_likelihood(x, y - np.mean(y), y_err)
        ll.parallel_fd = parallel
        t0 = time.perf_counter()
        fitted = ll.optimizer(start)
        walls.append(time.perf_counter() - t0)
        fits.append((fitted.theta, ll._logL))
    np.testing.assert_allclose(fits[1][0], fits[0][0], rtol=0, atol=1e-6)
    np.testing.assert_allclose(fits[1][1], fits[0][1], rtol=1e-10)
    np.testing.assert_allclose(fits[0][0], kernel.theta, atol=5e-1)
    print("ML fit N=600: sequential %.1f ms, concurrent differences %.1f ms" % (walls[0] * 1e3, walls[1] * 1e3))


def test_resident_problem_matches_host_boundary_solve():
    """ops.ResidentProblem / gp_solve_resident (what the evaluations of one ML fit use): same numbers as gp_solve,
    from several contexts, with and without errors, 1-D and 2-D coordinates."""
    from treegp_amd import _lib, ops
    rng = np.random.default_rng(21)
    for n, ndim, with_err in ((300, 1, True), (777, 2, True), (1500, 2, False)):
        X = rng.uniform(0, 1, (n, ndim)); y = rng.standard_normal(n); e = rng.uniform(0.05, 0.2, n) if with_err else None
        spec = ops.KernelSpec(_lib.TGP_ARBF, amp=1.1, a=50.0, b=(5.0 if ndim == 2 else 0.0), c=(70.0 if ndim == 2 else 1.0))
        if not with_err:
            spec = ops.KernelSpec(_lib.TGP_ARBF, amp=1.1, a=4000.0, b=300.0, c=5000.0)   # narrow: positive definite without a noise diagonal
        _, logdet, chi2, _ = ops.gp_solve(spec, X, y, e, want_alpha=False)
        prob = ops.ResidentProblem(X, y, e)
        for ctx in (None, _lib.new_ctx(0)):
            ld, c2 = ops.gp_solve_resident(spec, prob, ctx=ctx)
            np.testing.assert_allclose([ld, c2], [logdet, chi2], rtol=1e-13)
        prob.close()
    # not positive definite: same exception as the host-boundary call
    X = np.zeros((64, 2)); prob = ops.ResidentProblem(X, np.ones(64), None)
    with pytest.raises(np.linalg.LinAlgError):
        ops.gp_solve_resident(ops.KernelSpec(_lib.TGP_ARBF, amp=1.0, a=1.0, b=0.0, c=1.0), prob)
    prob.close()


def test_hyperparameter_search_two_pcf_1d():
    for ker, sig, ell, max_sep in (("RBF", 1.0, 0.5, 1.75), ("RBF", 2.0, 0.8, 1.75), ("VonKarman", 1.0, 8.0, 1.25),
                                   ("VonKarman", 2.0, 10.0, 1.25)):
        kernel = "%f**2 * %s(%f)" % (sig, ker, ell)
        kernel_skl = treegp.eval_kernel(kernel)
        x, y, y_err = _grf(kernel_skl, 0.01, 2000, 1)
        gp = treegp.GPInterpolation(kernel=kernel, optimizer="two-pcf", normalize=True, nbins=15, min_sep=0.1, max_sep=max_sep)
        gp.initialize(x, y, y_err=y_err)
        gp.solve()
        np.testing.assert_allclose(kernel_skl.theta, gp.kernel.theta, atol=7e-1)
        xi, xi_weight, distance, coord, mask = gp.return_2pcf()
        np.testing.assert_allclose(xi, gp._optimizer._2pcf, atol=1e-10)                     # :46-47


def test_hyperparameter_search_anisotropic():
    invL = np.linalg.inv(get_correlation_length_matrix(0.5, 0.2, 0.2))
    kernel = "%f**2*%s" % (2.0, "AnisotropicRBF") + "(invLam={0!r})".format(invL)
    kernel_skl = treegp.eval_kernel(kernel)
    x, y, y_err = _grf(kernel_skl, 0.01, 2000, 2)
    gp = treegp.GPInterpolation(kernel=kernel, optimizer="anisotropic", normalize=True, nbins=21, min_sep=0.0,
                                max_sep=1.0, p0=[0.3, 0.0, 0.0])
    gp.initialize(x, y, y_err=y_err)
    gp.solve()
    assert gp._optimizer._2pcf_weight.shape == (221, 221)
    np.testing.assert_allclose(kernel_skl.theta, gp.kernel.theta, atol=5e-1)
    y_predict, y_cov = gp.predict(x, return_cov=True)
    pull = (y - y_predict) / np.sqrt(y_err ** 2 + np.diag(y_cov))
    np.testing.assert_allclose(0.0, np.mean(pull), atol=3.0 * np.std(pull) / np.sqrt(2000))
    assert np.std(pull) <= 1.0


# ---------------------------------------------------------------- pair binning vs oracle -------
def test_kk_twod_and_log_vs_oracle():
    from oracle import gp_oracle as O
    from treegp_amd import ops
    rng = np.random.default_rng(3)
    for n in (50, 257, 1500):
        x, y = rng.uniform(0, 1, n), rng.uniform(0, 1, n)
        k = rng.standard_normal(n)
        for w in (None, rng.uniform(0.5, 2.0, n)):
            for nb, mn, mx in ((7, 0.0, 0.4), (21, 0.05, 0.15), (20, 0.0, 0.3)):
                xi, wt, npairs = ops.kk_twod(x, y, k, w, mn, mx, nb)
                xo, wo, no = O.kk_twod(x, y, k, w, mn, mx, nb)
                assert np.array_equal(npairs, no)                                    # exact pair counts
                np.testing.assert_allclose(wt, wo, rtol=1e-12)
                np.testing.assert_allclose(xi, xo, rtol=1e-10, atol=1e-12 * np.abs(xo).max())
            xi, wt, mr, mlr, npairs = ops.kk_log(x, y, k, w, 0.01, 0.5, 15)
            xo, wo, ro, lo, no = O.kk_log(x, y, k, w, 0.01, 0.5, 15)
            assert np.array_equal(npairs, no)
            np.testing.assert_allclose(wt, wo, rtol=1e-12)
            np.testing.assert_allclose(xi, xo, rtol=1e-10, atol=1e-12 * np.abs(xo).max())
            np.testing.assert_allclose(mr, ro, rtol=1e-12)
            np.testing.assert_allclose(mlr, lo, rtol=1e-11)


def test_kk_bootstrap_vs_oracle_loop():
    from oracle import gp_oracle as O
    from treegp_amd import ops
    rng = np.random.default_rng(5)
    n, nb, mx = 600, 9, 0.3
    X = rng.uniform(0, 1, (n, 2)); yv = rng.standard_normal(n) + 1.0; yerr = rng.uniform(0.05, 0.1, n)
    idx = O.bootstrap_indices(n, 5)
    got = ops.kk_twod_bootstrap(X[:, 0], X[:, 1], yv, yerr, idx, 0.0, mx, nb)
    got_unw = ops.kk_twod_bootstrap(X[:, 0], X[:, 1], yv, np.zeros(n), idx, 0.0, mx, nb)
    for b in range(5):
        ii = idx[b]
        xo, _, _, _ = O.comp_2pcf(X[ii], yv[ii], yerr[ii], 0.0, mx, nb, True)
        np.testing.assert_allclose(got[b], xo, rtol=1e-10, atol=1e-12 * np.abs(xo).max())
        xo, _, _, _ = O.comp_2pcf(X[ii], yv[ii], np.zeros(n), 0.0, mx, nb, True)
        np.testing.assert_allclose(got_unw[b], xo, rtol=1e-10, atol=1e-12 * np.abs(xo).max())
    # the host class reproduces the reference's bootstrap stream and covariance recipe
    t = treegp.two_pcf(X, yv, yerr, 0.0, mx, nbins=nb, anisotropic=True)
    cov = t.comp_xi_covariance(n_bootstrap=5, mask=O.twod_mask(nb))
    sel = got[:, O.twod_mask(nb)]
    d = sel - sel.mean(axis=0)
    np.testing.assert_allclose(cov, d.T @ d / 4.0, rtol=1e-9, atol=1e-14)


@pytest.mark.parametrize("shift", [0.0, 250.0])
def test_kk_bootstrap_pair_lists_vs_oracle(shift, monkeypatch):
    """n_boot >= 8 takes the pair-list formulation (kk_boot.hip): same resamples as the oracle's loop, and the same
    numbers as the per-resample kernels (TGP_BOOT_LISTS=0); a large common offset of the values must not matter."""
    from oracle import gp_oracle as O
    from treegp_amd import ops
    rng = np.random.default_rng(11)
    n, nb, mx, nboot = 700, 11, 0.25, 70
    X = rng.uniform(0, 1, (n, 2)); yv = rng.standard_normal(n) + shift; yerr = rng.uniform(0.05, 0.1, n)
    X[5] = X[6]                                     # a coincident pair (r = 0, excluded)
    idx = O.bootstrap_indices(n, nboot)
    tol = 1e-10 if shift == 0.0 else 1e-8
    for err in (yerr, np.zeros(n)):
        monkeypatch.setenv("TGP_BOOT_LISTS", "1")
        got = ops.kk_twod_bootstrap(X[:, 0], X[:, 1], yv, err, idx, 0.0, mx, nb)
        monkeypatch.setenv("TGP_BOOT_LISTS", "0")
        old = ops.kk_twod_bootstrap(X[:, 0], X[:, 1], yv, err, idx, 0.0, mx, nb)
        scale = np.abs(old).max()
        np.testing.assert_allclose(got, old, rtol=tol, atol=tol * 1e-2 * scale)
        for b in (0, 33, nboot - 1):
            ii = idx[b]
            xo, _, _, _ = O.comp_2pcf(X[ii], yv[ii], err[ii], 0.0, mx, nb, True)
            np.testing.assert_allclose(got[b], xo, rtol=tol, atol=tol * 1e-2 * np.abs(xo).max())
    # a min_sep > 0 and a pixel grid that leaves pixels empty
    monkeypatch.setenv("TGP_BOOT_LISTS", "1")
    got = ops.kk_twod_bootstrap(X[:, 0], X[:, 1], yv, yerr, idx[:9], 0.05, 0.02 + 0.05, 5)
    monkeypatch.setenv("TGP_BOOT_LISTS", "0")
    old = ops.kk_twod_bootstrap(X[:, 0], X[:, 1], yv, yerr, idx[:9], 0.05, 0.02 + 0.05, 5)
    np.testing.assert_allclose(got, old, rtol=tol, atol=tol * 1e-2 * max(np.abs(old).max(), 1e-300))


def test_knn_mean_vs_oracle(golden):
    from oracle import gp_oracle as O
    from treegp_amd import ops
    g = golden("g6_meanify.npz")
    rng = np.random.default_rng(2)
    Xq = rng.uniform(0, 2048, (5000, 2))
    for k in (1, 2, 4, 7, 8, 16):
        np.testing.assert_allclose(ops.knn_mean(g["X0"], g["y0"], Xq, k), O.knn_mean(g["X0"], g["y0"], Xq, k), rtol=1e-13)
    np.testing.assert_allclose(ops.knn_mean(g["X0"], g["y0"], g["X"], 4), g["spatial_average"], rtol=1e-13)


def test_kk_edge_cases_vs_oracle():
    """regular grids put many pairs exactly on pixel edges; duplicates give r == 0; tiny catalogues"""
    from oracle import gp_oracle as O
    from treegp_amd import ops
    gx, gy = np.meshgrid(np.linspace(0, 1, 33), np.linspace(0, 1, 29))
    x, y = gx.ravel(), gy.ravel()
    x = np.concatenate([x, x[:40]]); y = np.concatenate([y, y[:40]])          # coincident points
    rng = np.random.default_rng(9)
    k = rng.standard_normal(len(x))
    for nb, mn, mx in ((8, 0.0, 0.25), (21, 0.03125, 0.15625), (5, 0.0, 2.0)):
        xi, wt, npairs = ops.kk_twod(x, y, k, None, mn, mx, nb)
        xo, wo, no = O.kk_twod(x, y, k, None, mn, mx, nb)
        assert np.array_equal(npairs, no)
        np.testing.assert_allclose(xi, xo, rtol=1e-10, atol=1e-12 * np.abs(xo).max())
    xi, wt, mr, mlr, npairs = ops.kk_log(x, y, k, None, 1 / 32, 0.5, 12)
    xo, wo, ro, lo, no = O.kk_log(x, y, k, None, 1 / 32, 0.5, 12)
    assert np.array_equal(npairs, no)
    for n in (2, 3, 17):
        xs, ys, ks = rng.uniform(0, 1, n), rng.uniform(0, 1, n), rng.standard_normal(n)
        xi, wt, npairs = ops.kk_twod(xs, ys, ks, None, 0.0, 0.7, 5)
        xo, wo, no = O.kk_twod(xs, ys, ks, None, 0.0, 0.7, 5)
        assert np.array_equal(npairs, no)
        np.testing.assert_allclose(xi, xo, rtol=1e-10, atol=1e-14)


def test_robust_2dfit_model_and_chi2_against_reference(golden):
    """g12: robust_2dfit._model_skl and chi2 (treegp/two_pcf.py:96-148) for both anisotropic kernels at the parameter sets
    the reference itself evaluated (including |g| > 1 and a NaN: chi2 = inf)."""
    import sys
    T = sys.modules["treegp_amd.two_pcf"]
    g = golden("g12_two_pcf_host.npz")
    for tag in ("arbf", "avk"):
        kern = treegp.eval_kernel(str(g[tag + "_kernel"]))
        fit = T.robust_2dfit(kern, g[tag + "_data"], g["fit_x"], g["fit_y"], g["fit_W"], mask=g["fit_mask"])
        for q, chi2, alpha, model in zip(g["fit_trial"], g[tag + "_chi2"], g[tag + "_alpha"], g[tag + "_model"]):
            got = fit.chi2(q)
            if np.isfinite(chi2):
                np.testing.assert_allclose(fit._model_skl(1.0, *q), model, rtol=0, atol=1e-12)
                np.testing.assert_allclose(got, chi2, rtol=1e-8)
                np.testing.assert_allclose(np.ravel(fit.alpha), alpha, rtol=1e-8)
            else:
                assert got == np.inf


def test_ml_fit_reaches_the_reference_optimum(golden):
    """g11: the reference's own maximum-likelihood fits (L-BFGS-B with SciPy's finite differences, log_likelihood.py:43-62);
    the fit here starts from the same kernel and has to end at the same optimum: log-likelihood within 1e-6 of the
    reference's (relative), theta within 2e-3 (the finite-difference gradient's noise floor)."""
    g = golden("g11_ml_fit.npz")
    for tag, yerr in (("rbf1d", 0.01), ("arbf2d", 0.02)):
        X, y = g[tag + "_X"], g[tag + "_y"]
        gp = treegp.GPInterpolation(kernel=str(g[tag + "_kernel0"]), optimizer="log-likelihood", normalize=True)
        gp.initialize(X, y, y_err=yerr * np.ones(len(y)))
        gp.solve()
        ref_l = float(g[tag + "_logL"])
        assert gp._optimizer._logL >= ref_l - 1e-6 * abs(ref_l), (gp._optimizer._logL, ref_l)
        np.testing.assert_allclose(gp.kernel.theta, g[tag + "_theta"], atol=2e-3)
        # and the likelihood at the reference's optimum is the reference's number
        np.testing.assert_allclose(gp.return_log_likelihood(theta=g[tag + "_theta"]), ref_l, rtol=1e-10)

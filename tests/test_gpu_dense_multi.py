"""GPU parity tests of the rows added in round 2: caller-evaluated kernel matrices (any scikit-learn kernel tree), several
fields against one factor, and the big-step triangular sweeps against the 128-block chain."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

REL = 1e-10          # north star: predicted values within 1e-10 relative of the reference


def _rel(a, b):
    return np.abs(a - b).max() / np.abs(b).max()


@pytest.mark.parametrize("tag", ["sumwhite", "matern", "rq", "sum2"])
def test_sklearn_kernel_trees_against_reference_golden(golden, tag):
    """Sum / WhiteKernel / Matern / RationalQuadratic trees (treegp/kernels.py:17-59 evals any scikit-learn kernel) through
    GPInterpolation: kernel.__call__ on the host, factorisation / solve / covariance on the device (tgp_gp_solve_dense,
    tgp_gp_predict_cov_dense), against values produced by the reference itself (g10)."""
    import treegp_amd as treegp
    g = golden("g10_sklearn_kernels.npz")
    kern = str(g[tag + "_kernel"])
    with pytest.raises(NotImplementedError):   # none of these trees is one of the parametrised device kernels
        treegp.kernel_to_spec(treegp.eval_kernel(kern))
    gp = treegp.GPInterpolation(kernel=kern, optimizer="none", normalize=True, white_noise=0.0)
    gp.initialize(g["X"], g["y"], y_err=g["y_err"])
    yp, cov = gp.predict(g["Xs"][:128], return_cov=True)
    ref = g[tag + "_y_pred"]
    assert _rel(yp, ref[:128]) < REL
    assert _rel(gp.predict(g["Xs"]), ref) < REL
    np.testing.assert_allclose(gp._alpha, g[tag + "_alpha"], rtol=0, atol=1e-9 * np.abs(g[tag + "_alpha"]).max())
    np.testing.assert_allclose(cov, g[tag + "_cov128"], rtol=0, atol=1e-9 * np.abs(g[tag + "_cov128"]).max())
    np.testing.assert_allclose(gp.return_log_likelihood(), float(g[tag + "_logL"]), rtol=1e-11)
    # a second covariance request reuses the factor; one with other arguments does not (gp_interp.py:186-187)
    f0 = gp._factor
    gp.predict(g["Xs"][:64], return_cov=True)
    assert gp._factor is f0


def test_dense_entry_point_matches_parametrised_path():
    from oracle import gp_oracle as O
    from treegp_amd import _lib, ops
    from treegp_amd.synthetic import star_field, headline_invlam
    for n in (300, 1000, 2600):
        X, y, y_err, Xs = star_field(n, 500, seed=n)
        iL = headline_invlam()
        kw = dict(amp=1.3, a=iL[0, 0], b=iL[0, 1], c=iL[1, 1])
        spec = ops.KernelSpec(_lib.TGP_ARBF, **kw)
        K = O.kernel_matrix("gauss", X, **kw)
        a0, ld0, yd0, _ = ops.gp_solve(spec, X, y - y.mean(), y_err)
        a1, ld1, yd1, f = ops.gp_solve_dense(K, y - y.mean(), y_err, keep=True)
        np.testing.assert_allclose(a1, a0, rtol=0, atol=1e-11 * np.abs(a0).max())
        np.testing.assert_allclose([ld1, yd1], [ld0, yd0], rtol=1e-12)
        # upper triangle is never read
        Kl = np.tril(K) + np.triu(np.full_like(K, np.nan), 1)
        a2 = ops.gp_solve_dense(Kl, y - y.mean(), y_err)[0]
        np.testing.assert_array_equal(a2, a1)
        HT = O.kernel_matrix("gauss", Xs, X, **kw)
        cov = ops.gp_predict_cov_dense(f, HT, O.kernel_matrix("gauss", Xs, **kw))
        ref = O.gp_predict_cov(K, y_err, HT, O.kernel_matrix("gauss", Xs, **kw))
        np.testing.assert_allclose(cov, ref, rtol=0, atol=1e-10 * np.abs(ref).max())
        f.free()
    with pytest.raises(np.linalg.LinAlgError):
        ops.gp_solve_dense(-np.eye(40), np.ones(40), None)


@pytest.mark.parametrize("n,nf", [(900, 3), (2600, 7), (5000, 4)])
def test_factor_solve_many_right_hand_sides(n, nf):
    """tgp_factor_solve: every row against numpy.linalg.solve on the oracle's K (chain path below 2048 rows, big-step sweeps
    with groups of 4 / 2 / 1 right-hand sides above)."""
    from oracle import gp_oracle as O
    from treegp_amd import _lib, ops
    from treegp_amd.synthetic import star_field, headline_invlam
    X, y, y_err, _ = star_field(n, 8, seed=3 * n)
    iL = headline_invlam()
    kw = dict(amp=1.0, a=iL[0, 0], b=iL[0, 1], c=iL[1, 1])
    rng = np.random.default_rng(n)
    B = rng.standard_normal((nf, n))
    B[0] = y - y.mean()
    alpha, _, _, f = ops.gp_solve(ops.KernelSpec(_lib.TGP_ARBF, **kw), X, B[0], y_err, keep=True)
    got = ops.factor_solve(f, B)
    got2 = ops.factor_solve(f, B[1:2])                          # the cached slabs serve later calls
    f.free()
    K = O.kernel_matrix("gauss", X, **kw) + np.diag(y_err ** 2)
    ref = np.linalg.solve(K, B.T).T
    for v in range(nf):
        np.testing.assert_allclose(got[v], ref[v], rtol=0, atol=1e-9 * np.abs(ref[v]).max())
    np.testing.assert_allclose(got[0], alpha, rtol=0, atol=1e-12 * np.abs(alpha).max())
    np.testing.assert_array_equal(got2[0], got[1])


@pytest.mark.parametrize("kern", ["1.0**2 * AnisotropicRBF(scale_length=[0.08, 0.05])", "0.8**2 * VonKarman(length_scale=0.3)",
                                  "1.0**2 * RBF(0.1) + WhiteKernel(1e-4)"])
def test_predict_fields_equals_one_gp_per_field(kern):
    """GPInterpolation.predict_fields (one K build + one factorisation for all fields) against one GPInterpolation per field,
    which is what a treegp user writes today (README.rst:28)."""
    import treegp_amd as treegp
    rng = np.random.default_rng(8)
    n, m, nf = 2500, 400, 5
    X = rng.uniform(0, 1, (n, 2))
    Xs = rng.uniform(0, 1, (m, 2))
    Y = np.stack([np.sin((3 + v) * X[:, 0]) * np.cos((2 + v) * X[:, 1]) + 0.3 * v + 0.02 * rng.standard_normal(n) for v in range(nf)])
    y_err = 0.02 * rng.uniform(0.8, 1.2, n)
    gp = treegp.GPInterpolation(kernel=kern, optimizer="none", normalize=True, white_noise=1e-3)
    gp.initialize(X, Y[0], y_err=y_err)
    got = gp.predict_fields(Y, Xs)
    assert got.shape == (nf, m)
    for v in range(nf):
        one = treegp.GPInterpolation(kernel=kern, optimizer="none", normalize=True, white_noise=1e-3)
        one.initialize(X, Y[v], y_err=y_err)
        ref = one.predict(Xs)
        assert _rel(got[v], ref) < REL, (v, _rel(got[v], ref))


@pytest.mark.parametrize("n", [2048, 2300, 2900, 5000, 9000])
def test_big_step_sweeps_match_block_chain(n, monkeypatch):
    """trsv_big.hip (inverse slabs by recursive doubling, 1024-row steps; also 512) against the 128-block chain of
    trsv.hip on the same factor, full solve and forward-only (likelihood) path, ragged last super-blocks included."""
    import ctypes as C
    from treegp_amd import _lib, ops
    from treegp_amd.synthetic import star_field, headline_invlam
    lib, ctx = _lib.load_library(), _lib.get_ctx()
    X, y, y_err, _ = star_field(n, 8, seed=n)
    iL = headline_invlam()
    spec = ops.KernelSpec(_lib.TGP_ARBF, amp=1.0, a=iL[0, 0], b=iL[0, 1], c=iL[1, 1])
    Np = lib.tgp_padded_n(n)
    dX = ops.DeviceBuffer.from_array(ctx, X); de = ops.DeviceBuffer.from_array(ctx, y_err)
    dA = ops.DeviceBuffer(ctx, lib.tgp_panel_elems(Np) * 8); dW = ops.DeviceBuffer(ctx, Np * 128 * 8)
    _lib.check(ctx, lib.tgp_d_kbuild_lower(ctx, C.byref(spec.to_c()), dX.ptr, n, de.ptr, dA.ptr), "kbuild")
    assert lib.tgp_d_potrf(ctx, dA.ptr, Np, dW.ptr) == 0
    rhs = np.zeros(Np); rhs[:n] = y - y.mean()

    def solve():
        db = ops.DeviceBuffer.from_array(ctx, rhs)
        _lib.check(ctx, lib.tgp_d_potrs(ctx, dA.ptr, dW.ptr, Np, db.ptr), "potrs")
        out = db.to_array(Np)
        db.free()
        return out
    monkeypatch.setenv("TGP_POTRS_BIG_FROM", "0")
    ref = solve()
    monkeypatch.setenv("TGP_POTRS_BIG_FROM", "256")
    for step in ("1024", "512"):
        monkeypatch.setenv("TGP_POTRS_STEP", step)
        got = solve()
        assert np.abs(got - ref).max() <= 1e-11 * np.abs(ref).max(), (step, np.abs(got - ref).max() / np.abs(ref).max())
        assert np.all(got[n:] == 0.0)
    monkeypatch.delenv("TGP_POTRS_STEP")
    # likelihood path: forward sweep only (n a multiple of 256 has no spare padding row for the augmented right-hand side)
    monkeypatch.setenv("TGP_POTRS_BIG_FROM", "0")
    l0 = ops.gp_solve(spec, X, y - y.mean(), y_err, want_alpha=False)[1:3]
    monkeypatch.setenv("TGP_POTRS_BIG_FROM", "256")
    monkeypatch.setenv("TGP_NO_AUGMENT", "1")
    l1 = ops.gp_solve(spec, X, y - y.mean(), y_err, want_alpha=False)[1:3]
    np.testing.assert_allclose(l1, l0, rtol=1e-12)
    for b in (dX, de, dA, dW):
        b.free()


def test_ml_fit_with_a_scikit_learn_only_kernel():
    """optimizer="log-likelihood" on a kernel tree the device kernels cannot describe (Sum with a fitted WhiteKernel): every
    likelihood evaluation goes through the dense route; the result is checked against the same likelihood evaluated with
    NumPy / SciPy on the host at the fitted theta, and the fit has to improve on its starting point."""
    import treegp_amd as treegp
    from scipy.linalg import cho_factor, cho_solve
    rng = np.random.default_rng(17)
    n = 300
    X = rng.uniform(0, 1, (n, 2))
    truth = treegp.eval_kernel("0.8**2 * RBF(0.2) + WhiteKernel(0.05**2)")
    y = rng.multivariate_normal(np.zeros(n), truth(X))
    y_err = np.full(n, 1e-3)
    gp = treegp.GPInterpolation(kernel="0.5**2 * RBF(0.4) + WhiteKernel(0.2**2)", optimizer="log-likelihood", normalize=False)
    gp.initialize(X, y, y_err=y_err)
    l0 = gp.return_log_likelihood()
    gp.solve()
    l1 = gp.return_log_likelihood()
    assert l1 > l0 + 1.0

    def host_logl(kernel):
        K = kernel(X) + np.diag(y_err ** 2)
        c = cho_factor(K, lower=True)
        return -0.5 * y @ cho_solve(c, y) - 0.5 * n * np.log(2 * np.pi) - np.sum(np.log(np.diag(c[0])))
    np.testing.assert_allclose(l1, host_logl(gp.kernel), rtol=1e-10)
    np.testing.assert_allclose(gp._optimizer._logL, l1, rtol=1e-12)
    # the fitted noise level is in the right decade
    assert 0.01 < np.sqrt(gp.kernel.k2.noise_level) < 0.2


def test_device_level_multi_rhs_entry_point():
    """tgp_d_potrs_multi on device pointers (what a multi-GPU driver or a resident pipeline calls): every row equals
    tgp_d_potrs on that row, below and above the big-step threshold."""
    import ctypes as C
    from treegp_amd import _lib, ops
    from treegp_amd.synthetic import star_field, headline_invlam
    lib, ctx = _lib.load_library(), _lib.get_ctx()
    iL = headline_invlam()
    spec = ops.KernelSpec(_lib.TGP_ARBF, amp=1.0, a=iL[0, 0], b=iL[0, 1], c=iL[1, 1])
    for n, nrhs in ((700, 3), (3300, 6)):
        X, y, y_err, _ = star_field(n, 8, seed=n)
        Np = lib.tgp_padded_n(n)
        dX = ops.DeviceBuffer.from_array(ctx, X); de = ops.DeviceBuffer.from_array(ctx, y_err)
        dA = ops.DeviceBuffer(ctx, lib.tgp_panel_elems(Np) * 8); dW = ops.DeviceBuffer(ctx, Np * 128 * 8)
        _lib.check(ctx, lib.tgp_d_kbuild_lower(ctx, C.byref(spec.to_c()), dX.ptr, n, de.ptr, dA.ptr), "kbuild")
        assert lib.tgp_d_potrf(ctx, dA.ptr, Np, dW.ptr) == 0
        B = np.zeros((nrhs, Np))
        B[:, :n] = np.random.default_rng(n).standard_normal((nrhs, n))
        dB = ops.DeviceBuffer.from_array(ctx, B)
        _lib.check(ctx, lib.tgp_d_potrs_multi(ctx, dA.ptr, dW.ptr, Np, dB.ptr, nrhs), "potrs_multi")
        got = dB.to_array((nrhs, Np))
        for v in range(nrhs):
            db = ops.DeviceBuffer.from_array(ctx, B[v])
            _lib.check(ctx, lib.tgp_d_potrs(ctx, dA.ptr, dW.ptr, Np, db.ptr), "potrs")
            ref = db.to_array(Np)
            db.free()
            np.testing.assert_allclose(got[v], ref, rtol=0, atol=1e-12 * np.abs(ref).max())
        for b in (dX, de, dA, dW, dB):
            b.free()


@pytest.mark.parametrize("n,m", [(2300, 300), (3100, 700), (4096, 513)])
def test_posterior_covariance_big_step_substitution(n, m, monkeypatch):
    """return_cov from N = 2048 on: the substitution Bt = HT L^-T in 1024-column steps with the factor's inverse slabs and
    depth-1024 updates, against the oracle (gp_interp.py:184-192) and against the 128-block substitution on the same factor."""
    from oracle import gp_oracle as O
    from treegp_amd import _lib, ops
    from treegp_amd.synthetic import star_field, headline_invlam
    X, y, y_err, Xs = star_field(n, m, seed=n + m)
    iL = headline_invlam()
    kw = dict(amp=0.9, a=iL[0, 0], b=iL[0, 1], c=iL[1, 1])
    spec = ops.KernelSpec(_lib.TGP_ARBF, **kw)
    _, _, _, f = ops.gp_solve(spec, X, y - y.mean(), y_err, keep=True)
    cov = ops.gp_predict_cov(spec, f, X, Xs)
    monkeypatch.setenv("TGP_COV_BIG", "0")
    old = ops.gp_predict_cov(spec, f, X, Xs)
    monkeypatch.delenv("TGP_COV_BIG")
    f.free()
    K = O.kernel_matrix("gauss", X, **kw)
    ref = O.gp_predict_cov(K, y_err, O.kernel_matrix("gauss", Xs, X, **kw), O.kernel_matrix("gauss", Xs, **kw))
    np.testing.assert_allclose(cov, ref, rtol=0, atol=1e-9 * np.abs(ref).max())
    np.testing.assert_allclose(cov, old, rtol=0, atol=1e-11 * np.abs(old).max())
    np.testing.assert_allclose(cov, cov.T, rtol=0, atol=1e-13 * np.abs(cov).max())

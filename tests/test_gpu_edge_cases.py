"""Edge cases of the C-ABI on the GPU: tiny / ragged sizes, padding boundaries, error codes."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _spec():
    from treegp_amd import _lib, ops
    return _lib, ops, ops.KernelSpec(_lib.TGP_ARBF, amp=1.3, a=30.0, b=4.0, c=20.0)


@pytest.mark.parametrize("n", [1, 2, 3, 127, 128, 129, 255, 256, 257, 511, 513, 1025])
def test_ragged_sizes_against_oracle(n):
    from oracle import gp_oracle as O
    _lib, ops, spec = _spec()
    rng = np.random.default_rng(n)
    X = rng.uniform(0, 1, (n, 2)); y = rng.standard_normal(n); e = rng.uniform(0.05, 0.2, n)
    kw = dict(amp=spec.amp, a=spec.a, b=spec.b, c=spec.c)
    alpha, logdet, ydota, fac = ops.gp_solve(spec, X, y, e, keep=True)
    a_ref, ld_ref = O.gp_solve(O.kernel_matrix("gauss", X, **kw), y, e)
    np.testing.assert_allclose(alpha, a_ref, rtol=0, atol=1e-10 * np.abs(a_ref).max())
    np.testing.assert_allclose(logdet, ld_ref, rtol=1e-11, atol=1e-12)
    for m in (1, 5, 130):
        Xs = rng.uniform(0, 1, (m, 2))
        yp = ops.gp_predict(spec, X, alpha, Xs)
        HT = O.kernel_matrix("gauss", Xs, X, **kw)
        np.testing.assert_allclose(yp, HT @ a_ref, rtol=0, atol=1e-10 * max(np.abs(HT @ a_ref).max(), 1e-3))
        cov = ops.gp_predict_cov(spec, fac, X, Xs)
        cref = O.gp_predict_cov(O.kernel_matrix("gauss", X, **kw), e, HT, O.kernel_matrix("gauss", Xs, **kw))
        np.testing.assert_allclose(cov, cref, rtol=0, atol=1e-10 * spec.amp)
    fac.free()


def test_one_dimensional_inputs_and_shapes():
    from oracle import gp_oracle as O
    _lib, ops, _ = _spec()
    spec = ops.KernelSpec(_lib.TGP_VK, amp=2.0, ell=1.5)
    rng = np.random.default_rng(0)
    X = rng.uniform(-5, 5, (77, 1)); y = np.sin(X[:, 0]); e = 0.05 * np.ones(77)
    alpha, logdet, _, _ = ops.gp_solve(spec, X, y, e)
    K = O.kernel_matrix("vk", X, amp=2.0, ell=1.5)
    a_ref, ld_ref = O.gp_solve(K, y, e)
    np.testing.assert_allclose(alpha, a_ref, rtol=0, atol=1e-10 * np.abs(a_ref).max())
    # y_err = None means zero noise (gp_interp.py:208-210); needs white noise to stay PD -> LinAlgError here
    Xd = np.vstack([X, X[:3]])
    with pytest.raises(np.linalg.LinAlgError):
        ops.gp_solve(spec, Xd, np.concatenate([y, y[:3]]), None)
    with pytest.raises(ValueError):
        ops.gp_solve(spec, np.zeros((5, 3)), np.zeros(5), None)          # only 1-D / 2-D coordinates


def test_error_codes_do_not_poison_the_context():
    _lib, ops, spec = _spec()
    lib = _lib.load_library()
    ctx = _lib.get_ctx()
    rc = lib.tgp_gp_predict(ctx, None, None, 0, None, None, 0, None)
    assert rc < 0 and b"bad argument" in lib.tgp_last_error(ctx)
    x = np.zeros(1)
    with pytest.raises(_lib.TgpError):
        ops.kk_twod(x, x, x, None, 0.0, 1.0, 5)                           # needs at least one pair
    with pytest.raises(_lib.TgpError):
        ops.kk_twod(np.zeros(4), np.zeros(4), np.zeros(4), None, 0.0, 1.0, 64)   # nbins too large
    # the context still works
    rng = np.random.default_rng(1)
    X = rng.uniform(0, 1, (50, 2))
    alpha, _, _, _ = ops.gp_solve(spec, X, rng.standard_normal(50), 0.1 * np.ones(50))
    assert np.all(np.isfinite(alpha))


@pytest.mark.parametrize("n", [300, 3000])
def test_nan_coordinate_or_parameter_is_not_positive_definite(n):
    """A NaN coordinate (or NaN invLam) must poison K and stop the factorisation -- scipy.linalg.cholesky's check_finite
    raises at treegp/gp_interp.py:181 -- instead of silently counting the point as uncorrelated (the clamp of
    the K build's exponential dropped NaN); a wildly indefinite invLam must not wrap the exponent field into garbage."""
    _lib, ops, spec = _spec()
    rng = np.random.default_rng(n)
    X = rng.uniform(0, 1, (n, 2)); y = rng.standard_normal(n); e = rng.uniform(0.05, 0.2, n)
    Xn = X.copy()
    Xn[n // 2, 1] = np.nan
    with pytest.raises(np.linalg.LinAlgError):
        ops.gp_solve(spec, Xn, y, e)
    with pytest.raises(np.linalg.LinAlgError):
        ops.gp_solve(ops.KernelSpec(_lib.TGP_ARBF, amp=1.0, a=np.nan, b=0.0, c=1.0), X, y, e)
    with pytest.raises(np.linalg.LinAlgError):
        ops.gp_solve(ops.KernelSpec(_lib.TGP_ARBF, amp=1.0, a=-1e6, b=0.0, c=-1e6), X, y, e)
    K = ops.kernel_matrix(spec, Xn[:64])
    assert np.isnan(K[n // 2 if n // 2 < 64 else 0]).any() or n // 2 >= 64
    alpha, _, _, _ = ops.gp_solve(spec, X, y, e)                  # and the context is fine afterwards
    assert np.isfinite(alpha).all()

"""GPU parity tests of the likelihood gradient (include/tgp.h seam S2d): K^-1 formed on the device from the kept
factor, reduced against dK/dp evaluated from the coordinates, chain rule to theta on the host."""
import os
import sys

import numpy as np
import pytest

import treegp_amd as treegp
from treegp_amd import _lib, ops
from treegp_amd.kernels import kernel_to_spec, spec_jacobian

pytestmark = pytest.mark.gpu


def _ll_class():
    return sys.modules["treegp_amd.log_likelihood"].log_likelihood


@pytest.mark.parametrize("tag", ["arbf2d", "arbf1d", "rbf2d", "arbf2d_big"])
def test_gradient_against_reference_kernel_derivative(golden, tag):
    """g14: the reference's dK/dtheta (kernel(X, eval_gradient=True), kernels.py:128-150) in 1/2 tr((alpha alpha^T - K^-1) dK);
    n = 700 / 300 / 520 go through the 128-column substitution, n = 2300 through the 1024-column steps."""
    g = golden("g14_loglik_grad.npz")
    k = treegp.eval_kernel(str(g[tag + "_kernel"]))
    ll = _ll_class()(g[tag + "_X"], g[tag + "_y"], g[tag + "_y_err"])
    value, grad = ll.log_likelihood_gradient(k)
    ref = g[tag + "_grad"]
    np.testing.assert_allclose(value, float(g[tag + "_logL"]), rtol=1e-10)
    np.testing.assert_allclose(grad, ref, rtol=1e-7, atol=1e-7 * np.abs(ref).max())
    assert ll.log_likelihood(k) == pytest.approx(value, rel=1e-12)


@pytest.mark.parametrize("n", [1000, 3000, 4096, 16640])
def test_gradient_against_central_differences_of_the_device_likelihood(n):
    rng = np.random.default_rng(n)
    X = rng.uniform(0, 1, (n, 2))
    y = np.sin(7 * X[:, 0]) * np.cos(5 * X[:, 1]) + 0.05 * rng.standard_normal(n)
    y_err = 0.05 * rng.uniform(0.8, 1.2, n)
    k = treegp.eval_kernel("0.9**2 * AnisotropicRBF(invLam=array([[95., 21.], [21., 140.]]))")
    ll = _ll_class()(X, y, y_err)
    _, grad = ll.log_likelihood_gradient(k)
    theta = k.theta.copy()
    h = 1e-5
    for i in range(len(theta)):
        tp, tm = theta.copy(), theta.copy()
        tp[i] += h
        tm[i] -= h
        fd = (ll.log_likelihood(k.clone_with_theta(tp)) - ll.log_likelihood(k.clone_with_theta(tm))) / (2 * h)
        assert abs(fd - grad[i]) <= 2e-5 * max(np.abs(grad).max(), 1.0), (i, fd, grad[i])


def test_bigstep_and_block_substitution_agree():
    """TGP_COV_BIG=0 is the 128-block substitution; the default from n = 2048 is the 1024-column one with the inverse slabs."""
    rng = np.random.default_rng(5)
    n = 2600
    X = rng.uniform(0, 1, (n, 2))
    y = rng.standard_normal(n)
    y_err = 0.1 * np.ones(n)
    spec = kernel_to_spec(treegp.eval_kernel("1.0**2 * AnisotropicRBF(invLam=array([[300., 40.], [40., 200.]]))"))
    alpha, _, _, fac = ops.gp_solve(spec, X, y, y_err, keep=True)
    g_big = ops.gp_loglik_grad(spec, fac, X, alpha)
    os.environ["TGP_COV_BIG"] = "0"
    try:
        g_blk = ops.gp_loglik_grad(spec, fac, X, alpha)
    finally:
        del os.environ["TGP_COV_BIG"]
        fac.free()
    np.testing.assert_allclose(g_big, g_blk, rtol=1e-9, atol=1e-9 * np.abs(g_blk).max())
    assert np.array_equal(g_big, ops.gp_loglik_grad(spec, *_refactor(spec, X, y, y_err)))     # run to run: bit-identical


def _refactor(spec, X, y, y_err):
    alpha, _, _, fac = ops.gp_solve(spec, X, y, y_err, keep=True)
    return fac, X, alpha


def test_von_karman_has_no_analytic_derivative():
    rng = np.random.default_rng(1)
    X = rng.uniform(0, 1, (300, 2))
    y = rng.standard_normal(300)
    k = treegp.eval_kernel("1.0**2 * VonKarman(length_scale=0.3)")
    spec = kernel_to_spec(k)
    alpha, _, _, fac = ops.gp_solve(spec, X, y, 0.1 * np.ones(300), keep=True)
    with pytest.raises(_lib.TgpError, match="Gaussian"):
        ops.gp_loglik_grad(spec, fac, X, alpha)
    fac.free()
    with pytest.raises(NotImplementedError):
        spec_jacobian(k)


def test_fit_with_the_analytic_gradient_reaches_the_reference_optimum(golden, monkeypatch):
    """g11: the reference's fits (finite differences); L-BFGS-B given the exact gradient ends at the same optimum."""
    monkeypatch.setenv("TGP_ML_GRADIENT", "analytic")
    g = golden("g11_ml_fit.npz")
    for tag, yerr in (("rbf1d", 0.01), ("arbf2d", 0.02)):
        gp = treegp.GPInterpolation(kernel=str(g[tag + "_kernel0"]), optimizer="log-likelihood", normalize=True)
        gp.initialize(g[tag + "_X"], g[tag + "_y"], y_err=yerr * np.ones(len(g[tag + "_y"])))
        gp.solve()
        assert gp._optimizer.gradient == "analytic"
        ref_l = float(g[tag + "_logL"])
        assert gp._optimizer._logL >= ref_l - 1e-6 * abs(ref_l), (gp._optimizer._logL, ref_l)
        np.testing.assert_allclose(gp.kernel.theta, g[tag + "_theta"], atol=2e-3)


@pytest.mark.parametrize("n", [1, 2, 63, 128, 129, 255, 256, 257, 300, 513, 2049])
def test_gradient_against_the_oracle_at_ragged_sizes(n):
    """Partial tiles, partial panels, one point, one point past a panel / a 1024-step; 2-D sheared and 1-D kernels."""
    from oracle import gp_oracle as O
    rng = np.random.default_rng(100 + n)
    for nd, kern in ((2, "0.7**2 * AnisotropicRBF(invLam=array([[40., -9.], [-9., 25.]]))"), (1, "1.1**2 * AnisotropicRBF(scale_length=[0.2])")):
        X = rng.uniform(0, 1, (n, nd))
        y = np.sin(5 * X[:, 0]) + 0.1 * rng.standard_normal(n)
        y_err = 0.1 * rng.uniform(0.8, 1.2, n)
        k = treegp.eval_kernel(kern)
        spec = kernel_to_spec(k)
        alpha, _, _, fac = ops.gp_solve(spec, X, y, y_err, keep=True)
        g4 = ops.gp_loglik_grad(spec, fac, X, alpha)
        fac.free()
        invLam = np.array([[spec.a, spec.b], [spec.b, spec.c]])[:nd, :nd]
        basis = [np.array([[1.0, 0], [0, 0]])[:nd, :nd], np.array([[0, 1.0], [1.0, 0]])[:nd, :nd], np.array([[0, 0], [0, 1.0]])[:nd, :nd]]
        g_amp, g_abc = O.loglik_grad_invlam(X, y, y_err, spec.amp, invLam, basis)
        ref = np.concatenate([[g_amp], g_abc])
        np.testing.assert_allclose(g4, ref, rtol=1e-8, atol=1e-8 * max(np.abs(ref).max(), 1.0), err_msg="n=%d nd=%d" % (n, nd))


def test_kernel_call_with_eval_gradient_against_reference(golden):
    """g14 kg_*: ``kernel(X, eval_gradient=True)`` through scikit-learn's Product with K from the device: AnisotropicRBF's
    derivative (kernels.py:128-150), VonKarman's K * distance (kernels.py:278-288), and the errors of the other paths."""
    g = golden("g14_loglik_grad.npz")
    X = g["kg_X"]
    for tag in g["kg_tags"]:
        k = treegp.eval_kernel(str(g[tag + "_kernel"]))
        K, dK = k(X[:, :1] if tag == "kg_arbf1d" else X, eval_gradient=True)
        np.testing.assert_allclose(K, g[tag + "_K"], rtol=1e-12, atol=1e-14, err_msg=tag)
        assert dK.shape == g[tag + "_dK"].shape
        np.testing.assert_allclose(dK, g[tag + "_dK"], rtol=1e-11, atol=1e-13, err_msg=tag)
    with pytest.raises(ValueError, match="can not be evaluated"):
        treegp.eval_kernel("AnisotropicVonKarman(invLam=array([[2., 0.], [0., 3.]]))")(X, eval_gradient=True)
    with pytest.raises(ValueError, match="only be evaluated when Y is None"):
        treegp.eval_kernel("AnisotropicRBF(scale_length=[0.5, 0.2])")(X, Y=X, eval_gradient=True)


def test_fused_resident_call_equals_the_two_step_route():
    """tgp_d_gp_solve_grad (data on the device, factor back to the context's cache) = tgp_gp_solve(keep) + tgp_gp_loglik_grad,
    bit for bit; repeated calls with other kernels reuse the cache; a failing factorisation raises as gp_solve does."""
    rng = np.random.default_rng(9)
    n = 900
    X = rng.uniform(0, 1, (n, 2))
    y = np.cos(4 * X[:, 0]) + 0.05 * rng.standard_normal(n)
    y_err = 0.05 * rng.uniform(0.8, 1.2, n)
    prob = ops.ResidentProblem(X, y, y_err)
    try:
        for kern in ("0.9**2 * AnisotropicRBF(invLam=array([[95., 21.], [21., 140.]]))", "1.4**2 * RBF(0.11)",
                     "0.9**2 * AnisotropicRBF(invLam=array([[60., -5.], [-5., 80.]]))"):
            spec = kernel_to_spec(treegp.eval_kernel(kern))
            ld, chi2, g4 = ops.gp_solve_grad_resident(spec, prob)
            alpha, ld2, chi22, fac = ops.gp_solve(spec, X, y, y_err, keep=True)
            g4b = ops.gp_loglik_grad(spec, fac, X, alpha)
            fac.free()
            assert ld == ld2 and chi2 == chi22
            assert np.array_equal(g4, g4b)
        bad = ops.ResidentProblem(np.vstack([X[:5], X[:5]]), np.ones(10), None)       # duplicate points, no noise: singular
        with pytest.raises(np.linalg.LinAlgError):
            ops.gp_solve_grad_resident(spec, bad)
        bad.close()
        ld3, _, g4c = ops.gp_solve_grad_resident(spec, prob)                          # the context still works afterwards
        assert ld3 == ld and np.array_equal(g4c, g4)
    finally:
        prob.close()


def test_auto_mode_uses_the_exact_gradient_above_the_side_by_side_range(monkeypatch):
    """n = 16 640 > 16 384: the default fit calls tgp_d_gp_solve_grad (one solve + one gradient per evaluation instead of
    ntheta + 1 solves) and ends at a likelihood no lower than where it started, at a stationary point of the exact gradient."""
    monkeypatch.delenv("TGP_ML_GRADIENT", raising=False)
    rng = np.random.default_rng(3)
    n = 16640
    X = rng.uniform(0, 1, (n, 2))
    y = np.sin(6 * X[:, 0]) * np.cos(4 * X[:, 1]) + 0.05 * rng.standard_normal(n)
    calls = []
    real = ops.gp_solve_grad_resident
    monkeypatch.setattr(ops, "gp_solve_grad_resident", lambda *a, **k: (calls.append(1), real(*a, **k))[1])
    gp = treegp.GPInterpolation(kernel="0.7**2 * AnisotropicRBF(invLam=array([[60., 0.], [0., 60.]]))", optimizer="log-likelihood",
                                normalize=True)
    gp.initialize(X, y, y_err=0.05 * np.ones(n))
    l0 = gp.return_log_likelihood()
    gp.solve()
    assert len(calls) >= 3
    assert gp._optimizer._logL > l0
    _, grad = gp._optimizer.log_likelihood_gradient(gp.kernel)
    assert np.abs(grad).max() < 1e-3 * abs(gp._optimizer._logL)

"""Randomised sweep over sizes, kernels, dimensions and error models through the public API, against the oracle: the seams S1-S3b
and the likelihood gradient on whatever partial-tile / partial-panel geometry the draw produces.  Fixed seed, ~30 s."""
import numpy as np
import pytest

import treegp_amd as treegp
from oracle import gp_oracle as O

pytestmark = pytest.mark.gpu


def _draw(rng):
    nd = int(rng.integers(1, 3))
    kind = str(rng.choice(["rbf", "arbf", "vk", "avk"]))
    amp = float(rng.uniform(0.3, 2.0))
    scale = float(rng.uniform(0.04, 0.4))
    if kind == "rbf":
        return nd, "%r**2 * RBF(%r)" % (amp, scale), dict(kind="gauss", amp=amp ** 2, a=scale ** -2, b=0.0, c=scale ** -2 if nd == 2 else 0.0)
    if kind == "vk":
        return nd, "%r**2 * VonKarman(length_scale=%r)" % (amp, scale), dict(kind="vk", amp=amp ** 2, ell=scale)
    if nd == 1:
        inv = np.array([[scale ** -2]])
    else:
        g = rng.uniform(-0.5, 0.5, (2, 2))
        inv = (np.eye(2) + g).dot((np.eye(2) + g).T) * scale ** -2
    cls = "AnisotropicRBF" if kind == "arbf" else "AnisotropicVonKarman"
    p = dict(kind="gauss" if kind == "arbf" else "avk", amp=amp ** 2, a=inv[0, 0], b=inv[0, 1] if nd == 2 else 0.0, c=inv[1, 1] if nd == 2 else 0.0)
    return nd, "%r**2 * %s(invLam=array(%s))" % (amp, cls, np.array2string(inv, separator=", ", precision=17).replace("\n", "")), p


@pytest.mark.parametrize("block", range(6))
def test_random_problems_against_the_oracle(block):
    rng = np.random.default_rng(20240 + block)
    for case in range(7):
        n = int(rng.choice([rng.integers(1, 40), rng.integers(40, 700), rng.integers(700, 2600)]))
        m = int(rng.integers(1, 900))
        nd, kern, p = _draw(rng)
        X = rng.uniform(0, 1, (n, nd))
        Xs = rng.uniform(0, 1, (m, nd))
        if case % 3 == 0 and n > 2:
            Xs[: min(3, m)] = X[: min(3, m)]                                   # coincident query points (lim0 branch of the von Karman kernels)
        y = np.sin(5 * X[:, 0]) + 0.1 * rng.standard_normal(n) + 0.7
        mode = case % 3
        y_err = None if mode == 0 else (0.1 * np.ones(n) if mode == 1 else 0.1 * rng.uniform(0.5, 1.5, n))
        wn = 0.05 if mode == 0 else 0.0                                         # no errors: white noise keeps K positive definite
        tag = "block %d case %d: n=%d m=%d %s" % (block, case, n, m, kern)
        gp = treegp.GPInterpolation(kernel=kern, optimizer="none", normalize=True, white_noise=wn)
        gp.initialize(X, y, y_err=y_err)
        mc = min(m, 96)
        yp, cov = gp.predict(Xs[:mc], return_cov=True)
        yall = gp.predict(Xs)
        err = np.sqrt((np.zeros(n) if y_err is None else y_err) ** 2 + wn ** 2)
        mean = np.mean(y)
        K = O.kernel_matrix(X=X, **p)
        alpha, logdet = O.gp_solve(K, y - mean, err)
        HT = O.kernel_matrix(X=Xs, Y=X, **p)
        ref = O.gp_predict(HT, alpha) + mean
        scale = max(np.abs(ref).max(), 1.0)
        np.testing.assert_allclose(yall, ref, rtol=0, atol=1e-9 * scale, err_msg=tag)
        np.testing.assert_allclose(yp, ref[:mc], rtol=0, atol=1e-9 * scale, err_msg=tag)
        cref = O.gp_predict_cov(K, err, HT[:mc], O.kernel_matrix(X=Xs[:mc], **p))
        np.testing.assert_allclose(cov, cref, rtol=0, atol=1e-8 * p["amp"], err_msg=tag)
        np.testing.assert_allclose(gp.return_log_likelihood(), O.log_likelihood(K, y - mean, err), rtol=1e-9, atol=1e-7, err_msg=tag)


@pytest.mark.parametrize("block", range(3))
def test_random_pair_binning_against_the_oracle(block):
    """S4 on random catalogues: clustered and uniform points, duplicates (bootstrap-like), weights or none, random bin grids;
    pair counts bit-exact, sums to 1e-11 of their scale."""
    from treegp_amd import ops
    rng = np.random.default_rng(777 + block)
    for case in range(6):
        n = int(rng.choice([rng.integers(2, 60), rng.integers(60, 1500), rng.integers(1500, 3500)]))
        if case % 2:
            x, y = rng.uniform(0, 1, n), rng.uniform(0, 1, n)
        else:                                                     # clumps: crowded bins, empty bins
            c = rng.uniform(0, 1, (5, 2))
            pick = rng.integers(0, 5, n)
            x, y = c[pick, 0] + 0.03 * rng.standard_normal(n), c[pick, 1] + 0.03 * rng.standard_normal(n)
        if case % 3 == 0 and n > 4:
            dup = rng.integers(0, n, n // 4)
            x[: len(dup)], y[: len(dup)] = x[dup], y[dup]         # coincident points: r == 0 pairs are not counted
        k = rng.standard_normal(n)
        w = None if case % 2 else rng.uniform(0.2, 3.0, n)
        nb = int(rng.integers(3, 24))
        max_sep = float(rng.uniform(0.05, 0.6))
        tag = "block %d case %d n=%d nbins=%d" % (block, case, n, nb)
        got = ops.kk_twod(x, y, k, w, 0.0, max_sep, nb)
        ref = O.kk_twod(x, y, k, w, 0.0, max_sep, nb)
        assert np.array_equal(got[2], ref[2]), tag
        for a, b in zip(got[:2], ref[:2]):
            np.testing.assert_allclose(a, b, rtol=0, atol=1e-11 * max(np.abs(b).max(), 1.0), err_msg=tag)
        min_sep = max_sep * float(rng.uniform(0.01, 0.2))
        got = ops.kk_log(x, y, k, w, min_sep, max_sep, nb)
        ref = O.kk_log(x, y, k, w, min_sep, max_sep, nb)
        assert np.array_equal(got[4], ref[4]), tag
        for a, b in zip(got[:4], ref[:4]):
            np.testing.assert_allclose(a, b, rtol=0, atol=1e-11 * max(np.abs(b).max(), 1.0), err_msg=tag)

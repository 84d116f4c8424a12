"""The reference's own CPU call sequence for the hot path, issued call for call.

TEST INFRASTRUCTURE / REPORTED BASELINE ONLY (see oracle/__init__.py).  Unlike
``gp_oracle.py`` (explicit formulas), this file makes the very SciPy/NumPy calls the
reference makes, in its order, so that timing it on the GPU box's host cores is what a
treegp user gets today:

  kernels.py:117-121   pdist(mahalanobis, VI) -> exp(-.5 d^2) -> squareform -> fill_diagonal
  sklearn Product      np.full(amp) * K                      (sklearn kernels.py:931-966)
  gp_interp.py:180     + np.eye(N) * y_err**2
  gp_interp.py:181-182 cholesky(overwrite_a=True, lower=False) ; cho_solve
  kernels.py:125-126   cdist(mahalanobis, VI) -> exp           (gp_interp.py:177)
  gp_interp.py:183     HT @ alpha
"""
import time

import numpy as np
from scipy.linalg import cho_solve, cholesky
from scipy.spatial.distance import cdist, pdist, squareform


def anisotropic_rbf_self(X, invLam, amp):
    dists = pdist(X, metric="mahalanobis", VI=invLam)
    K = np.exp(-0.5 * dists ** 2)
    K = squareform(K)
    np.fill_diagonal(K, 1)
    return np.full((len(X), len(X)), amp) * K


def anisotropic_rbf_cross(Xs, X, invLam, amp):
    dists = cdist(Xs, X, metric="mahalanobis", VI=invLam)
    K = np.exp(-0.5 * dists ** 2)
    return np.full(K.shape, amp) * K


def solve_predict(X, y, y_err, Xs, invLam, amp, timings=None):
    """alpha, y_pred exactly as GPInterpolation.return_gp_predict computes them
    (gp_interp.py:168-183), with per-phase wall times appended to ``timings``."""
    t = [time.perf_counter()]
    K = anisotropic_rbf_self(X, invLam, amp) + np.eye(len(y)) * y_err ** 2
    t.append(time.perf_counter())
    factor = (cholesky(K, overwrite_a=True, lower=False), False)
    t.append(time.perf_counter())
    alpha = cho_solve(factor, y, overwrite_b=False)
    t.append(time.perf_counter())
    HT = anisotropic_rbf_cross(Xs, X, invLam, amp)
    t.append(time.perf_counter())
    y_pred = np.dot(HT, alpha.reshape((len(alpha), 1))).T[0]
    t.append(time.perf_counter())
    if timings is not None:
        timings.update(kbuild=t[1] - t[0], cholesky=t[2] - t[1], cho_solve=t[3] - t[2], cross_kernel=t[4] - t[3],
                       matvec=t[5] - t[4], total=t[5] - t[0])
    return alpha, y_pred

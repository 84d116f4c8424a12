"""Maximum-likelihood hyper-parameter search; mirrors ``treegp/log_likelihood.py:7-62``.

Each likelihood evaluation is one fused GPU solve (tgp_gp_solve: K build + noise diagonal +
Cholesky + solve + logdet, K never leaves the device).  The L-BFGS-B driver stays on the host,
as in the reference (no analytic gradient is supplied there either, :57).
"""
import copy

import numpy as np
from scipy import optimize

from . import ops
from .kernels import kernel_to_spec


class log_likelihood(object):
    """:param X: coordinates (n_samples, 1 or 2)  :param y: values  :param y_err: errors."""

    def __init__(self, X, y, y_err):
        self.X, self.y, self.y_err = X, y, y_err
        self.ndata = len(X[:, 0])

    def log_likelihood(self, kernel):
        """-0.5 y.K^-1.y - (n/2) log 2 pi - 0.5 log det K; any failure (e.g. K not positive
        definite) gives -inf, as at log_likelihood.py:28-39."""
        try:
            _, log_det, chi2, _ = ops.gp_solve(kernel_to_spec(kernel), self.X, self.y, self.y_err, want_alpha=False)
            ll = -0.5 * chi2 - (0.5 * self.ndata) * np.log(2.0 * np.pi) - 0.5 * log_det
        except (np.linalg.LinAlgError, FloatingPointError, ValueError, ops._lib.TgpError):
            ll = -np.inf
        if not np.isfinite(ll):
            ll = -np.inf
        return ll

    def optimizer(self, kernel):
        """L-BFGS-B on -log L over theta (log_likelihood.py:43-62)."""
        template = kernel

        def cost(theta):
            return -self.log_likelihood(template.clone_with_theta(theta))

        best = optimize.minimize(cost, template.theta, method="L-BFGS-B")["x"]
        fitted = template.clone_with_theta(best)
        self._kernel = copy.deepcopy(fitted)
        self._logL = self.log_likelihood(self._kernel)
        return fitted

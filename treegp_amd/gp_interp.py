"""``GPInterpolation`` with treegp's constructor / initialize / solve / predict API, running the
kernel-matrix build, Cholesky solve and prediction on the GPU.

Mirrors ``treegp/gp_interp.py:15-291`` of the reference (plotting, :293-377, is out of scope).
State and cache semantics are the reference's: ``_alpha`` is computed on the first ``predict``
and invalidated by ``initialize`` (:227) and ``_fit`` (:119); ``predict`` works without
``solve``; ``normalize`` uses the mean of ``y - spatial_average`` taken before white noise is
folded in (:217-224); a mean-function table counts as present when X0 is not all zeros (:235).
"""
import copy

import numpy as np
from sklearn.neighbors import KNeighborsRegressor

from . import ops
from .fits_io import read_bintable_row
from .kernels import eval_kernel, kernel_to_spec


class GPInterpolation(object):
    """Gaussian-process interpolation of one scalar field over 1-D / 2-D coordinates.

    :param kernel:        string that ``eval_kernel`` turns into a scikit-learn kernel. [default 'RBF(1)']
    :param optimizer:     "none", "two-pcf", "anisotropic" or "log-likelihood".
    :param normalize:     subtract the mean of the data before interpolating. [default True]
    :param p0:            start point (size, g1, g2) of the anisotropic 2-pcf fit.
    :param white_noise:   extra uncorrelated noise added in quadrature to y_err. [default 0.]
    :param n_neighbors:   neighbours of the KNN interpolation of the mean function. [default 4]
    :param average_fits:  FITS table (meanify output) holding the mean function. [default None]
    :param indice_meanify: column of the mean function to use when it has several.
    :param nbins, min_sep, max_sep: binning of the 2-point correlation function.
    """

    def __init__(self, kernel="RBF(1)", optimizer="two-pcf", normalize=True, p0=[3000.0, 0.0, 0.0],
                 white_noise=0.0, n_neighbors=4, average_fits=None, indice_meanify=None, nbins=20,
                 min_sep=None, max_sep=None):
        self.normalize = normalize
        self.optimizer = optimizer
        self.white_noise = white_noise
        self.n_neighbors = n_neighbors
        self.nbins = nbins
        self.min_sep = min_sep
        self.max_sep = max_sep
        self.robust_fit = (self.optimizer == "anisotropic")
        self.p0_robust_fit = p0
        self.indice_meanify = indice_meanify

        if not isinstance(kernel, str):
            raise TypeError("kernel should be a string a list or a numpy.ndarray of string")
        self.kernel_template = eval_kernel(kernel)

        if self.optimizer not in ["anisotropic", "two-pcf", "log-likelihood", "none"]:
            raise ValueError("Only anisotropic, two-pcf, log-likelihood and none are supported for optimizer. "
                             "Current value: %s" % (self.optimizer))

        if average_fits is not None:
            # gp_interp.py:97-102 (fitsio.read(...)["COORDS0"][0]); fitsio is replaced by a
            # minimal BINTABLE reader so the path has no dependency the image lacks
            average = read_bintable_row(average_fits)
            X0 = average["COORDS0"]
            y0 = average["PARAMS0"]
        else:
            X0 = None
            y0 = None
        self._X0 = X0
        self._y0 = y0
        self._alpha = None
        self._factor = None

    # -- hyper-parameter fit ---------------------------------------------------------------------
    def _fit(self, kernel, X, y, y_err):
        """gp_interp.py:111-141."""
        from .two_pcf import two_pcf
        from .log_likelihood import log_likelihood
        self._drop_solution()
        if self.optimizer != "none":
            if self.optimizer in ["two-pcf", "anisotropic"]:
                self._optimizer = two_pcf(X, y, y_err, self.min_sep, self.max_sep, nbins=self.nbins,
                                          anisotropic=(self.optimizer == "anisotropic"),
                                          robust_fit=self.robust_fit, p0=self.p0_robust_fit)
                kernel = self._optimizer.optimizer(kernel)
            if self.optimizer == "log-likelihood":
                self._optimizer = log_likelihood(X, y, y_err)
                kernel = self._optimizer.optimizer(kernel)
        return kernel

    def _drop_solution(self):
        self._alpha = None
        if getattr(self, "_factor", None) is not None:
            self._factor.free()
        self._factor = None

    # -- prediction ------------------------------------------------------------------------------
    def predict(self, X, return_cov=False):
        """Interpolated values (and optionally the posterior covariance) at X (n_samples, 1 or 2).
        gp_interp.py:143-166."""
        y_init = copy.deepcopy(self._y)
        y_err = copy.deepcopy(self._y_err)
        y_interp, y_cov = self.return_gp_predict(y_init - self._mean - self._spatial_average, self._X, X,
                                                 self.kernel, y_err=y_err, return_cov=return_cov)
        y_interp = y_interp.T
        spatial_average = self._build_average_meanify(X)
        y_interp += self._mean + spatial_average
        if return_cov:
            return y_interp, y_cov
        return y_interp

    def return_gp_predict(self, y, X1, X2, kernel, y_err, return_cov=False):
        """gp_interp.py:168-194 on the GPU: fused K build + Cholesky + solve (tgp_gp_solve), fused
        cross-kernel mat-vec (tgp_gp_predict) and, for return_cov, Kss - HT K^-1 HT^T from the
        factor kept on the device (tgp_gp_predict_cov) instead of a second factorisation."""
        spec = kernel_to_spec(kernel)
        need_factor = return_cov and self._factor is None
        if self._alpha is None or need_factor:
            alpha, _, _, factor = ops.gp_solve(spec, X1, y, y_err, keep=return_cov)
            self._alpha = alpha
            if factor is not None:
                if self._factor is not None:
                    self._factor.free()
                self._factor = factor
        y_predict = ops.gp_predict(spec, X1, self._alpha, X2)
        if return_cov:
            y_cov = ops.gp_predict_cov(spec, self._factor, X1, X2)
            return y_predict, y_cov
        return y_predict, None

    # -- data ------------------------------------------------------------------------------------
    def initialize(self, X, y, y_err=None):
        """gp_interp.py:196-227."""
        self.kernel = copy.deepcopy(self.kernel_template)
        self._X = X
        self._y = y
        if y_err is None:
            y_err = np.zeros_like(y)
        self._y_err = y_err

        if self._X0 is None:
            self._X0 = np.zeros_like(self._X)
            self._y0 = np.zeros_like(self._y)
        self._spatial_average = self._build_average_meanify(X)

        if self.white_noise > 0:
            y_err = np.sqrt(copy.deepcopy(self._y_err) ** 2 + self.white_noise ** 2)
        self._y_err = y_err

        if self.normalize:
            self._mean = np.mean(y - self._spatial_average)
        else:
            self._mean = 0.0
        self._drop_solution()

    def _build_average_meanify(self, X):
        """Mean function at X by K-nearest-neighbour interpolation of the meanify table, zeros when
        there is none.  gp_interp.py:229-243."""
        if np.sum(np.equal(self._X0, 0)) != len(self._X0[:, 0]) * len(self._X0[0]):
            y0 = np.asarray(self._y0)
            k = self.n_neighbors
            on_gpu = k <= 8 or k == 16
            if y0.ndim == 1 and on_gpu:
                return ops.knn_mean(self._X0, y0, X, k)                          # tgp_knn_mean
            if y0.ndim == 2 and self.indice_meanify is not None and on_gpu:
                return ops.knn_mean(self._X0, y0[:, self.indice_meanify], X, k)
            # multi-column table without a column pick / unusual k: scikit-learn, as the reference does
            neigh = KNeighborsRegressor(n_neighbors=k)
            neigh.fit(self._X0, self._y0)
            average = neigh.predict(X)
            if self.indice_meanify is not None:
                average = average[:, self.indice_meanify]
            return average
        return np.zeros((len(X[:, 0])))

    def solve(self):
        """Fit the hyper-parameters if an optimizer was requested.  gp_interp.py:245-258."""
        self._init_theta = []
        kernel = copy.deepcopy(self.kernel)
        self._init_theta.append(kernel.theta)
        self.kernel = self._fit(self.kernel, self._X, self._y - self._mean - self._spatial_average, self._y_err)

    def return_2pcf(self):
        """xi, xi_weight, distance, coord, mask of the measured 2-point correlation function.
        gp_interp.py:260-275."""
        from .two_pcf import two_pcf
        pcf = two_pcf(self._X, self._y - self._mean - self._spatial_average, self._y_err, self.min_sep,
                      self.max_sep, nbins=self.nbins, anisotropic=(self.optimizer == "anisotropic"))
        return pcf.return_2pcf()

    def return_log_likelihood(self, theta=None):
        """Log-likelihood of the data for the current (or given) hyper-parameters.
        gp_interp.py:277-291."""
        from .log_likelihood import log_likelihood
        kernel = copy.deepcopy(self.kernel)
        if theta is not None:
            kernel = kernel.clone_with_theta(theta)
        logl = log_likelihood(self._X, self._y - self._mean - self._spatial_average, self._y_err)
        return logl.log_likelihood(kernel)

    def plot_fitted_kernel(self):
        raise NotImplementedError("plotting (treegp/gp_interp.py:293-377) is outside the GPU hot path")

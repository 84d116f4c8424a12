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
    :param backend:       not in the reference.  None: one GPU, unless the multi-GPU route is enabled
                          (``treegp_amd.dist.enable()`` or TGP_DIST=1 under ``torchrun``) and the problem is at least its
                          size threshold; "dist": always the multi-GPU route (row-block-cyclic Cholesky over the ranks of
                          torch.distributed, query points sharded; every rank makes the same calls with the same data and
                          gets the same results); "single": always this rank's GPU alone.
    """

    def __init__(self, kernel="RBF(1)", optimizer="two-pcf", normalize=True, p0=[3000.0, 0.0, 0.0],
                 white_noise=0.0, n_neighbors=4, average_fits=None, indice_meanify=None, nbins=20,
                 min_sep=None, max_sep=None, backend=None):
        if backend not in (None, "dist", "single"):
            raise ValueError("backend must be None, 'dist' or 'single'. Current value: %s" % (backend,))
        self.backend = backend
        self.normalize, self.optimizer, self.white_noise = normalize, optimizer, white_noise
        self.n_neighbors, self.indice_meanify = n_neighbors, indice_meanify
        self.nbins, self.min_sep, self.max_sep = nbins, min_sep, max_sep
        self.robust_fit = optimizer == "anisotropic"       # the 2-D fit of (size, g1, g2) goes with the TwoD pcf
        self.p0_robust_fit = p0

        if not isinstance(kernel, str):
            raise TypeError("kernel should be a string a list or a numpy.ndarray of string")
        self.kernel_template = eval_kernel(kernel)

        if optimizer not in ("anisotropic", "two-pcf", "log-likelihood", "none"):
            raise ValueError("Only anisotropic, two-pcf, log-likelihood and none are supported for optimizer. "
                             "Current value: %s" % (optimizer))

        # mean function: the meanify table (gp_interp.py:97-107, read there with fitsio) or none yet
        self._X0 = self._y0 = None
        if average_fits is not None:
            table = read_bintable_row(average_fits)
            self._X0, self._y0 = table["COORDS0"], table["PARAMS0"]
        self._alpha = None
        self._factor = self._factor_key = None

    def _scope(self):
        """the backend choice of this object, in force for the solves made inside (treegp_amd.dist.scope)"""
        if self.backend is None:
            import contextlib
            return contextlib.nullcontext()
        from . import dist
        return dist.scope(self.backend)

    # -- hyper-parameter fit ---------------------------------------------------------------------
    def _fit(self, kernel, X, y, y_err):
        """Run the requested optimiser on the (mean-subtracted) data and return the fitted kernel; the
        cached solution is dropped either way (gp_interp.py:111-141)."""
        from .two_pcf import two_pcf
        from .log_likelihood import log_likelihood
        self._drop_solution()
        if self.optimizer in ("two-pcf", "anisotropic"):
            self._optimizer = two_pcf(X, y, y_err, self.min_sep, self.max_sep, nbins=self.nbins,
                                      anisotropic=(self.optimizer == "anisotropic"),
                                      robust_fit=self.robust_fit, p0=self.p0_robust_fit)
        elif self.optimizer == "log-likelihood":
            self._optimizer = log_likelihood(X, y, y_err)
        else:
            return kernel
        return self._optimizer.optimizer(kernel)

    def _drop_solution(self):
        self._alpha = None
        self._set_factor(None, None)

    def __del__(self):
        # a dropped object hands its factor's memory to the context (the next fit of this size allocates nothing: at N = 65 536
        # a hipFree + hipMalloc of 17 GB between two fits otherwise); raw ops.Factor handles free their memory by default
        try:
            self._set_factor(None, None)
        except Exception:
            pass

    def _set_factor(self, factor, key):
        if getattr(self, "_factor", None) is not None:
            self._factor.free(keep_memory=True)        # a refit of this object solves the same size next: its memory is reused
        self._factor, self._factor_key = factor, key

    @staticmethod
    def _factor_fingerprint(spec, X1, y_err):
        import hashlib
        h = hashlib.blake2b(digest_size=16)
        h.update(np.asarray([spec.kind, spec.amp, spec.a, spec.b, spec.c, spec.ell], dtype=np.float64).tobytes())
        for arr in (X1, y_err):
            arr = np.ascontiguousarray(arr, dtype=np.float64)
            h.update(str(arr.shape).encode())
            h.update(arr.tobytes())
        return h.digest()

    # -- prediction ------------------------------------------------------------------------------
    def predict(self, X, return_cov=False):
        """Interpolated values (and optionally the posterior covariance) at X (n_samples, 1 or 2).
        gp_interp.py:143-166: the GP acts on y - mean - mean function; both are added back."""
        residual = self._y - self._mean - self._spatial_average
        with self._scope():
            y_star, y_cov = self.return_gp_predict(residual, self._X, X, self.kernel, y_err=self._y_err,
                                                   return_cov=return_cov)
        y_star = y_star + (self._mean + self._build_average_meanify(X))
        return (y_star, y_cov) if return_cov else y_star

    def return_gp_predict(self, y, X1, X2, kernel, y_err, return_cov=False):
        """gp_interp.py:168-194 on the GPU: fused K build + Cholesky + solve (tgp_gp_solve), fused
        cross-kernel mat-vec (tgp_gp_predict) and, for return_cov, Kss - HT K^-1 HT^T from the
        factor kept on the device (tgp_gp_predict_cov) instead of a second factorisation."""
        try:
            spec = kernel_to_spec(kernel)
        except NotImplementedError:
            # any other scikit-learn kernel tree (Sum, WhiteKernel, Matern, ...): the kernel object evaluates itself on
            # the host, exactly as in the reference, and the device factorises what it returns (tgp_gp_solve_dense)
            return self._return_gp_predict_dense(y, X1, X2, kernel, y_err, return_cov)
        # The reference caches only alpha (computed when it is None, whatever the arguments: gp_interp.py:179) and
        # rebuilds K + diag(y_err^2) from its ARGUMENTS for every covariance request (:186-187).  The factor kept on the
        # device therefore carries the fingerprint of what it was built from and is rebuilt when that differs.
        key = self._factor_fingerprint(spec, X1, y_err) if return_cov else None
        if self._alpha is None:
            self._alpha, _, _, factor = ops.gp_solve(spec, X1, y, y_err, keep=return_cov)
            self._set_factor(factor, key)
        elif return_cov and (self._factor is None or self._factor_key != key):
            factor = ops.gp_solve(spec, X1, y, y_err, keep=True)[3]          # alpha stays the cached one, as in the reference
            self._set_factor(factor, key)
        y_predict = ops.gp_predict(spec, X1, self._alpha, X2)
        if return_cov:
            y_cov = ops.gp_predict_cov(spec, self._factor, X1, X2)
            return y_predict, y_cov
        return y_predict, None

    def _return_gp_predict_dense(self, y, X1, X2, kernel, y_err, return_cov):
        """gp_interp.py:177-192 for a kernel only scikit-learn can evaluate: HT, K and k(X2) come from ``kernel.__call__``
        on the host; factorisation, solve and the posterior covariance run on the device."""
        HT = kernel(X2, Y=X1)
        key = None
        if return_cov:
            import hashlib
            h = hashlib.blake2b(digest_size=16)
            h.update(repr(kernel).encode())
            h.update(np.asarray(kernel.theta, dtype=np.float64).tobytes())
            for arr in (X1, y_err):
                arr = np.ascontiguousarray(arr, dtype=np.float64)
                h.update(str(arr.shape).encode())
                h.update(arr.tobytes())
            key = h.digest()
        need_alpha = self._alpha is None
        if need_alpha or (return_cov and (self._factor is None or self._factor_key != key)):
            alpha, _, _, factor = ops.gp_solve_dense(kernel(X1), y, y_err, keep=return_cov)
            if need_alpha:
                self._alpha = alpha
            self._set_factor(factor, key)
        y_predict = np.dot(HT, self._alpha.reshape((len(self._alpha), 1))).T[0]
        if return_cov:
            return y_predict, ops.gp_predict_cov_dense(self._factor, HT, kernel(X2))
        return y_predict, None

    def predict_fields(self, Y, X, y_err=None):
        """Several fields measured at the SAME positions with the same kernel and errors (one GP per PSF parameter, the Piff
        pattern of treegp/README.rst:28; with the reference each field is its own GPInterpolation, i.e. its own K build,
        cholesky and cho_solve).  Y: (n_fields, n) values at the positions given to ``initialize``; returns (n_fields, m)
        predictions at X.  One K build and one factorisation serve all fields; every field keeps its own mean
        (``normalize``) and shares the mean function table, the kernel and the errors of ``initialize``."""
        Y = np.atleast_2d(np.asarray(Y, dtype=np.float64))
        if Y.shape[1] != len(self._X):
            raise ValueError("Y must be (n_fields, %d)" % len(self._X))
        sigma = self._y_err if y_err is None else np.sqrt(np.asarray(y_err, dtype=np.float64) ** 2 + self.white_noise ** 2)
        means = np.mean(Y - self._spatial_average, axis=1) if self.normalize else np.zeros(len(Y))
        R = Y - means[:, None] - self._spatial_average[None, :]
        try:
            spec = kernel_to_spec(self.kernel)
        except NotImplementedError:
            spec = None
        with self._scope():
            if spec is not None:
                _, _, _, factor = ops.gp_solve(spec, self._X, R[0], sigma, keep=True, want_alpha=False)
            else:
                _, _, _, factor = ops.gp_solve_dense(self.kernel(self._X), R[0], sigma, keep=True, want_alpha=False)
            try:
                alphas = ops.factor_solve(factor, R)
            finally:
                factor.free(keep_memory=True)
            if spec is not None:
                pred = np.stack([ops.gp_predict(spec, self._X, a, X) for a in alphas])
            else:
                pred = alphas.dot(self.kernel(X, Y=self._X).T)
        return pred + means[:, None] + self._build_average_meanify(X)[None, :]

    # -- data ------------------------------------------------------------------------------------
    def initialize(self, X, y, y_err=None):
        """Take the data: coordinates (n, 1 or 2), values, errors (zeros when None).  gp_interp.py:196-227:
        a fresh copy of the kernel template, white noise added to the errors in quadrature, the mean taken
        of y minus the mean function, any cached solution dropped."""
        self.kernel = copy.deepcopy(self.kernel_template)
        self._X, self._y = X, y
        sigma = np.zeros_like(y) if y_err is None else y_err
        if self._X0 is None:
            # no mean-function table: an all-zero one, created once and kept across initialize() calls
            self._X0, self._y0 = np.zeros_like(X), np.zeros_like(y)
        self._spatial_average = self._build_average_meanify(X)
        if self.white_noise > 0:
            sigma = np.sqrt(sigma ** 2 + self.white_noise ** 2)
        self._y_err = sigma
        self._mean = np.mean(y - self._spatial_average) if self.normalize else 0.0
        self._drop_solution()

    def _build_average_meanify(self, X):
        """Mean function at X by K-nearest-neighbour interpolation of the meanify table, zeros when
        there is none.  gp_interp.py:229-243."""
        if np.count_nonzero(self._X0) > 0:               # "a table is present" = X0 is not all zeros
            y0 = np.asarray(self._y0)
            k = self.n_neighbors
            on_gpu = k <= 8 or k == 16
            if y0.ndim == 1 and on_gpu:
                return ops.knn_mean(self._X0, y0, X, k)                          # tgp_knn_mean
            if y0.ndim == 2 and self.indice_meanify is not None and on_gpu:
                return ops.knn_mean(self._X0, y0[:, self.indice_meanify], X, k)
            # multi-column table without a column pick / unusual k: scikit-learn, as the reference does
            table = KNeighborsRegressor(n_neighbors=k).fit(self._X0, self._y0)
            average = table.predict(X)
            return average if self.indice_meanify is None else average[:, self.indice_meanify]
        return np.zeros(len(X))

    def _residual(self):
        return self._y - self._mean - self._spatial_average

    def solve(self):
        """Fit the hyper-parameters if an optimizer was requested (gp_interp.py:245-258); the starting theta
        is kept in ``_init_theta``."""
        self._init_theta = [copy.deepcopy(self.kernel).theta]
        with self._scope():
            self.kernel = self._fit(self.kernel, self._X, self._residual(), self._y_err)

    def return_2pcf(self):
        """xi, xi_weight, distance, coord, mask of the measured 2-point correlation function
        (gp_interp.py:260-275)."""
        from .two_pcf import two_pcf
        measured = two_pcf(self._X, self._residual(), self._y_err, self.min_sep, self.max_sep, nbins=self.nbins,
                           anisotropic=(self.optimizer == "anisotropic"))
        return measured.return_2pcf()

    def return_log_likelihood(self, theta=None):
        """Log-likelihood of the data for the current (or the given) hyper-parameters (gp_interp.py:277-291)."""
        from .log_likelihood import log_likelihood
        kernel = copy.deepcopy(self.kernel)
        if theta is not None:
            kernel = kernel.clone_with_theta(theta)
        with self._scope():
            return log_likelihood(self._X, self._residual(), self._y_err).log_likelihood(kernel)

    def plot_fitted_kernel(self):
        raise NotImplementedError("plotting (treegp/gp_interp.py:293-377) is outside the GPU hot path")

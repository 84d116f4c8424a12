"""Hyper-parameter estimation from the binned 2-point correlation function.

Mirrors ``treegp/two_pcf.py`` of the reference: ``get_correlation_length_matrix`` (:12-31),
``robust_2dfit`` (:68-206) and ``two_pcf`` (:209-464).  The pair binning that the reference
delegates to TreeCorr (``kk.process``, :297-305 and :330-334) and the bootstrap loop around
it (:342-362) run on the GPU through ``tgp_kk_twod`` / ``tgp_kk_log`` /
``tgp_kk_twod_bootstrap`` (include/tgp.h, seam S4) with exact binning; all resamples of the
bootstrap are one batched launch.  The chi^2 minimisers evaluate the model at <= nbins^2
points per step and stay host-side drivers, as in the reference.
"""
from __future__ import print_function

import copy
import warnings

import numpy as np
import sklearn
from scipy import optimize

from . import kernels as _kernels
from . import ops
from .synthetic import correlation_length_matrix as get_correlation_length_matrix  # same formula as two_pcf.py:12-31

try:                                   # the reference requires iminuit (two_pcf.py:6); it is
    import iminuit                     # optional here because this image does not ship it
except ImportError:                    # pragma: no cover
    iminuit = None


def get_kernel_class(A):
    """AnisotropicRBF / AnisotropicVonKarman class of a kernel or of a Product holding one
    (two_pcf.py:34-65); anything else is a ValueError."""
    ok_classes = (_kernels.AnisotropicVonKarman, _kernels.AnisotropicRBF)
    msg = "Work only with treegp.kernels.AnisotropicVonKarman and treegp.kernels.AnisotropicRBF"
    if A.__class__ in ok_classes:
        return A.__class__
    if A.__class__ is sklearn.gaussian_process.kernels.Product:
        found = None
        for key in A.__dict__:
            if A.__dict__[key].__class__ in ok_classes:
                found = A.__dict__[key].__class__
        if found is None:
            raise ValueError(msg)
        return found
    raise ValueError(msg)



def _inv2x2(m):
    """Inverse of a 2x2 matrix by the adjugate (the reference calls numpy.linalg.inv here, two_pcf.py:109,139; LAPACK
    on a 2x2 costs ~80 us per call with a many-threaded BLAS, and the chi2 of the fit makes two per evaluation)."""
    a, b, c, d = m[0, 0], m[0, 1], m[1, 0], m[1, 1]
    det = a * d - b * c
    if det == 0.0:
        raise np.linalg.LinAlgError("Singular matrix")
    return np.array([[d, -b], [-c, a]]) / det


class robust_2dfit(object):
    """Fit (size, g1, g2) of an anisotropic kernel to the 2-D correlation function; amplitude and
    additive constant are linear and solved in closed form inside chi2 (two_pcf.py:68-206).

    :param kernel: kernel whose class fixes the radial profile  :param flat_data: flattened xi
    :param x, y: pixel coordinates  :param W: inverse covariance of xi  :param mask: pixels to use
    """

    _warned_no_minuit = False

    def __init__(self, kernel, flat_data, x, y, W, mask=None):
        self.kernel_class = get_kernel_class(kernel)
        self.mask = np.ones(len(x), dtype=bool) if mask is None else mask
        self.flat_data, self.x, self.y, self.W = flat_data, x, y, W
        self.coord = np.column_stack([x, y])
        self.N = int(np.sqrt(len(x)))

    def _model_skl(self, sigma, corr_length, g1, g2):
        """sigma^2 * kernel_class(invLam(corr_length, g1, g2)) sampled at the pixel centres; None
        outside |g| <= 1 (two_pcf.py:96-113)."""
        if max(abs(g1), abs(g2)) > 1:
            return None
        invLam = _inv2x2(get_correlation_length_matrix(corr_length, g1, g2))
        self.kernel_fit = sigma ** 2 * self.kernel_class(invLam=invLam)
        # the reference evaluates the kernel against a whole array of zeros and keeps column 0 (two_pcf.py:111-113);
        # one origin row gives the same column at 1/npix of the work
        return self.kernel_fit(self.coord, Y=np.zeros((1, self.coord.shape[1])))[:, 0]

    def chi2(self, param):
        """chi^2 over the non-linear parameters; the best amplitude (made positive) and constant
        for them are kept in self.alpha (two_pcf.py:115-148)."""
        bad = not np.isfinite(np.sum(param))
        model = None if bad else self._model_skl(1.0, param[0], param[1], param[2])
        if model is None:
            self.chi2_value = [np.inf]
            return np.inf
        m = model[self.mask]
        data = self.flat_data[self.mask]
        F = np.column_stack([m, np.ones_like(m)])
        FtW = F.T.dot(self.W)
        self.alpha = _inv2x2(FtW.dot(F)).dot(FtW.dot(data.reshape(-1, 1)))
        self.alpha[0] = abs(self.alpha[0])
        self.residuals = data - (self.alpha[0] * m + self.alpha[1])
        self.chi2_value = self.residuals.dot(self.W).dot(self.residuals.reshape(-1, 1))
        return self.chi2_value[0]

    def _minimize_minuit(self, p0=[3000.0, 0.2, 0.2]):
        """One minimisation from p0: MIGRAD when iminuit is installed (what the reference uses,
        two_pcf.py:150-176), otherwise Nelder-Mead polished by BFGS on the same 3-parameter chi2."""
        if iminuit is None and not robust_2dfit._warned_no_minuit:
            robust_2dfit._warned_no_minuit = True
            warnings.warn("iminuit is not installed: the anisotropic 2-pcf fit uses Nelder-Mead + BFGS on the same chi2 "
                          "instead of the reference's MIGRAD (treegp/two_pcf.py:150-176); fitted hyper-parameters can "
                          "differ from the reference's within the fit's own tolerance", RuntimeWarning, stacklevel=3)
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            if iminuit is None:
                nm = optimize.minimize(self.chi2, np.asarray(p0, float), method="Nelder-Mead",
                                       options=dict(xatol=1e-6, fatol=1e-8, maxiter=2000))
                pol = optimize.minimize(self.chi2, nm.x, method="BFGS")
                best = pol if (np.isfinite(pol.fun) and pol.fun <= nm.fun) else nm
                results = list(best.x)
                self._fit_ok = bool(nm.success and np.isfinite(best.fun))
            else:
                results, self._fit_ok = self._run_migrad(p0)
            self.chi2(results)              # leaves self.alpha at the solution
        self._minuit_result = results
        self.result = [np.sqrt(self.alpha[0][0])] + list(results[:3]) + [self.alpha[1][0]]

    def _run_migrad(self, p0):
        """MIGRAD through whichever iminuit API is installed: 2.x takes the start vector directly, 1.x goes
        through ``from_array_func`` (two_pcf.py:157-168).  The minimiser object stays in ``self.m``."""
        if int(iminuit.__version__.split(".")[0]) >= 2:
            m = iminuit.Minuit(self.chi2, p0)
            m.migrad()
            values, ok = [m.params[name].value for name in m.parameters], m.accurate
        else:
            m = iminuit.Minuit.from_array_func(self.chi2, p0, print_level=0)
            m.migrad()
            values, ok = [m.values[name] for name in m.values.keys()], m.migrad_ok()
        self.m = m
        return values, ok

    def minimize_minuit(self, p0=[3000.0, 0.2, 0.2]):
        """Minimise from p0; on failure retry from a 3x3x3 grid of start points until one
        converges (two_pcf.py:178-206)."""
        self._minimize_minuit(p0=p0)
        if not self._fit_ok:
            shear = np.linspace(-0.3, 0.3, 3)
            sizes = np.linspace(p0[0] - p0[0] / 10.0, 2 * p0[0], 3)
            G1, G2, S = (a.reshape(-1) for a in np.meshgrid(shear, shear, sizes))
            for start in zip(S, G1, G2):
                print("restart fit because failure")
                print(list(start))
                self._minimize_minuit(p0=list(start))
                if self._fit_ok:
                    break
        self._model_skl(*self.result[:4])


class two_pcf(object):
    """Measured 2-point correlation function, its bootstrap covariance and the kernel fit.

    :param X: coordinates (n, 1 or 2)   :param y: values   :param y_err: errors
    :param min_sep, max_sep, nbins: binning  :param anisotropic: 2-D (TwoD pixels) correlation
    :param robust_fit: fit (size, g1, g2) with ``robust_2dfit``  :param seed: bootstrap seed
    """

    def __init__(self, X, y, y_err, min_sep, max_sep, nbins=20, anisotropic=False, robust_fit=False,
                 p0=[3000.0, 0.0, 0.0], seed=610639139):
        ndim = np.shape(X)[1]
        if ndim not in (1, 2):
            raise ValueError("two-pcf support only 1d and 2d modeling for the moment. curent ndim: %i" % (ndim))
        self.ndim = ndim
        # 1-D positions get a zero second coordinate: the pair kernels are 2-D (two_pcf.py:248-251)
        self.X = X if ndim == 2 else np.column_stack([X[:, 0], np.zeros(len(X))])
        self.y, self.y_err = y, y_err
        self.min_sep, self.max_sep, self.nbins = min_sep, max_sep, nbins
        self.anisotropic, self.robust_fit, self.p0_robust_fit = anisotropic, robust_fit, p0
        self.seed = seed
        self._rng = None

    @property
    def rng(self):
        """PCG64 stream of the bootstrap, (re)started lazily from ``seed`` (two_pcf.py:260-267)."""
        gen = self._rng
        if gen is None:
            gen = self._rng = np.random.default_rng(self.seed)
        return gen

    def _bootstrap_index(self):
        # two_pcf.py:273-275: `integers(0, n-1)` has an exclusive upper end, so the last point is
        # never drawn -- kept, since it fixes the random stream and therefore the covariance
        npsfs = len(self.y)
        return self.rng.integers(0, npsfs - 1, size=npsfs)

    def resample_bootstrap(self):
        """One bootstrap resample u, v, y, y_err (two_pcf.py:269-281)."""
        ind = self._bootstrap_index()
        return self.X[:, 0][ind], self.X[:, 1][ind], self.y[ind], self.y_err[ind]

    def _twod_geometry(self):
        """mask (two_pcf.py:311-321) and pixel centres (:323-326) of the TwoD grid."""
        nb = self.nbins
        npixels = nb ** 2
        mask = np.ones((nb, nb), dtype=bool)
        nmask = int((nb / 2) + nb % 2)
        mask[nmask:, :] = False
        mask[nmask - 1][nmask:] = (nb % 2 == 0)
        bs = 2.0 * self.max_sep / nb
        bottom = np.linspace(-self.max_sep, self.max_sep, nb, endpoint=False)
        centre = (bottom + (bottom + bs)) / 2.0
        dy = np.repeat(centre, nb).reshape(nb, nb)       # varies along axis 0
        dx = dy.T
        distance = np.array([dx.reshape(npixels), dy.reshape(npixels)]).T
        return mask.reshape(npixels), distance

    def comp_2pcf(self, X, y, y_err):
        """xi, distance, Coord, mask for one catalogue (two_pcf.py:283-340)."""
        w = None if np.sum(y_err) == 0 else 1.0 / y_err ** 2
        k = y - np.mean(y)
        if self.anisotropic:
            xi, _, _ = ops.kk_twod(X[:, 0], X[:, 1], k, w, self.min_sep, self.max_sep, self.nbins)
            mask, distance = self._twod_geometry()
            Coord = distance
        else:
            xi, _, meanr, _, _ = ops.kk_log(X[:, 0], X[:, 1], k, w, self.min_sep, self.max_sep, self.nbins)
            distance = meanr
            mask = np.ones_like(xi, dtype=bool)
            Coord = np.array([distance, np.zeros_like(distance)]).T
        return xi, distance, Coord, mask

    def comp_xi_covariance(self, n_bootstrap=1000, mask=None, seed=610639139):
        """Bootstrap covariance of xi (two_pcf.py:342-362).  The random stream restarts at ``seed``; in
        the anisotropic case all n_bootstrap pair-binning passes are one batched GPU launch over an
        (n_bootstrap, n) index matrix."""
        self.seed, self._rng = seed, None
        if self.anisotropic:
            draws = np.stack([self._bootstrap_index() for _ in range(n_bootstrap)])
            samples = ops.kk_twod_bootstrap(self.X[:, 0], self.X[:, 1], self.y, self.y_err, draws, self.min_sep,
                                            self.max_sep, self.nbins)
        else:
            rows = []
            for _ in range(n_bootstrap):
                u, v, yb, eb = self.resample_bootstrap()
                rows.append(self.comp_2pcf(np.column_stack([u, v]), yb, eb)[0])
            samples = np.array(rows)
        if mask is not None:
            samples = samples[:, mask]
        centred = samples - samples.mean(axis=0)
        return 1.0 / (len(centred) - 1.0) * np.dot(centred.T, centred)

    def _n_bootstrap(self, npix):
        """Number of resamples that makes the de-biasing factor of the inverse covariance equal to 2
        (Taylor et al. 2012, eq. 35): root of (x-1)/(x-npix-2) = 2 found as the reference finds it,
        fsolve from npix + 10 and truncation (two_pcf.py:371-378; 444 for the 221 pixels of nbins=21)."""
        root = optimize.fsolve(lambda x: (x - 1.0) / (x - npix - 2.0) - 2.0, npix + 10)
        return int(root[0])

    def return_2pcf(self, seed=610639139):
        """xi, xi_weight, distance, coord, mask (two_pcf.py:364-391): weights are the de-biased inverse
        bootstrap covariance (anisotropic) or identity / var(y)."""
        xi, distance, coord, mask = self.comp_2pcf(self.X, self.y, self.y_err)
        if not self.anisotropic:
            return xi, np.eye(len(xi)) * 1.0 / np.var(self.y), distance, coord, mask
        npix = int(np.sum(mask))
        nboot = self._n_bootstrap(npix)
        cov = self.comp_xi_covariance(n_bootstrap=nboot, mask=mask, seed=seed)
        debias = (nboot - 1.0) / (nboot - npix - 2.0)
        return xi, np.linalg.inv(cov) * debias, distance, coord, mask

    def _default_separations(self):
        """min_sep / max_sep when not given (two_pcf.py:409-421): mean distance between neighbouring points
        (0 for the TwoD grid) and half the diagonal of the field."""
        span = np.ptp(self.X, axis=0)
        if self.ndim == 1:
            span[1] = 0.0
        lo, hi = self.min_sep, self.max_sep
        if lo is None:
            area = span[0] * span[1] if self.ndim == 2 else span[0]
            density = float(len(self.X)) / area
            lo = 0.0 if self.anisotropic else np.sqrt(1.0 / density)
        if hi is None:
            hi = np.sqrt(span[0] ** 2 + span[1] ** 2) / 2.0
        return lo, hi

    def optimizer(self, kernel):
        """Fit the kernel's hyper-parameters to the measured correlation function (two_pcf.py:393-464):
        chi^2 with the weights of ``return_2pcf``; either the robust 2-D fit, or the better of a simplex
        and an L-BFGS-B run from the kernel's current theta."""
        self.min_sep, self.max_sep = self._default_separations()
        xi, xi_weight, distance, coord, mask = self.return_2pcf()
        origin = np.zeros((1, coord.shape[1]))      # the reference passes zeros_like(coord) and keeps column 0: same values

        work = kernel.clone_with_theta(kernel.theta)     # one working copy: theta is set in place per evaluation, which is
                                                          # what clone_with_theta does after its (much slower) clone

        def model(theta):
            work.theta = theta
            return work(coord, Y=origin)[:, 0]

        def chi2(theta):
            r = xi[mask] - model(theta)[mask]
            return r.dot(xi_weight.dot(r))

        offset = 0
        if self.robust_fit:
            fit = robust_2dfit(kernel, xi, coord[:, 0], coord[:, 1], xi_weight, mask=mask)
            fit.minimize_minuit(p0=self.p0_robust_fit)
            self._results_robust = fit.result
            kernel, offset = copy.deepcopy(fit.kernel_fit), fit.result[-1]
        else:
            start = kernel.theta
            candidates = [optimize.fmin(chi2, start, disp=False), optimize.minimize(chi2, start, method="L-BFGS-B")["x"]]
            scores = [chi2(c) for c in candidates]
            kernel = kernel.clone_with_theta(candidates[scores.index(min(scores))])

        self._2pcf, self._2pcf_weight, self._2pcf_dist, self._2pcf_mask = xi, xi_weight, distance, mask
        self._2pcf_fit = model(kernel.theta) + offset
        self._kernel = copy.deepcopy(kernel)
        return kernel

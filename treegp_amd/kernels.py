"""Covariance functions with treegp's interface, evaluated on the GPU.

Mirrors ``treegp/kernels.py`` of the reference: ``eval_kernel`` (:17-59), ``AnisotropicRBF``
(:62-186), ``VonKarman`` (:189-301), ``AnisotropicVonKarman`` (:304-420).  The classes stay
scikit-learn ``Kernel`` objects (theta, bounds, clone_with_theta, ``amp * kernel`` products all
work as in the reference) but ``__call__`` goes to ``tgp_kernel_matrix`` (include/tgp.h, seam
S1) instead of scipy ``pdist/cdist`` + ``special.kv``.  There is no host evaluation path.

``kernel_to_spec`` turns a kernel object tree into the plain numbers the device kernels take.
"""
import numpy as np
from sklearn.gaussian_process.kernels import (ConstantKernel, Hyperparameter, Kernel, NormalizedKernelMixin,
                                              Product, RBF, StationaryKernelMixin)

from . import _lib
from . import ops


def _all_kernel_classes(cls=Kernel):
    found = []
    for sub in cls.__subclasses__():
        found.append(sub)
        found.extend(_all_kernel_classes(sub))
    return found


def eval_kernel(kernel):
    """Build a kernel object from its repr-like string, e.g. ``"0.5**2 * AnisotropicRBF(invLam=array(...))"``.

    Same contract as the reference (kernels.py:17-59): every scikit-learn ``Kernel`` subclass
    (which includes the classes of this module) and ``array`` are visible to the expression;
    a failing expression raises RuntimeError, an un-instantiated class raises TypeError.
    """
    scope = {c.__name__: c for c in _all_kernel_classes()}
    scope["array"] = np.array
    try:
        built = eval(kernel, scope)
    except Exception as exc:
        msg = "Failed to evaluate kernel string {0!r}.  Original exception: {1}".format(kernel, exc)
        raise RuntimeError(msg)
    if type(built.theta) is property:        # the expression named a class without instantiating it
        raise TypeError("String provided was not initialized properly")
    return built


class _CholeskyParametrised(StationaryKernelMixin, NormalizedKernelMixin, Kernel):
    """Shared parametrisation of the two invLam kernels (kernels.py:95-112,154-186 and
    :336-353,388-420): invLam = L L^T, theta = [log diag(L), strictly-lower L row-wise]."""

    _tgp_kind = None

    def __init__(self, invLam=None, scale_length=None, bounds=(-5, 5)):
        if scale_length is not None and invLam is not None:
            raise TypeError("Cannot set both invLam and scale_length in %s." % type(self).__name__)
        if scale_length is not None:                     # axis-aligned: invLam = diag(1 / l^2)
            invLam = np.diag(np.reciprocal(np.square(np.array(scale_length, dtype=float))))
        n = invLam.shape[0]
        self.ndim, self.ntheta = n, n * (n + 1) // 2
        self._d, self._t = np.diag_indices(n), np.tril_indices(n, -1)      # where theta lands in L
        self.set_params(invLam)
        limits = np.array(bounds)
        if limits.ndim == 1:                             # one (min, max) for every element of theta
            limits = np.tile(limits, (self.ntheta, 1))
        assert limits.shape == (self.ntheta, 2)
        self._bounds = limits

    # -- scikit-learn plumbing -------------------------------------------------------------
    @property
    def hyperparameter_cholesky_factor(self):
        return Hyperparameter("CholeskyFactor", "numeric", (1e-5, 1e5), int(self.ntheta))

    def get_params(self, deep=True):
        # only invLam: a clone therefore comes back with the default bounds (reference quirk)
        return dict(invLam=self.invLam)

    def set_params(self, invLam=None):
        """invLam -> L (lower Cholesky factor) -> theta = [log diag(L), strictly-lower L]."""
        if invLam is None:
            return
        chol = np.linalg.cholesky(invLam)
        self.invLam, self._L = invLam, chol
        self._theta = np.concatenate([np.log(chol[self._d]), chol[self._t]])

    @property
    def theta(self):
        return self._theta

    @theta.setter
    def theta(self, theta):
        """theta -> L -> invLam = L L^T (always positive definite)."""
        chol = np.zeros_like(self.invLam)
        chol[self._d] = np.exp(theta[:self.ndim])
        chol[self._t] = theta[self.ndim:]
        self._theta, self._L = theta, chol
        self.invLam = chol.dot(chol.T)

    @property
    def bounds(self):
        return self._bounds

    def __repr__(self):
        return "%s(invLam=%r)" % (type(self).__name__, self.invLam)

    # -- evaluation ------------------------------------------------------------------------
    def _spec(self):
        return _invlam_spec(self._tgp_kind, self.invLam, 1.0)

    def __call__(self, X, Y=None, eval_gradient=False):
        X = np.atleast_2d(X)
        if eval_gradient and Y is not None:
            raise ValueError("Gradient can only be evaluated when Y is None.")
        if X.shape[1] != self.ndim:
            raise ValueError("X has %d columns, kernel has ndim=%d" % (X.shape[1], self.ndim))
        K = ops.kernel_matrix(self._spec(), X, None if Y is None else np.atleast_2d(Y))
        if not eval_gradient:
            return K
        return K, self._gradient(X, K)

    def invLam_gradient(self):
        """d invLam / d theta_k, shape (ntheta, ndim, ndim): dL L^T + L dL^T with dL the single-element matrices of
        ``treegp/kernels.py:138-145`` (L_kk for the log-diagonal thetas, 1 for the strictly-lower ones)."""
        dL = np.zeros((self.ntheta, self.ndim, self.ndim))
        dL[(np.arange(self.ndim),) + self._d] = self._L[self._d]
        dL[(np.arange(self.ndim, self.ntheta),) + self._t] = 1.0
        half = np.dot(dL, self._L.T)
        return half + np.transpose(half, (0, 2, 1))

    def _gradient(self, X, K):
        raise ValueError("Gradient can not be evaluated.")          # AnisotropicVonKarman, kernels.py:383-384


class AnisotropicRBF(_CholeskyParametrised):
    """exp(-0.5 (x-x')^T invLam (x-x'))  -- treegp/kernels.py:62-186.

    :param invLam:        inverse covariance matrix (ndim x ndim, ndim in {1, 2} on the GPU).
    :param scale_length:  axis-aligned scale lengths instead of invLam (exactly one of the two).
    :param bounds:        bounds on theta, (2,) or (ntheta, 2).
    """
    _tgp_kind = _lib.TGP_ARBF

    def _gradient(self, X, K):
        """dK/dtheta_k = -1/2 K dX^T (d invLam/d theta_k) dX (kernels.py:128-150), from the device's K; host arithmetic on
        (n, n) arrays, one per theta -- nothing on the fit's path asks for it (log_likelihood.py:57 passes no jac; the
        likelihood gradient has its own device entry point, ops.gp_loglik_grad)."""
        if self.hyperparameter_cholesky_factor.fixed:            # kernels.py:128-130: no free parameter, an (n, n, 0) gradient
            return np.empty((X.shape[0], X.shape[0], 0))
        d = [X[:, None, a] - X[None, :, a] for a in range(self.ndim)]
        out = np.empty(K.shape + (self.ntheta,))
        for k, g in enumerate(self.invLam_gradient()):
            q = np.zeros_like(K)
            for a in range(self.ndim):
                for b in range(self.ndim):
                    if g[a, b] != 0.0:
                        q += g[a, b] * d[a] * d[b]
            out[:, :, k] = -0.5 * K * q
        return out


class AnisotropicVonKarman(_CholeskyParametrised):
    """d^(5/6) K_{5/6}(2 pi d) / lim0 with the Mahalanobis distance d -- treegp/kernels.py:304-420."""
    _tgp_kind = _lib.TGP_AVK


class VonKarman(StationaryKernelMixin, NormalizedKernelMixin, Kernel):
    """(d/l)^(5/6) K_{5/6}(2 pi d/l) / lim0, Euclidean d -- treegp/kernels.py:189-301.

    :param length_scale:         scalar length scale (theta = log length_scale).
    :param length_scale_bounds:  bounds on length_scale.
    """

    def __init__(self, length_scale=1.0, length_scale_bounds=(1e-5, 1e5)):
        self.length_scale, self.length_scale_bounds = length_scale, length_scale_bounds

    @property
    def anisotropic(self):
        scale = self.length_scale
        return bool(np.iterable(scale) and len(scale) > 1)

    @property
    def hyperparameter_length_scale(self):
        count = len(self.length_scale) if self.anisotropic else 1
        return Hyperparameter("length_scale", "numeric", self.length_scale_bounds, count)

    def _spec(self):
        if self.anisotropic:
            raise NotImplementedError("per-axis length scales are not defined for VonKarman distances")
        return ops.KernelSpec(_lib.TGP_VK, amp=1.0, ell=float(np.ravel(self.length_scale)[0]))

    def __call__(self, X, Y=None, eval_gradient=False):
        X = np.atleast_2d(X)
        if eval_gradient and Y is not None:
            raise ValueError("Gradient can only be evaluated when Y is None.")
        K = ops.kernel_matrix(self._spec(), X, None if Y is None else np.atleast_2d(Y))
        if not eval_gradient:
            return K
        # what the reference returns (kernels.py:278-288): K times the Euclidean distances -- not dK/d log l, reproduced as is
        if self.hyperparameter_length_scale.fixed:
            return K, np.empty((X.shape[0], X.shape[0], 0))
        diff = X[:, None, :] - X[None, :, :]
        return K, (K * np.sqrt(np.sum(diff * diff, axis=-1)))[:, :, np.newaxis]

    def __repr__(self):
        name = type(self).__name__
        if self.anisotropic:
            return "%s(length_scale=[%s])" % (name, ", ".join("%.3g" % v for v in self.length_scale))
        return "%s(length_scale=%.3g)" % (name, np.ravel(self.length_scale)[0])


# ---- kernel object -> device description ---------------------------------------------------
def _invlam_spec(kind, invLam, amp):
    invLam = np.asarray(invLam, dtype=np.float64)
    nd = invLam.shape[0]
    if nd == 1:
        return ops.KernelSpec(kind, amp=amp, a=invLam[0, 0], b=0.0, c=0.0)
    if nd == 2:
        return ops.KernelSpec(kind, amp=amp, a=invLam[0, 0], b=0.5 * (invLam[0, 1] + invLam[1, 0]), c=invLam[1, 1])
    raise NotImplementedError("the GPU kernels cover 1-D and 2-D coordinates (treegp's own scope), got ndim=%d" % nd)


def kernel_to_spec(kernel):
    """{kind, amp, a, b, c, ell} for the supported kernel trees:
    [ConstantKernel *]* one of RBF / AnisotropicRBF / VonKarman / AnisotropicVonKarman
    (what every reference test and doc string builds, e.g. tests/test_gp_interp.py:31,123).
    Anything else raises NotImplementedError -- there is no silent host fallback."""
    amp = 1.0
    node = kernel
    while isinstance(node, Product):
        if isinstance(node.k1, ConstantKernel):
            amp *= float(node.k1.constant_value)
            node = node.k2
        elif isinstance(node.k2, ConstantKernel):
            amp *= float(node.k2.constant_value)
            node = node.k1
        else:
            raise NotImplementedError("only ConstantKernel * <stationary kernel> products run on the GPU, got %r" % (kernel,))
    if isinstance(node, AnisotropicRBF):
        return _invlam_spec(_lib.TGP_ARBF, node.invLam, amp)
    if isinstance(node, AnisotropicVonKarman):
        return _invlam_spec(_lib.TGP_AVK, node.invLam, amp)
    if isinstance(node, VonKarman):
        s = node._spec()
        s.amp = amp
        return s
    if type(node) is RBF:                       # exactly RBF: scikit-learn's Matern derives from it
        ls = np.ravel(np.asarray(node.length_scale, dtype=np.float64))
        if ls.size == 1:
            inv = 1.0 / ls[0] ** 2
            return ops.KernelSpec(_lib.TGP_RBF, amp=amp, a=inv, b=0.0, c=inv)
        if ls.size == 2:
            return ops.KernelSpec(_lib.TGP_RBF, amp=amp, a=1.0 / ls[0] ** 2, b=0.0, c=1.0 / ls[1] ** 2)
        raise NotImplementedError("RBF with %d length scales" % ls.size)
    raise NotImplementedError("kernel %r is not supported by the GPU hot path" % (kernel,))


def spec_jacobian(kernel):
    """d(log amp, a, b, c) / d theta, shape (ntheta, 4), for the Gaussian kernel trees ``kernel_to_spec`` accepts
    (theta in scikit-learn's order: a Product's k1 first).  ``ops.gp_loglik_grad`` returns d logL / d(log amp, a, b, c);
    its product with this matrix is d logL / d theta with the kernel derivative of the reference:
    dK/dtheta_k = -1/2 K dX^T (dInvLam/dtheta_k) dX with dInvLam = dL L^T + L dL^T, dL the single-element matrices of
    ``treegp/kernels.py:138-145`` (AnisotropicRBF), and scikit-learn's own for ConstantKernel (K) and RBF."""
    if isinstance(kernel, Product):
        return np.vstack([spec_jacobian(kernel.k1), spec_jacobian(kernel.k2)])
    if isinstance(kernel, ConstantKernel):
        if kernel.hyperparameter_constant_value.fixed:
            return np.zeros((0, 4))
        return np.array([[1.0, 0.0, 0.0, 0.0]])
    if isinstance(kernel, AnisotropicRBF):
        if kernel.hyperparameter_cholesky_factor.fixed:         # theta is empty then (kernels.py:128-130)
            return np.zeros((0, 4))
        nd = kernel.ndim
        rows = []
        for g in kernel.invLam_gradient():
            rows.append([0.0, g[0, 0], g[0, 1] if nd == 2 else 0.0, g[1, 1] if nd == 2 else 0.0])
        return np.array(rows).reshape(kernel.ntheta, 4)
    if type(kernel) is RBF:
        if kernel.hyperparameter_length_scale.fixed:
            return np.zeros((0, 4))
        ls = np.ravel(np.asarray(kernel.length_scale, dtype=np.float64))
        if ls.size == 1:                                    # a = c = l^-2, theta = log l
            return np.array([[0.0, -2.0 / ls[0] ** 2, 0.0, -2.0 / ls[0] ** 2]])
        if ls.size == 2:
            return np.array([[0.0, -2.0 / ls[0] ** 2, 0.0, 0.0], [0.0, 0.0, 0.0, -2.0 / ls[1] ** 2]])
    raise NotImplementedError("no analytic derivative for %r (the reference has one for AnisotropicRBF only: "
                              "treegp/kernels.py:128-150)" % (kernel,))

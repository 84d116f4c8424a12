"""Maximum-likelihood hyper-parameter search; mirrors ``treegp/log_likelihood.py:7-62``.

Each likelihood evaluation is one fused GPU solve (tgp_gp_solve: K build + noise diagonal +
Cholesky + solve + logdet, K never leaves the device).  The L-BFGS-B driver stays on the host,
as in the reference, and so does its gradient: forward differences with SciPy's own step (the
reference passes no ``jac``, :57, so SciPy differentiates numerically).  The difference is that the
ntheta + 1 evaluations one gradient needs are independent solves of a small problem, which leave
most of the GPU idle one at a time: they are issued together, one context / stream per evaluation
(``parallel_fd``; 2-2.5x on a fit at N = 512 ... 8192).  Same optimiser, same formula, same iterates.
During a fit the data stay on the device (``ops.ResidentProblem``): an evaluation sends the kernel parameters and gets
two scalars back; the quadratic form is |L^-1 y|^2, with y carried through the factorisation as an extra matrix row
(no triangular sweep; the forward sweep alone when n is a multiple of 256).
"""
import copy
import os
import threading
from concurrent.futures import ThreadPoolExecutor

import numpy as np
from scipy import optimize

from . import _lib, ops
from ._lib import TgpError
from .kernels import kernel_to_spec, spec_jacobian

_FD_STEP = 1e-8                  # scipy.optimize.minimize(method="L-BFGS-B") default `eps` (absolute forward step)
_PARALLEL_MAX_N = 16384          # beyond this one solve fills the GPU by itself
_pool_lock = threading.Lock()
_ctx_pool = []                   # extra contexts (one stream each) for concurrent evaluations, created on demand


def _contexts(count):
    with _pool_lock:
        while len(_ctx_pool) < count:
            ctx = _lib.new_ctx(int(os.environ.get("TGP_DEVICE", os.environ.get("LOCAL_RANK", "0"))))
            _lib.load_library().tgp_set_lookahead(ctx, 0)     # side by side: one stream each (hardware queues are few)
            _ctx_pool.append(ctx)
        return _ctx_pool[:count]


def _rejects_theta(ex):
    """What a library error inside ONE likelihood evaluation means for the fit.  The reference catches everything
    (``except BaseException``, log_likelihood.py:38) so that any failure is a rejected theta and L-BFGS-B carries on.  Here a
    run-time device error (rc -2: out of device memory, a failed HIP call) is treated the same way, with a RuntimeWarning so
    that it cannot pass unnoticed; an argument error (rc -1) is a bug in the caller and always propagates.  TGP_ML_STRICT=1:
    every TgpError propagates (departures)."""
    if os.environ.get("TGP_ML_STRICT") == "1" or getattr(ex, "rc", -1) != -2:
        return False
    import warnings
    warnings.warn("likelihood evaluation failed on the device and counts as -inf (set TGP_ML_STRICT=1 to raise): %s" % ex,
                  RuntimeWarning, stacklevel=3)
    return True


class log_likelihood(object):
    """:param X: coordinates (n_samples, 1 or 2)  :param y: values  :param y_err: errors."""

    def __init__(self, X, y, y_err):
        self.X, self.y, self.y_err = X, y, y_err
        self.ndata = len(X[:, 0])
        # the multi-GPU route (treegp_amd.dist) takes every evaluation by itself: all ranks factorise one K together
        self.distributed = ops._dist_engine(self.ndata, None) is not None
        self.parallel_fd = (os.environ.get("TGP_ML_PARALLEL", "1") != "0" and self.ndata <= _PARALLEL_MAX_N
                            and not self.distributed)
        # "fd": the reference's fit (SciPy differentiates numerically, log_likelihood.py:57).  "analytic": L-BFGS-B is given
        # the exact gradient instead (log_likelihood_gradient; Gaussian kernels only) -- a different path through theta-space
        # to the same optimum.  "auto" (default): finite differences, the reference's iterates, wherever their ntheta + 1
        # evaluations run side by side (n <= 16 384); above that, where each of them is a full solve of its own, the exact
        # gradient (1.7 - 2 x a solve instead of ntheta more solves) if the kernel has one.
        self.gradient = os.environ.get("TGP_ML_GRADIENT", "auto")

    def log_likelihood(self, kernel, ctx=None, resident=None):
        """-0.5 y.K^-1.y - (n/2) log 2 pi - 0.5 log det K; any failure (e.g. K not positive
        definite) gives -inf, as at log_likelihood.py:28-39.  `resident`: the data already on the device
        (ops.ResidentProblem, used by optimizer() for the evaluations of one fit)."""
        try:
            try:
                spec = kernel_to_spec(kernel)
            except NotImplementedError:
                spec = None                       # any other scikit-learn kernel tree: it evaluates itself on the host
            if spec is None:
                _, log_det, chi2, _ = ops.gp_solve_dense(kernel(self.X), self.y, self.y_err, want_alpha=False, ctx=ctx)
            elif resident is not None:
                log_det, chi2 = ops.gp_solve_resident(spec, resident, ctx=ctx)
            else:
                _, log_det, chi2, _ = ops.gp_solve(spec, self.X, self.y, self.y_err, want_alpha=False, ctx=ctx)
            ll = -0.5 * chi2 - (0.5 * self.ndata) * np.log(2.0 * np.pi) - 0.5 * log_det
        except (np.linalg.LinAlgError, FloatingPointError, ValueError):
            # the mathematical failures the reference turns into -inf (log_likelihood.py:38-39)
            ll = -np.inf
        except TgpError as ex:
            if not _rejects_theta(ex):
                raise
            ll = -np.inf
        if not np.isfinite(ll):
            ll = -np.inf
        return ll

    def log_likelihood_gradient(self, kernel, ctx=None, resident=None):
        """(log L, d log L / d theta): 1/2 tr((alpha alpha^T - K^-1) dK/dtheta_k) with the kernel derivative of
        ``treegp/kernels.py:128-150``; K^-1 is formed on the device from the factor of the same solve
        (``ops.gp_loglik_grad``).  Failures give (-inf, zeros) as ``log_likelihood`` gives -inf."""
        ntheta = len(kernel.theta)
        try:
            spec = kernel_to_spec(kernel)
            jac = spec_jacobian(kernel)
            if resident is not None:
                log_det, chi2, g4 = ops.gp_solve_grad_resident(spec, resident, ctx=ctx)
            else:
                alpha, log_det, chi2, factor = ops.gp_solve(spec, self.X, self.y, self.y_err, keep=True, ctx=ctx)
                try:
                    g4 = ops.gp_loglik_grad(spec, factor, self.X, alpha, ctx=ctx)
                finally:
                    factor.free(keep_memory=True)          # the next evaluation solves the same size on this context
            ll = -0.5 * chi2 - (0.5 * self.ndata) * np.log(2.0 * np.pi) - 0.5 * log_det
        except (np.linalg.LinAlgError, FloatingPointError, ValueError):
            return -np.inf, np.zeros(ntheta)
        except TgpError as ex:
            if not _rejects_theta(ex):
                raise
            return -np.inf, np.zeros(ntheta)
        if not np.isfinite(ll):
            return -np.inf, np.zeros(ntheta)
        return ll, jac.dot(g4)

    def optimizer(self, kernel):
        """L-BFGS-B on -log L over theta (log_likelihood.py:43-62)."""
        template = kernel
        # X, y, y_err do not change during the fit: they go to the device once, every evaluation sends only theta
        resident = None
        if os.environ.get("TGP_ML_RESIDENT", "1") != "0" and not self.distributed:
            resident = ops.ResidentProblem(self.X, self.y, self.y_err)

        def cost(theta, ctx=None, work=None):
            # `work`: a kernel object owned by this evaluation slot whose theta is set in place -- what clone_with_theta
            # does after its (much slower) clone; the host side of an evaluation is otherwise mostly that clone
            if work is None:
                work = template.clone_with_theta(theta)
            else:
                work.theta = theta
            return -self.log_likelihood(work, ctx=ctx, resident=resident)

        try:
            best = self._minimise(cost, template, resident)
        finally:
            if resident is not None:
                resident.close()
        fitted = template.clone_with_theta(best)
        self._kernel = copy.deepcopy(fitted)
        self._logL = self.log_likelihood(self._kernel)
        return fitted

    def _use_exact_gradient(self, template):
        if self.gradient not in ("analytic", "auto"):
            return False
        try:
            spec_jacobian(template)
            gaussian = kernel_to_spec(template).kind in (_lib.TGP_RBF, _lib.TGP_ARBF)
        except NotImplementedError:
            gaussian = False
        if self.gradient == "analytic":
            if not gaussian:
                raise NotImplementedError("TGP_ML_GRADIENT=analytic: %r has no analytic derivative (Gaussian kernels only, as in "
                                          "the reference: treegp/kernels.py:128-150)" % (template,))
            return True
        # (on the multi-GPU route K^-1 would be formed by every rank on its replicated factor: not a saving there)
        return gaussian and self.ndata > _PARALLEL_MAX_N and not self.distributed and self._gradient_fits()

    def _gradient_fits(self):
        """The device gradient forms K^-1: two Np x Np buffers beside the factor, and its launches index row tiles of a
        matrix of at most 65535 rows (csrc/cov.hip: cov_plan).  Beyond either limit the fit keeps SciPy's finite differences."""
        lib = _lib.load_library()
        Np = int(lib.tgp_padded_n(self.ndata))
        if Np > 65535:
            return False
        need = 8.0 * (2.0 * Np * Np + 2.0 * float(lib.tgp_panel_elems(Np)))
        import ctypes
        free, total = ctypes.c_int64(), ctypes.c_int64()
        try:
            ctx = _lib.get_ctx()
        except RuntimeError:              # no device: nothing will run anyway, the choice is made on the size alone
            return True
        if lib.tgp_mem_info(ctx, ctypes.byref(free), ctypes.byref(total)) != 0:
            return True
        return need < 0.9 * free.value

    def _minimise(self, cost, template, resident=None):
        if self._use_exact_gradient(template):
            work = template.clone_with_theta(template.theta)

            state = {"exact": True}

            def cost_and_gradient(theta):
                work.theta = theta
                if state["exact"]:
                    try:
                        ll, grad = self.log_likelihood_gradient(work, resident=resident)
                        return -ll, -grad
                    except ops._lib.TgpError as err:
                        if self.gradient == "analytic":
                            raise                         # asked for explicitly: fail loudly
                        import warnings
                        warnings.warn("exact likelihood gradient unavailable (%s); continuing with finite differences" % err)
                        state["exact"] = False
                # SciPy's forward differences (the reference's scheme), one evaluation after the other
                f0 = -self.log_likelihood(work, resident=resident)
                grad = np.zeros(len(theta))
                for i in range(len(theta)):
                    shifted = np.array(theta, dtype=float)
                    shifted[i] += _FD_STEP
                    work.theta = shifted
                    grad[i] = (-self.log_likelihood(work, resident=resident) - f0) / (shifted[i] - theta[i])
                return f0, grad

            best = optimize.minimize(cost_and_gradient, template.theta, jac=True, method="L-BFGS-B")["x"]
        elif self.parallel_fd:
            ntheta = len(template.theta)
            ctxs = _contexts(ntheta + 1)
            works = [template.clone_with_theta(template.theta) for _ in range(ntheta + 1)]
            # evaluations in flight at once: all ntheta + 1 by default; TGP_ML_MAX_CONCURRENT caps it (the rest queue up)
            pool = ThreadPoolExecutor(max_workers=min(ntheta + 1, int(os.environ.get("TGP_ML_MAX_CONCURRENT", ntheta + 1))))

            def cost_and_gradient(theta):
                # SciPy's 2-point scheme for L-BFGS-B (approx_derivative, abs_step = eps): x_i + h, df / actual dx
                points = [np.array(theta, dtype=float)]
                for i in range(ntheta):
                    shifted = points[0].copy()
                    shifted[i] = points[0][i] + _FD_STEP
                    points.append(shifted)
                values = list(pool.map(cost, points, ctxs, works))
                grad = np.array([(values[i + 1] - values[0]) / (points[i + 1][i] - points[0][i]) for i in range(ntheta)])
                return values[0], grad

            try:
                best = optimize.minimize(cost_and_gradient, template.theta, jac=True, method="L-BFGS-B")["x"]
            finally:
                pool.shutdown()
        else:
            work = template.clone_with_theta(template.theta)
            best = optimize.minimize(lambda theta: cost(theta, None, work), template.theta, method="L-BFGS-B")["x"]
        return best

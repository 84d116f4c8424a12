"""NumPy-facing wrappers of the C-ABI seams (include/tgp.h).  Every function runs on the GPU
through libtgp.so; nothing here computes on the host.

A kernel is described by ``KernelSpec`` -- the plain numbers the device needs -- which
``treegp_amd.kernels.kernel_to_spec`` derives from a scikit-learn kernel object.
"""
import ctypes as C
import os
import threading

import numpy as np

from . import _lib
from ._lib import TgpKernel, check, f64, ptr, as_xy


class KernelSpec(object):
    __slots__ = ("kind", "amp", "a", "b", "c", "ell")

    def __init__(self, kind, amp=1.0, a=1.0, b=0.0, c=1.0, ell=1.0):
        self.kind, self.amp, self.a, self.b, self.c, self.ell = int(kind), float(amp), float(a), float(b), float(c), float(ell)

    def to_c(self):
        return TgpKernel(self.kind, 0, self.amp, self.a, self.b, self.c, self.ell)

    def __repr__(self):
        return "KernelSpec(kind=%d, amp=%r, a=%r, b=%r, c=%r, ell=%r)" % (self.kind, self.amp, self.a, self.b, self.c, self.ell)


class Factor(object):
    """Device-resident Cholesky factor handle (tgp_factor*), freed with the object."""

    def __init__(self, ctx, handle, n, keepalive=None):
        self._ctx, self._h, self.n = ctx, handle, n
        self._keepalive = keepalive          # a borrowed handle (tgp_factor_borrow): the tensors that own its memory

    def free(self, keep_memory=False):
        """Default: the memory goes back to the device at once (tgp_factor_free) -- a dropped handle must not leave 17 GB (N =
        65 536; 69 GB at 131 072) attached to its context, where it can turn another solve's allocation -- the multi-GPU engine's
        replicated factor, on one rank only -- into an out-of-memory error.  keep_memory=True (the refit paths of
        GPInterpolation and the ML loop, which solve the same size again right away): the packed matrix stays with the context
        as the cache of its next solve of this size (tgp_factor_release), no hipFree + hipMalloc; TGP_FACTOR_FREE=1 overrides."""
        if self._h:
            lib = _lib.load_library()
            if keep_memory and os.environ.get("TGP_FACTOR_FREE") != "1":
                lib.tgp_factor_release(self._ctx, self._h)
            else:
                lib.tgp_factor_free(self._ctx, self._h)                 # (synchronises the context's stream first)
            self._h = None
        self._keepalive = None

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


def kernel_matrix(spec, X, Y=None, ctx=None):
    """amp * k(X, Y) as an (n, m) array; Y=None gives the self kernel with an exact diagonal.
    Replaces kernel.__call__ (treegp/kernels.py:114-126, 249-276, 355-381)."""
    ctx = ctx or _lib.get_ctx()
    lib = _lib.load_library()
    X2 = as_xy(X)
    n = X2.shape[0]
    if Y is None:
        out = np.empty((n, n))
        rc = lib.tgp_kernel_matrix(ctx, C.byref(spec.to_c()), ptr(X2), n, None, 0, ptr(out))
    else:
        Y2 = as_xy(Y)
        m = Y2.shape[0]
        out = np.empty((n, m))
        rc = lib.tgp_kernel_matrix(ctx, C.byref(spec.to_c()), ptr(X2), n, ptr(Y2), m, ptr(out))
    check(ctx, rc, "tgp_kernel_matrix")
    return out


def _dist_engine(n, ctx):
    """The multi-GPU engine this call should go through (treegp_amd.dist.enable / TGP_DIST=1 / backend="dist"), or None.
    A caller that names its own context stays on that context's GPU."""
    if ctx is not None:
        return None
    import sys
    d = sys.modules.get("treegp_amd.dist")
    if d is None:
        if os.environ.get("TGP_DIST") != "1":
            return None
        from . import dist as d
    return d.engine_for(n)


def gp_solve(spec, X, y, y_err=None, keep=False, want_alpha=True, ctx=None):
    """(alpha, logdet, y.alpha, factor|None) for K = amp k(X) + diag(y_err^2).
    Raises numpy.linalg.LinAlgError when K is not positive definite, as scipy.linalg.cholesky
    does at treegp/gp_interp.py:181."""
    eng = _dist_engine(len(X), ctx)
    if eng is not None:
        return eng.gp_solve(spec, X, y, y_err, keep=keep, want_alpha=want_alpha)
    ctx = ctx or _lib.get_ctx()
    lib = _lib.load_library()
    X2 = as_xy(X)
    n = X2.shape[0]
    y = f64(y)
    e = None if y_err is None else f64(y_err)
    alpha = np.empty(n) if want_alpha else None
    logdet, ydota = C.c_double(0.0), C.c_double(0.0)
    h = C.c_void_p()
    rc = lib.tgp_gp_solve(ctx, C.byref(spec.to_c()), ptr(X2), n, ptr(y), ptr(e), ptr(alpha), C.byref(logdet),
                          C.byref(ydota), C.byref(h) if keep else None)
    check(ctx, rc, "tgp_gp_solve")
    if rc > 0:
        raise np.linalg.LinAlgError("%d-th leading minor of the array is not positive definite" % rc)
    return alpha, logdet.value, ydota.value, (Factor(ctx, h, n) if keep else None)


def gp_solve_dense(K, y, y_err=None, keep=False, want_alpha=True, ctx=None):
    """gp_solve for a kernel matrix evaluated by the caller (any scikit-learn kernel tree): K (n, n), lower triangle
    read; y_err^2 is added to the diagonal on the device (tgp_gp_solve_dense)."""
    ctx = ctx or _lib.get_ctx()
    lib = _lib.load_library()
    K = f64(K)
    n = K.shape[0]
    if K.shape != (n, n):
        raise ValueError("K must be square")
    y = f64(y)
    e = None if y_err is None else f64(y_err)
    alpha = np.empty(n) if want_alpha else None
    logdet, ydota = C.c_double(0.0), C.c_double(0.0)
    h = C.c_void_p()
    rc = lib.tgp_gp_solve_dense(ctx, ptr(K), n, ptr(y), ptr(e), ptr(alpha), C.byref(logdet), C.byref(ydota),
                                C.byref(h) if keep else None)
    check(ctx, rc, "tgp_gp_solve_dense")
    if rc > 0:
        raise np.linalg.LinAlgError("%d-th leading minor of the array is not positive definite" % rc)
    return alpha, logdet.value, ydota.value, (Factor(ctx, h, n) if keep else None)


def factor_solve(factor, B, ctx=None):
    """(K + D)^-1 applied to every row of B (nrhs, n) with a kept factor (tgp_factor_solve): the factor is read once per
    sweep for up to 4 right-hand sides."""
    ctx = ctx or factor._ctx
    lib = _lib.load_library()
    B = np.atleast_2d(f64(B))
    if B.shape[1] != factor.n:
        raise ValueError("B must be (nrhs, %d)" % factor.n)
    out = np.empty_like(B)
    check(ctx, lib.tgp_factor_solve(ctx, factor._h, ptr(B), B.shape[0], ptr(out)), "tgp_factor_solve")
    return out


def gp_predict_cov_dense(factor, HT, Kss, ctx=None):
    """Kss - HT (K + D)^-1 HT^T for caller-evaluated HT = kernel(X2, Y=X1) (m, n) and Kss = kernel(X2) (m, m)."""
    ctx = ctx or factor._ctx
    lib = _lib.load_library()
    HT, Kss = f64(HT), f64(Kss)
    m = HT.shape[0]
    if HT.shape != (m, factor.n) or Kss.shape != (m, m):
        raise ValueError("HT must be (m, n) and Kss (m, m)")
    cov = np.empty((m, m))
    check(ctx, lib.tgp_gp_predict_cov_dense(ctx, factor._h, ptr(HT), ptr(Kss), m, ptr(cov)), "tgp_gp_predict_cov_dense")
    return cov


class ResidentProblem(object):
    """X, y, y_err of one GP problem kept on the device (``tgp_dev_alloc`` / ``tgp_h2d``) for a series of solves
    that differ only in the kernel -- the likelihood evaluations of a maximum-likelihood fit.  Uploading the three
    arrays costs ~90 us per call at the host boundary, as much as the whole solve at N = 256.  The buffers are plain
    device memory: any context of the same device may use them.  ``close()`` frees them."""

    def __init__(self, X, y, y_err=None, ctx=None):
        self._ctx = ctx or _lib.get_ctx()
        self._lib = _lib.load_library()
        X2 = as_xy(X)
        self.n = X2.shape[0]
        self._bufs = []
        self.d_X = self._upload(X2)
        self.d_y = self._upload(f64(y))
        self.d_e = None if y_err is None else self._upload(f64(y_err))

    def _upload(self, a):
        d = C.c_void_p()
        check(self._ctx, self._lib.tgp_dev_alloc(self._ctx, a.nbytes, C.byref(d)), "tgp_dev_alloc")
        self._bufs.append(d)
        check(self._ctx, self._lib.tgp_h2d(self._ctx, d, ptr(a), a.nbytes), "tgp_h2d")
        return d

    def close(self):
        for d in self._bufs:
            self._lib.tgp_dev_free(self._ctx, d)
        self._bufs = []

    def __del__(self):
        try:
            self.close()
        except Exception:       # interpreter shutdown
            pass


def gp_solve_resident(spec, problem, ctx=None):
    """(logdet, y.K^-1.y) for a ResidentProblem: ``tgp_d_gp_solve`` without alpha, nothing but the kernel parameters
    crosses the host boundary.  Raises numpy.linalg.LinAlgError like gp_solve."""
    ctx = ctx or _lib.get_ctx()
    lib = _lib.load_library()
    logdet, ydota = C.c_double(0.0), C.c_double(0.0)
    rc = lib.tgp_d_gp_solve(ctx, C.byref(spec.to_c()), problem.d_X, problem.n, problem.d_y, problem.d_e, None,
                            C.byref(logdet), C.byref(ydota), None)
    check(ctx, rc, "tgp_d_gp_solve")
    if rc > 0:
        raise np.linalg.LinAlgError("%d-th leading minor of the array is not positive definite" % rc)
    return logdet.value, ydota.value


def gp_solve_grad_resident(spec, problem, ctx=None):
    """(logdet, y.K^-1.y, d logL / d (log amp, a, b, c)) for a ResidentProblem in one call (``tgp_d_gp_solve_grad``):
    one evaluation of a gradient-driven maximum-likelihood fit.  Gaussian kernels only."""
    ctx = ctx or _lib.get_ctx()
    lib = _lib.load_library()
    logdet, ydota = C.c_double(0.0), C.c_double(0.0)
    grad = np.empty(4)
    rc = lib.tgp_d_gp_solve_grad(ctx, C.byref(spec.to_c()), problem.d_X, problem.n, problem.d_y, problem.d_e,
                                 C.byref(logdet), C.byref(ydota), ptr(grad))
    check(ctx, rc, "tgp_d_gp_solve_grad")
    if rc > 0:
        raise np.linalg.LinAlgError("%d-th leading minor of the array is not positive definite" % rc)
    return logdet.value, ydota.value, grad


def gp_predict(spec, X, alpha, Xs, ctx=None):
    """ys = k(Xs, X) @ alpha without materialising the cross kernel (gp_interp.py:177,183)."""
    eng = _dist_engine(len(X), ctx)
    if eng is not None:
        return eng.gp_predict(spec, X, alpha, Xs)
    ctx = ctx or _lib.get_ctx()
    lib = _lib.load_library()
    X2, Xs2 = as_xy(X), as_xy(Xs)
    alpha = f64(alpha)
    ys = np.empty(Xs2.shape[0])
    rc = lib.tgp_gp_predict(ctx, C.byref(spec.to_c()), ptr(X2), X2.shape[0], ptr(alpha), ptr(Xs2), Xs2.shape[0], ptr(ys))
    check(ctx, rc, "tgp_gp_predict")
    return ys


def gp_predict_cov(spec, factor, X, Xs, ctx=None):
    """Posterior covariance k(Xs,Xs) - HT K^-1 HT^T (gp_interp.py:184-192) from a kept factor."""
    ctx = ctx or factor._ctx
    lib = _lib.load_library()
    X2, Xs2 = as_xy(X), as_xy(Xs)
    m = Xs2.shape[0]
    cov = np.empty((m, m))
    rc = lib.tgp_gp_predict_cov(ctx, factor._h, C.byref(spec.to_c()), ptr(X2), X2.shape[0], ptr(Xs2), m, ptr(cov))
    check(ctx, rc, "tgp_gp_predict_cov")
    return cov


def gp_loglik_grad(spec, factor, X, alpha, ctx=None):
    """1/2 sum_ij (alpha_i alpha_j - [K^-1]_ij) dK_ij/dp for p = (log amp, a, b, c), from the factor and alpha of one
    ``gp_solve(..., keep=True)`` (include/tgp.h, seam S2d).  Gaussian kernels only; ``kernels.spec_jacobian`` maps the four
    numbers to d logL / d theta."""
    ctx = ctx or factor._ctx
    lib = _lib.load_library()
    X2 = as_xy(X)
    alpha = f64(alpha)
    grad = np.empty(4)
    rc = lib.tgp_gp_loglik_grad(ctx, factor._h, C.byref(spec.to_c()), ptr(X2), X2.shape[0], ptr(alpha), ptr(grad))
    check(ctx, rc, "tgp_gp_loglik_grad")
    return grad


# ---- pair binning across ranks: i-tiles / bootstrap resamples dealt to the ranks ------
_pair = threading.local()                  # per thread: the tests run virtual ranks as threads


def _pair_comm():
    return getattr(_pair, "comm", None)


def set_pair_comm(comm):
    """Shard every following kk_twod / kk_log / kk_twod_bootstrap call over the ranks of ``comm``
    (an object with rank, size, all_reduce_sum(tensor), e.g. treegp_amd.dist.TorchComm); None
    switches back to single-GPU.  Every rank must make the same calls with the same data."""
    _pair.comm = comm if (comm is not None and comm.size > 1) else None


def kk_partial(bin_type, x, y, k, w, min_sep, max_sep, nbins, part, nparts, ctx=None):
    """Raw accumulators (3, nbins^2) [TwoD, bin_type 0] or (5, nbins) [Log, 1] of one shard."""
    ctx = ctx or _lib.get_ctx()
    lib = _lib.load_library()
    x, y, k = f64(x), f64(y), f64(k)
    w = None if w is None else f64(w)
    acc = np.zeros((3, nbins * nbins) if bin_type == 0 else (5, nbins))
    rc = lib.tgp_kk_partial(ctx, int(bin_type), ptr(x), ptr(y), ptr(k), ptr(w), len(x), float(min_sep),
                            float(max_sep), int(nbins), int(part), int(nparts), ptr(acc))
    check(ctx, rc, "tgp_kk_partial")
    return acc


def _kk_sharded(bin_type, x, y, k, w, min_sep, max_sep, nbins, ctx, comm):
    import torch
    acc = kk_partial(bin_type, x, y, k, w, min_sep, max_sep, nbins, comm.rank, comm.size, ctx)
    t = torch.from_numpy(acc)
    dev = getattr(comm, "device", None)
    if dev is not None:
        t = t.to(dev)
    comm.all_reduce_sum(t)                               # <= 5 nbins^2 doubles
    acc = t.cpu().numpy()
    ww = acc[1]
    nz = ww != 0.0
    safe = np.where(nz, ww, 1.0)
    xi = np.where(nz, acc[0] / safe, 0.0)
    if bin_type == 0:
        return xi, ww, acc[2]
    return xi, ww, np.where(nz, acc[2] / safe, 0.0), np.where(nz, acc[3] / safe, 0.0), acc[4]


def kk_twod(x, y, k, w, min_sep, max_sep, nbins, ctx=None):
    if _pair_comm() is not None:
        return _kk_sharded(0, x, y, k, w, min_sep, max_sep, nbins, ctx, _pair_comm())
    ctx = ctx or _lib.get_ctx()
    lib = _lib.load_library()
    x, y, k = f64(x), f64(y), f64(k)
    w = None if w is None else f64(w)
    nb2 = nbins * nbins
    xi, wt, npairs = np.empty(nb2), np.empty(nb2), np.empty(nb2)
    rc = lib.tgp_kk_twod(ctx, ptr(x), ptr(y), ptr(k), ptr(w), len(x), float(min_sep), float(max_sep), int(nbins),
                         ptr(xi), ptr(wt), ptr(npairs))
    check(ctx, rc, "tgp_kk_twod")
    return xi, wt, npairs


def kk_log(x, y, k, w, min_sep, max_sep, nbins, ctx=None):
    if _pair_comm() is not None:
        return _kk_sharded(1, x, y, k, w, min_sep, max_sep, nbins, ctx, _pair_comm())
    ctx = ctx or _lib.get_ctx()
    lib = _lib.load_library()
    x, y, k = f64(x), f64(y), f64(k)
    w = None if w is None else f64(w)
    out = [np.empty(nbins) for _ in range(5)]
    rc = lib.tgp_kk_log(ctx, ptr(x), ptr(y), ptr(k), ptr(w), len(x), float(min_sep), float(max_sep), int(nbins),
                        *[ptr(o) for o in out])
    check(ctx, rc, "tgp_kk_log")
    return tuple(out)       # xi, weight, meanr, meanlogr, npairs


def kk_twod_bootstrap(x, y, yv, y_err, idx, min_sep, max_sep, nbins, ctx=None):
    ctx = ctx or _lib.get_ctx()
    lib = _lib.load_library()
    x, y, yv = f64(x), f64(y), f64(yv)
    e = None if y_err is None else f64(y_err)
    idx = np.ascontiguousarray(idx, dtype=np.int64)
    n_boot, n = idx.shape
    assert n == len(x)
    comm = _pair_comm()
    if comm is not None:
        # whole resamples per rank (rows rank, rank + size, ...), then one sum puts the rows together
        import torch
        mine = idx[comm.rank::comm.size]
        full = np.zeros((n_boot, nbins * nbins))
        if len(mine):
            full[comm.rank::comm.size] = _kk_twod_bootstrap_local(ctx, lib, x, y, yv, e, mine, min_sep, max_sep, nbins)
        t = torch.from_numpy(full)
        dev = getattr(comm, "device", None)
        if dev is not None:
            t = t.to(dev)
        comm.all_reduce_sum(t)
        return t.cpu().numpy()
    return _kk_twod_bootstrap_local(ctx, lib, x, y, yv, e, idx, min_sep, max_sep, nbins)


def _kk_twod_bootstrap_local(ctx, lib, x, y, yv, e, idx, min_sep, max_sep, nbins):
    idx = np.ascontiguousarray(idx, dtype=np.int64)
    n_boot, n = idx.shape
    out = np.empty((n_boot, nbins * nbins))
    rc = lib.tgp_kk_twod_bootstrap(ctx, ptr(x), ptr(y), ptr(yv), ptr(e), n, ptr(idx), n_boot, float(min_sep),
                                   float(max_sep), int(nbins), ptr(out))
    check(ctx, rc, "tgp_kk_twod_bootstrap")
    return out


def vcorr_sums(x, y, dx, dy, edges, ctx=None):
    """Raw per-bin sums (7, nbins) of the vector pair correlation over all pairs (utils.py:36-72)."""
    ctx = ctx or _lib.get_ctx()
    lib = _lib.load_library()
    x, y, dx, dy, edges = f64(x), f64(y), f64(dx), f64(dy), f64(edges)
    nbins = len(edges) - 1
    acc = np.zeros((7, nbins))
    rc = lib.tgp_vcorr(ctx, ptr(x), ptr(y), ptr(dx), ptr(dy), len(x), ptr(edges), nbins, ptr(acc))
    check(ctx, rc, "tgp_vcorr")
    return acc


def knn_mean(X0, y0, X, k=4, ctx=None):
    """Uniform mean of the k nearest table rows for every row of X (gp_interp.py:236-238)."""
    ctx = ctx or _lib.get_ctx()
    lib = _lib.load_library()
    X02, X2 = as_xy(X0), as_xy(X)
    y0 = f64(y0)
    out = np.empty(X2.shape[0])
    rc = lib.tgp_knn_mean(ctx, ptr(X02), ptr(y0), X02.shape[0], ptr(X2), X2.shape[0], int(k), ptr(out))
    check(ctx, rc, "tgp_knn_mean")
    return out


_STATS = {"mean": 0, "median": 1, "weighted": 2}


def binned_stat_2d(u, v, values, u_edges, v_edges, statistic="mean", err=None, ctx=None):
    """scipy.stats.binned_statistic_2d(u, v, values, bins=[u_edges, v_edges], statistic) on the GPU
    (meanify.py:76-107).  Returns average, wrms, count, each (len(u_edges)-1, len(v_edges)-1)."""
    ctx = ctx or _lib.get_ctx()
    lib = _lib.load_library()
    u, v, values = f64(u), f64(v), f64(values)
    ue, ve = f64(u_edges), f64(v_edges)
    e = None if err is None else f64(err)
    if statistic not in _STATS:
        raise ValueError("statistic must be one of %s" % sorted(_STATS))
    if statistic == "weighted" and e is None:
        raise ValueError("the weighted statistic needs errors")
    shape = (len(ue) - 1, len(ve) - 1)
    avg, wrms, cnt = np.empty(shape), np.empty(shape), np.empty(shape)
    rc = lib.tgp_binned_stat_2d(ctx, ptr(u), ptr(v), ptr(values), ptr(e), len(u), ptr(ue), len(ue), ptr(ve), len(ve),
                                _STATS[statistic], ptr(avg), ptr(wrms), ptr(cnt))
    check(ctx, rc, "tgp_binned_stat_2d")
    return avg, wrms, cnt


# ---- device-resident tier ---------------------------------------------------------------------
class DeviceBuffer(object):
    def __init__(self, ctx, nbytes):
        self._ctx, self.nbytes = ctx, int(nbytes)
        p = C.c_void_p()
        check(ctx, _lib.load_library().tgp_dev_alloc(ctx, self.nbytes, C.byref(p)), "tgp_dev_alloc")
        self.ptr = p

    @classmethod
    def from_array(cls, ctx, a):
        a = np.ascontiguousarray(a)
        buf = cls(ctx, a.nbytes)
        check(ctx, _lib.load_library().tgp_h2d(ctx, buf.ptr, ptr(a), a.nbytes), "tgp_h2d")
        return buf

    def to_array(self, shape, dtype=np.float64):
        out = np.empty(shape, dtype=dtype)
        assert out.nbytes <= self.nbytes
        check(self._ctx, _lib.load_library().tgp_d2h(self._ctx, ptr(out), self.ptr, out.nbytes), "tgp_d2h")
        return out

    def free(self):
        if self.ptr:
            _lib.load_library().tgp_dev_free(self._ctx, self.ptr)
            self.ptr = None

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass

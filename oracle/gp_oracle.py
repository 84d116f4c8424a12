"""CPU restatement (NumPy/SciPy) of the treegp GP hot path.

TEST INFRASTRUCTURE ONLY -- see ``oracle/__init__.py``.  Every function cites the
reference lines it restates (paths relative to /root/reference).

Parity status
-------------
* kernel matrices, GP solve / predict / posterior covariance / log-likelihood:
  PINNED against the imported reference (``tests/golden/make_golden.py`` ->
  ``tests/golden/*.npz``; checked by ``tests/test_oracle_golden.py``).
* Log-binned pair sums (``kk_log`` with unit weights, ``vcorr``): PINNED against the
  reference's own exact O(N^2) pair binner ``treegp/utils.py:5-74`` (``vcorr``), run
  unmodified by ``tests/golden/make_golden.py`` -> ``g9_vcorr.npz`` (vector fields; a
  scalar field with ``dy = 0`` on a grid that coincides with a KK log-bin grid, where
  ``xi+`` / ``logr`` are KK's ``xi`` / ``meanlogr``; with ``dx = w k`` and ``dx = w`` the ratio of its bin averages is the
  WEIGHTED ``xi`` and their product with the pair count the ``weight``: case f).
* TwoD-pixel pair sums (``kk_twod``) and what TreeCorr does beyond exact binning:
  **parity unpinned**.  The arithmetic lives in the third-party TreeCorr library (PyPI
  ``treecorr``, ``requirements.txt:6`` pins ``>=5.0``), which is neither vendored in the
  reference nor installed in this image; no reference test pins a bin value.  The
  restatement follows the reference's call sites (``treegp/two_pcf.py:283-340``) and
  TreeCorr's published binning rules (exact binning, i.e. ``bin_slop=0``); the reference's
  isotropic path runs TreeCorr with its default ``bin_slop`` (approximate binning), which
  exact binning does not reproduce bit for bit.

* ``meanify_grid``: PINNED against ``treegp/meanify.py`` run by the reference itself (``g13_meanify.npz``, mean and median;
  the reference's weighted branch raises a NameError and has no values).

All kernels are described by plain numbers, never by the product's classes:
``kind`` in {"gauss", "vk", "avk"}:
  gauss : exp(-0.5 * q),            q = a dx^2 + 2 b dx dy + c dy^2
  vk    : von Karman, isotropic:    u = sqrt(dx^2+dy^2)/ell
  avk   : von Karman, Mahalanobis:  u = sqrt(q)
  with  vk(u) = u^(5/6) K_{5/6}(2 pi u) / lim0,  vk(0) = 1.
"""
import numpy as np
from scipy import special, optimize
from scipy.linalg import cholesky, cho_solve

LIM0 = special.gamma(5.0 / 6.0) / (2 * (np.pi ** (5.0 / 6.0)))  # kernels.py:260


def _as2d(X):
    """1-D inputs are handled as 2-D with a zero second coordinate
    (two_pcf.py:250-251 does the same padding for the pair counter)."""
    X = np.atleast_2d(np.asarray(X, dtype=np.float64))
    if X.shape[1] == 1:
        X = np.hstack([X, np.zeros_like(X)])
    return X


def _quad(X, Y, a, b, c):
    dx = X[:, 0][:, None] - Y[:, 0][None, :]
    dy = X[:, 1][:, None] - Y[:, 1][None, :]
    return a * dx * dx + 2.0 * b * dx * dy + c * dy * dy


def _vk_of_u(u):
    """kernels.py:253-262 / 268-276: (u)^(5/6) K_{5/6}(2 pi u) / lim0, limit 1 at u == 0."""
    out = np.zeros_like(u)
    nz = u != 0.0
    out[nz] = u[nz] ** (5.0 / 6.0) * special.kv(5.0 / 6.0, 2 * np.pi * u[nz])
    out[~nz] = LIM0
    out /= LIM0
    return out


def kernel_matrix(kind, X, Y=None, amp=1.0, a=1.0, b=0.0, c=1.0, ell=1.0):
    """amp * k(X, Y).

    gauss: kernels.py:114-126 (AnisotropicRBF; sklearn RBF(l) is gauss with
           a=c=1/l^2, b=0).  vk: kernels.py:249-276.  avk: kernels.py:355-381.
    The amplitude is sklearn's Product(ConstantKernel(amp), k), i.e. amp * k.
    Self-kernel (Y is None) has an exact-1 diagonal (kernels.py:121, 261, 367).
    """
    X = _as2d(X)
    self_k = Y is None
    Yv = X if self_k else _as2d(Y)
    if kind == "gauss":
        K = np.exp(-0.5 * _quad(X, Yv, a, b, c))
    elif kind == "vk":
        dx = X[:, 0][:, None] - Yv[:, 0][None, :]
        dy = X[:, 1][:, None] - Yv[:, 1][None, :]
        K = _vk_of_u(np.sqrt(dx * dx + dy * dy) / ell)
    elif kind == "avk":
        K = _vk_of_u(np.sqrt(_quad(X, Yv, a, b, c)))
    else:
        raise ValueError(kind)
    if self_k:
        np.fill_diagonal(K, 1.0)
    return amp * K


def gp_solve(K, y, y_err):
    """gp_interp.py:180-182 : alpha = (K + diag(y_err^2))^-1 y by upper Cholesky.
    Returns (alpha, logdet) with logdet = sum 2 log diag(U) (log_likelihood.py:33)."""
    A = K + np.eye(len(y)) * y_err ** 2
    U = cholesky(A, overwrite_a=True, lower=False)
    alpha = cho_solve((U, False), y, overwrite_b=False)
    return alpha, float(np.sum(2.0 * np.log(np.diag(U))))


def gp_predict(HT, alpha):
    """gp_interp.py:183."""
    return np.dot(HT, alpha.reshape((len(alpha), 1))).T[0]


def gp_predict_cov(K, y_err, HT, Kss):
    """gp_interp.py:186-191."""
    A = K + np.eye(K.shape[0]) * y_err ** 2
    fact = cholesky(A, lower=True)
    v = cho_solve((fact, True), HT.T)
    return Kss - HT.dot(v)


def log_likelihood(K, y, y_err):
    """log_likelihood.py:28-39."""
    try:
        alpha, logdet = gp_solve(K, y, y_err)
        chi2 = np.dot(y, alpha)
        ll = -0.5 * chi2 - (len(y) / 2.0) * np.log(2.0 * np.pi) - 0.5 * logdet
    except BaseException:
        ll = -np.inf
    return ll


def loglik_grad_invlam(X, y, y_err, amp, invLam, dinvLam):
    """d logL / d theta_k = 1/2 sum_ij (alpha_i alpha_j - [K^-1]_ij) dK_ij/dtheta_k for K = amp exp(-1/2 dX^T invLam dX)
    + diag(y_err^2) (gp_interp.py:180), with the reference's kernel derivative dK/dtheta_k = -1/2 K dX^T dinvLam[k] dX
    (kernels.py:128-150; dinvLam[k] = d invLam / d theta_k).  Also returns d logL / d log amp (scikit-learn's ConstantKernel:
    dK/dtheta = K).  O(n^2) memory per theta: small cases only."""
    X = _as2d(X)
    n, nd = X.shape
    dX = X[:, None, :] - X[None, :, :]
    K = amp * np.exp(-0.5 * np.einsum("ijk,kl,ijl->ij", dX, invLam, dX))
    np.fill_diagonal(K, amp)
    alpha, _ = gp_solve(K, y, y_err)
    Kn = K + np.diag(np.asarray(y_err, float) ** 2)
    M = np.outer(alpha, alpha) - cho_solve((cholesky(Kn, lower=False), False), np.eye(n))
    g_amp = 0.5 * np.sum(M * K)
    g = [0.5 * np.sum(M * (-0.5 * K * np.einsum("ijk,kl,ijl->ij", dX, np.asarray(G, float), dX))) for G in dinvLam]
    return g_amp, np.array(g)


def knn_mean(X0, y0, X, k=4):
    """gp_interp.py:236-238: KNeighborsRegressor(n_neighbors=k) = uniform mean of the
    k nearest (Euclidean) neighbours; brute force restatement."""
    X0 = np.asarray(X0, float)
    X = np.asarray(X, float)
    out = np.empty(len(X))
    for s in range(0, len(X), 4096):
        d2 = ((X[s:s + 4096, None, :] - X0[None, :, :]) ** 2).sum(-1)
        idx = np.argpartition(d2, k - 1, axis=1)[:, :k]
        out[s:s + 4096] = y0[idx].mean(axis=1)
    return out


# --------------------------------------------------------------------------------------
# host-side scalars of the 2-point-correlation fit
# --------------------------------------------------------------------------------------
def twod_mask(nbins):
    """two_pcf.py:311-321: keep half of the point-symmetric TwoD grid."""
    npixels = nbins ** 2
    mask = np.ones((nbins, nbins), dtype=bool)
    boolean_mask_odd = nbins % 2 == 0
    even_or_odd = nbins % 2
    nmask = int((nbins / 2) + even_or_odd)
    mask[nmask:, :] = False
    mask[nmask - 1][nmask:] = boolean_mask_odd
    return mask.reshape(npixels)


def n_bootstrap(npixel):
    """two_pcf.py:375-383 (int() of an fsolve root that sits just below 2*npixel+3)."""
    def f_bias(x, npixel=npixel):
        return ((x - 1.0) / (x - npixel - 2.0)) - 2.0
    results = optimize.fsolve(f_bias, npixel + 10)
    return int(results[0])


def bootstrap_indices(n, n_boot, seed=610639139):
    """two_pcf.py:264-281, 348-351: one PCG64 stream, `integers(0, n-1, size=n)` per
    resample (high exclusive: the last point is never drawn)."""
    rng = np.random.default_rng(seed)
    return np.stack([rng.integers(0, n - 1, size=n) for _ in range(n_boot)])


def twod_pixel_centres(nbins, max_sep):
    """two_pcf.py:323-326: dy varies along axis 0 of the (nbins, nbins) grid, dx = dy.T."""
    bs = 2.0 * max_sep / nbins
    edges_lo = -max_sep + bs * np.arange(nbins)
    dy = np.repeat((edges_lo + (edges_lo + bs)) / 2.0, nbins).reshape(nbins, nbins)
    dx = dy.T
    return np.array([dx.reshape(-1), dy.reshape(-1)]).T


# --------------------------------------------------------------------------------------
# binned scalar-scalar (KK) pair correlation, exact binning  [Log bins pinned, TwoD unpinned: see header]
# --------------------------------------------------------------------------------------
def kk_twod(x, y, k, w, min_sep, max_sep, nbins, chunk=1024):
    """Restates treecorr.KKCorrelation(bin_type="TwoD", bin_slop=0).process(cat) as used
    at two_pcf.py:297-305.  Every unordered pair i<j with r != 0, r >= min_sep,
    max(|dx|,|dy|) < max_sep is deposited in pixel (ix, iy) = (int((dx+max_sep)/bs),
    int((dy+max_sep)/bs)) of d = p_j - p_i AND in the pixel of -d (flat index iy*n+ix).
    Returns xi (0 where weight == 0), weight = sum w_i w_j, npairs -- each of length n^2.
    """
    x = np.asarray(x, float); y = np.asarray(y, float); k = np.asarray(k, float)
    n = len(x)
    w = np.ones(n) if w is None else np.asarray(w, float)
    bs = 2.0 * max_sep / nbins
    nb2 = nbins * nbins
    s_wkk = np.zeros(nb2); s_w = np.zeros(nb2); s_n = np.zeros(nb2)
    for s in range(0, n, chunk):
        e = min(n, s + chunk)
        ii = np.arange(s, e)[:, None]
        jj = np.arange(n)[None, :]
        dx = x[None, :] - x[s:e, None]
        dy = y[None, :] - y[s:e, None]
        rsq = dx * dx + dy * dy
        ok = (jj > ii) & (rsq != 0.0) & (rsq >= min_sep * min_sep) & \
             (np.maximum(np.abs(dx), np.abs(dy)) < max_sep)
        ww = (w[s:e, None] * w[None, :])[ok]
        kk = (k[s:e, None] * k[None, :])[ok]
        dxo = dx[ok]; dyo = dy[ok]
        for sgn in (1.0, -1.0):
            ix = ((sgn * dxo + max_sep) / bs).astype(np.int64)
            iy = ((sgn * dyo + max_sep) / bs).astype(np.int64)
            good = (ix >= 0) & (ix < nbins) & (iy >= 0) & (iy < nbins)
            b = iy[good] * nbins + ix[good]
            s_wkk += np.bincount(b, weights=(ww * kk)[good], minlength=nb2)
            s_w += np.bincount(b, weights=ww[good], minlength=nb2)
            s_n += np.bincount(b, minlength=nb2)
    xi = np.where(s_w != 0.0, s_wkk / np.where(s_w != 0.0, s_w, 1.0), 0.0)
    return xi, s_w, s_n


def kk_log(x, y, k, w, min_sep, max_sep, nbins, chunk=1024):
    """Restates treecorr.KKCorrelation(min_sep, max_sep, nbins).process(cat)
    (two_pcf.py:330-334) with EXACT log binning: bin = int((ln r - ln min_sep)/bs),
    bs = ln(max_sep/min_sep)/nbins, for min_sep <= r < max_sep; each unordered pair once.
    (TreeCorr's default bin_slop makes the reference's own result an approximation of
    this.)  Returns xi, weight, meanr, meanlogr, npairs."""
    x = np.asarray(x, float); y = np.asarray(y, float); k = np.asarray(k, float)
    n = len(x)
    w = np.ones(n) if w is None else np.asarray(w, float)
    bs = np.log(max_sep / min_sep) / nbins
    lmin = np.log(min_sep)
    s_wkk = np.zeros(nbins); s_w = np.zeros(nbins); s_wr = np.zeros(nbins)
    s_wl = np.zeros(nbins); s_n = np.zeros(nbins)
    for s in range(0, n, chunk):
        e = min(n, s + chunk)
        ii = np.arange(s, e)[:, None]
        jj = np.arange(n)[None, :]
        dx = x[None, :] - x[s:e, None]
        dy = y[None, :] - y[s:e, None]
        rsq = dx * dx + dy * dy
        ok = (jj > ii) & (rsq >= min_sep * min_sep) & (rsq < max_sep * max_sep)
        r = np.sqrt(rsq[ok])
        lr = 0.5 * np.log(rsq[ok])
        b = ((lr - lmin) / bs).astype(np.int64)
        good = (b >= 0) & (b < nbins)
        b = b[good]
        ww = (w[s:e, None] * w[None, :])[ok][good]
        kk = (k[s:e, None] * k[None, :])[ok][good]
        s_wkk += np.bincount(b, weights=ww * kk, minlength=nbins)
        s_w += np.bincount(b, weights=ww, minlength=nbins)
        s_wr += np.bincount(b, weights=ww * r[good], minlength=nbins)
        s_wl += np.bincount(b, weights=ww * lr[good], minlength=nbins)
        s_n += np.bincount(b, minlength=nbins)
    nz = s_w != 0.0
    den = np.where(nz, s_w, 1.0)
    return (np.where(nz, s_wkk / den, 0.0), s_w, np.where(nz, s_wr / den, 0.0),
            np.where(nz, s_wl / den, 0.0), s_n)


def comp_2pcf(X, y, y_err, min_sep, max_sep, nbins, anisotropic):
    """two_pcf.py:283-340 on top of the exact binners above."""
    X = _as2d(X)
    w = None if np.sum(y_err) == 0 else 1.0 / y_err ** 2
    k = y - np.mean(y)
    if anisotropic:
        xi, _, _ = kk_twod(X[:, 0], X[:, 1], k, w, min_sep, max_sep, nbins)
        mask = twod_mask(nbins)
        distance = twod_pixel_centres(nbins, max_sep)
        return xi, distance, distance, mask
    xi, _, meanr, _, _ = kk_log(X[:, 0], X[:, 1], k, w, min_sep, max_sep, nbins)
    mask = np.ones_like(xi, dtype=bool)
    coord = np.array([meanr, np.zeros_like(meanr)]).T
    return xi, meanr, coord, mask


# --------------------------------------------------------------------------------------
# synthetic star field of SURVEY.md 8(d) -- shared by tests and bench (inputs only)
# --------------------------------------------------------------------------------------
def correlation_length_matrix(size, e1, e2):
    """two_pcf.py:12-31."""
    e = np.sqrt(e1 ** 2 + e2 ** 2)
    q = (1 - e) / (1 + e)
    phi = 0.5 * np.arctan2(e2, e1)
    rot = np.array([[np.cos(phi), np.sin(phi)], [-np.sin(phi), np.cos(phi)]])
    ell = np.array([[size ** 2, 0], [0, (size * q) ** 2]])
    return np.dot(rot.T, ell.dot(rot))


def meanify_grid(coords, params, params_err=None, bin_spacing=120.0, statistics="mean",
                 lu_min=None, lu_max=None, lv_min=None, lv_max=None):
    """meanify.py:49-137 restated with the same SciPy call the reference makes
    (scipy.stats.binned_statistic_2d): bin edges np.linspace(min, max, int((max-min)/bin_spacing)),
    statistic "mean" / "median", or the three "sum" passes of the weighted branch (:76-101, on the
    same edges; the reference itself stops with a NameError there, :108).  Returns a dict with
    average / wrms (already transposed, :105-106), xedge, yedge, u0, v0 (bin-centre meshgrid,
    :121-124), and the filtered coords0, params0, wrms0 (:131-137)."""
    from scipy.stats import binned_statistic_2d
    coords = np.asarray(coords, float); params = np.asarray(params, float)
    lu_min = np.min(coords[:, 0]) if lu_min is None else lu_min
    lu_max = np.max(coords[:, 0]) if lu_max is None else lu_max
    lv_min = np.min(coords[:, 1]) if lv_min is None else lv_min
    lv_max = np.max(coords[:, 1]) if lv_max is None else lv_max
    nbin_u = int((lu_max - lu_min) / bin_spacing)
    nbin_v = int((lv_max - lv_min) / bin_spacing)
    binning = [np.linspace(lu_min, lu_max, nbin_u), np.linspace(lv_min, lv_max, nbin_v)]
    if statistics == "weighted":
        weights = 1.0 / np.asarray(params_err, float) ** 2
        sum_wpp, xedge, yedge, _ = binned_statistic_2d(coords[:, 0], coords[:, 1], weights * params * params,
                                                       bins=binning, statistic="sum")
        sum_wp = binned_statistic_2d(coords[:, 0], coords[:, 1], weights * params, bins=binning, statistic="sum")[0]
        sum_w = binned_statistic_2d(coords[:, 0], coords[:, 1], weights, bins=binning, statistic="sum")[0]
        with np.errstate(invalid="ignore", divide="ignore"):
            average = sum_wp / sum_w
            wvar = (1.0 / sum_w) * (sum_wpp - 2.0 * average * sum_wp + average * average * sum_w)
            wrms = np.sqrt(wvar)
    else:
        average, xedge, yedge, _ = binned_statistic_2d(coords[:, 0], coords[:, 1], params, bins=binning,
                                                       statistic=statistics)
        wrms = np.zeros_like(average)
    average = average.T
    wrms = wrms.T
    flat_a, flat_w = average.reshape(-1), wrms.reshape(-1)
    keep = np.isfinite(flat_a) & np.isfinite(flat_w)
    u0 = xedge[:-1] + (xedge[1] - xedge[0]) / 2.0
    v0 = yedge[:-1] + (yedge[1] - yedge[0]) / 2.0
    u0, v0 = np.meshgrid(u0, v0)
    coords0 = np.array([u0.reshape(-1), v0.reshape(-1)]).T
    return dict(average=average, wrms=wrms, xedge=xedge, yedge=yedge, u0=u0, v0=v0,
                coords0=coords0[keep], params0=flat_a[keep], wrms0=flat_w[keep])


def vcorr(x, y, dx, dy, rmin=5.0 / 3600.0, rmax=1.5, dlogr=0.05, chunk=512):
    """utils.py:36-74 restated without the index arrays: every pair i < j, complex separation
    d = (x_j - x_i) + i (y_j - y_i), log|d| histogrammed with np.histogram on (bins, range)
    exactly as there, weights 1, log|d|, dx_i dx_j + dy_i dy_j, v_i v_j and v_i v_j conj(d)^2/|d|^2
    (v = dx + i dy).  Pinned by tests/golden/g9_vcorr.npz (the reference's vcorr run unmodified).
    Returns logr, xiplus, ximinus, xicross, xiz2 and the pair counts."""
    x = np.asarray(x, float); y = np.asarray(y, float)
    v = np.asarray(dx, float) + 1j * np.asarray(dy, float)
    n = len(x)
    logrmin = np.log(rmin)
    bins = int(np.ceil(np.log(rmax / rmin) / dlogr))
    hrange = (logrmin, logrmin + bins * dlogr)
    counts = np.zeros(bins); s_logr = np.zeros(bins); s_plus = np.zeros(bins)
    s_z2 = np.zeros(bins, complex); s_minus = np.zeros(bins, complex)
    z = x + 1j * y
    for s in range(0, n, chunk):
        e = min(n, s + chunk)
        i1, i2 = np.nonzero(np.arange(n)[None, :] > np.arange(s, e)[:, None])
        i1 = i1 + s
        dr = 1j * (y[i2] - y[i1])
        dr += x[i2] - x[i1]
        with np.errstate(divide="ignore", invalid="ignore"):
            logdr = np.log(np.absolute(dr))
            h = lambda w=None: np.histogram(logdr, bins=bins, range=hrange, weights=w)[0]   # noqa: E731
            counts += h()
            s_logr += h(np.where(np.isfinite(logdr), logdr, 0.0))
            s_plus += h(v[i1].real * v[i2].real + v[i1].imag * v[i2].imag)
            vv = v[i1] * v[i2]
            s_z2 += h(vv)
            vv = vv * np.conj(dr) * np.conj(dr) / (dr.real * dr.real + dr.imag * dr.imag)
            s_minus += h(np.where(np.isfinite(vv), vv, 0.0))
    with np.errstate(divide="ignore", invalid="ignore"):
        return (s_logr / counts, s_plus / counts, (s_minus / counts).real, (s_minus / counts).imag,
                s_z2 / counts, counts)

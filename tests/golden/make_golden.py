#!/usr/bin/env python3
"""Generate the golden vectors under tests/golden/ by running the REFERENCE itself.

Run in the build container only (needs /root/reference):

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden.py

The reference package's ``__init__`` imports ``two_pcf`` -> ``treecorr``/``iminuit`` and
``meanify`` -> ``fitsio``, none of which is installed here (ordinary ModuleNotFoundError).
Those imports are only needed by the TreeCorr-based optimisers, so this script registers
a bare ``treegp`` package object whose ``__path__`` points at the reference and imports
the three modules of the hot path directly -- ``treegp.kernels``, ``treegp.log_likelihood``
and ``treegp.gp_interp`` -- which run unmodified (no stand-ins for the missing libraries).

Outputs (inputs + the reference's outputs, nothing else):
  g1_c1_rbf1d.npz       config 1: 1-D AnisotropicRBF N=512 / M=1024, alpha, y_pred, logL
  g2_aniso2d.npz        2-D sheared AnisotropicRBF N=1024/M=2048, y_err, white_noise, cov diag
  g3_vonkarman.npz      VonKarman + AnisotropicVonKarman GP incl. coincident X* (lim0 branch)
  g4_kernels.npz        kernel tables K(X), K(X,Y) for RBF / AnisoRBF / VK / AnisoVK
  g5_loglike.npz        return_log_likelihood(theta) at 5 thetas
  g6_meanify.npz        mean-function case, X0/y0 decoded from the reference's FITS fixture
  g7_host_scalars.npz   n_bootstrap / mask sizes, bootstrap index rows, kernel theta maps
  g8_reftests.npz       the reference's own test problems (tests/test_gp_interp.py) end to end
  g9_vcorr.npz          treegp/utils.py vcorr / xiB / comp_eb -- the reference's own exact O(N^2)
                        log-binned pair binner -- on vector fields and, with dy = 0, on a scalar
                        field whose bins coincide with a KK log-bin grid (pins kk_log / vcorr)

  g10_sklearn_kernels.npz  kernel trees only scikit-learn can evaluate (Sum + WhiteKernel, Matern, RationalQuadratic):
                        the reference's predict / covariance / log-likelihood for them (dense-K entry point)

  g11_ml_fit.npz        the reference's maximum-likelihood fits (log_likelihood.optimizer: L-BFGS-B with SciPy's own finite
                        differences) on its test_hyp_search.py-style problems: fitted theta and log-likelihood
  g12_two_pcf_host.npz  the pure host pieces of treegp/two_pcf.py run by the reference itself: get_correlation_length_matrix,
                        robust_2dfit's model / chi2 / linear amplitudes at given parameters (both anisotropic kernels), the
                        bootstrap resampling stream of a seeded two_pcf object, the 1-D coordinate padding

  g14_loglik_grad.npz   d logL / d theta = 1/2 tr((alpha alpha^T - K^-1) dK/dtheta) with the reference's own dK/dtheta
                        (kernel(X, eval_gradient=True), kernels.py:128-150) for Gaussian kernel trees; pins tgp_gp_loglik_grad
  g13_meanify.npz       treegp/meanify.py (add_field + meanify, statistics "mean" and "median", default and explicit limits, with empty
                        bins) run by the reference itself; its module-level ``import fitsio`` (meanify.py:7) resolves to an empty
                        module object -- only save_results, not used here, calls it.  (statistics="weighted" stops with a NameError
                        inside the reference, meanify.py:108, and therefore has no golden values.)

``treegp/utils.py`` starts with ``import treecorr`` (utils.py:2) although ``vcorr``, ``xiB`` and
``comp_eb`` (utils.py:5-107) never touch it.  For G9 only, an EMPTY module object named
``treecorr`` is registered so that this one import statement resolves; it has no attributes,
so any code path that really needed TreeCorr would stop with AttributeError (the route
SURVEY.md 8(c) prescribes).  ``treegp/two_pcf.py`` (G12) likewise starts with ``import treecorr`` and ``import iminuit``
(two_pcf.py:3,6); the functions used here -- :12-31, :34-65, :68-148, :209-281 -- call neither, and both names resolve to
EMPTY module objects.
"""
import importlib
import os
import sys
import types

import numpy as np

REF = "/root/reference"
OUT = os.path.dirname(os.path.abspath(__file__))


def load_reference():
    pkg = types.ModuleType("treegp")
    pkg.__path__ = [os.path.join(REF, "treegp")]
    sys.modules["treegp"] = pkg
    k = importlib.import_module("treegp.kernels")
    ll = importlib.import_module("treegp.log_likelihood")
    gi = importlib.import_module("treegp.gp_interp")
    # what treegp/__init__.py:9-16 would have bound (hot-path names only)
    pkg.GPInterpolation = gi.GPInterpolation
    pkg.log_likelihood = ll.log_likelihood
    pkg.eval_kernel = k.eval_kernel
    pkg.AnisotropicRBF = k.AnisotropicRBF
    pkg.VonKarman = k.VonKarman
    pkg.AnisotropicVonKarman = k.AnisotropicVonKarman
    return pkg


def load_reference_two_pcf():
    """treegp/two_pcf.py unmodified; ``import treecorr`` / ``import iminuit`` resolve to empty module objects."""
    sys.modules.setdefault("treecorr", types.ModuleType("treecorr"))
    sys.modules.setdefault("iminuit", types.ModuleType("iminuit"))
    return importlib.import_module("treegp.two_pcf")


def load_reference_meanify():
    """treegp/meanify.py unmodified; ``import fitsio`` resolves to an empty module object (only save_results uses it)."""
    sys.modules.setdefault("fitsio", types.ModuleType("fitsio"))
    return importlib.import_module("treegp.meanify")


def load_reference_utils():
    """treegp/utils.py unmodified; its module-level ``import treecorr`` resolves to an empty
    module object (see the header)."""
    sys.modules.setdefault("treecorr", types.ModuleType("treecorr"))
    return importlib.import_module("treegp.utils")


def corr_len_matrix(size, e1, e2):
    # same formula as tests/treegp_test_helper.py:24-44 (inputs to kernel strings only)
    e = np.sqrt(e1 ** 2 + e2 ** 2)
    q = (1 - e) / (1 + e)
    phi = 0.5 * np.arctan2(e2, e1)
    rot = np.array([[np.cos(phi), np.sin(phi)], [-np.sin(phi), np.cos(phi)]])
    ell = np.array([[size ** 2, 0], [0, (size * q) ** 2]])
    return np.dot(rot.T, ell.dot(rot))


def sine_field(rng, X, nmodes=8):
    """Smooth multi-sine field of SURVEY.md 8(d) (inputs only)."""
    X2 = X if X.shape[1] == 2 else np.hstack([X, np.zeros_like(X)])
    y = np.zeros(len(X2))
    for _ in range(nmodes):
        A = rng.uniform(0.2, 1.0)
        f = rng.uniform(1.0, 6.0, size=2)
        ph = rng.uniform(0, 2 * np.pi)
        y += A * np.sin(2 * np.pi * (X2 @ f) + ph)
    return y


def read_fits_bintable_row0(path):
    """Minimal FITS BINTABLE reader (one row, 'nD' columns) for the reference fixture."""
    raw = open(path, "rb").read()
    pos, hdus = 0, []
    while pos < len(raw):
        cards = {}
        while True:
            blk = raw[pos:pos + 2880]
            pos += 2880
            end = False
            for i in range(0, 2880, 80):
                c = blk[i:i + 80].decode("ascii")
                if c.startswith("END"):
                    end = True
                    break
                if c[8:10] == "= ":
                    cards[c[:8].strip()] = c[10:].split("/")[0].strip().strip("'").strip()
            if end:
                break
        nbytes = 0
        if int(cards.get("NAXIS", 0)) > 0:
            nbytes = int(cards["NAXIS1"]) * int(cards["NAXIS2"])
        hdus.append((cards, pos))
        pos += (nbytes + 2879) // 2880 * 2880
    cards, start = hdus[1]
    out, off = {}, start
    for i in range(1, int(cards["TFIELDS"]) + 1):
        n = int(cards["TFORM%d" % i][:-1])
        out[cards["TTYPE%d" % i]] = np.frombuffer(raw, dtype=">f8", count=n, offset=off).astype(np.float64)
        off += 8 * n
    return out


def make_g14(tg):
    """Gradient of the log marginal likelihood with the REFERENCE's kernel derivative (``kernel(X, eval_gradient=True)``:
    treegp/kernels.py:128-150 for AnisotropicRBF, composed by scikit-learn's Product / ConstantKernel / RBF) in the standard
    formula 1/2 tr((alpha alpha^T - K^-1) dK/dtheta_k), K = kernel(X) + diag(y_err^2) as at gp_interp.py:180.  The reference
    never forms this gradient itself (log_likelihood.py:57 passes no jac); what is pinned is its dK/dtheta convention and the
    likelihood it belongs to (logL from the reference's own log_likelihood class)."""
    from scipy import linalg
    out = {}
    rng = np.random.default_rng(1414)
    cases = []
    X = rng.uniform(0, 1, (700, 2))
    invL = np.linalg.inv(corr_len_matrix(0.08, 0.25, -0.1))
    cases.append(("arbf2d", "0.8**2 * AnisotropicRBF(invLam={0!r})".format(invL), X, 0.03 * rng.uniform(0.8, 1.2, 700)))
    X = rng.uniform(-10, 10, (300, 1))
    cases.append(("arbf1d", "1.3**2 * AnisotropicRBF(scale_length=[1.5])", X, 0.1 * np.ones(300)))
    X = rng.uniform(0, 1, (520, 2))
    cases.append(("rbf2d", "0.6**2 * RBF(0.07)", X, 0.02 * rng.uniform(0.8, 1.2, 520)))
    X = rng.uniform(0, 1, (2300, 2))
    invL = np.linalg.inv(corr_len_matrix(0.05, 0.2, 0.1))
    cases.append(("arbf2d_big", "1.0**2 * AnisotropicRBF(invLam={0!r})".format(invL), X, 0.03 * rng.uniform(0.8, 1.2, 2300)))
    for tag, kern, X, y_err in cases:
        k = tg.eval_kernel(kern)
        y = sine_field(rng, X if X.shape[1] == 2 else X / 20.0) + y_err * rng.standard_normal(len(X))
        K, dK = k(X, eval_gradient=True)
        Kn = K + np.eye(len(X)) * y_err ** 2
        fac = linalg.cho_factor(Kn, lower=False)
        alpha = linalg.cho_solve(fac, y)
        Kinv = linalg.cho_solve(fac, np.eye(len(X)))
        grad = 0.5 * np.einsum("ij,ijk->k", np.outer(alpha, alpha) - Kinv, dK)
        logL = tg.log_likelihood(X, y, y_err).log_likelihood(k)
        out.update({tag + "_kernel": kern, tag + "_X": X, tag + "_y": y, tag + "_y_err": y_err, tag + "_theta": k.theta,
                    tag + "_grad": grad, tag + "_logL": logL})
    out["tags"] = np.array([c[0] for c in cases])
    # the kernel derivatives themselves, kernel(X, eval_gradient=True), 40 points: AnisotropicRBF 2-D / 1-D under a Product
    # (kernels.py:128-150 + scikit-learn's Product rule) and what VonKarman returns there (kernels.py:278-288)
    Xk = rng.uniform(0, 1, (40, 2))
    kg = {"kg_arbf2d": "0.8**2 * AnisotropicRBF(invLam={0!r})".format(np.linalg.inv(corr_len_matrix(0.3, 0.25, -0.1))),
          "kg_arbf1d": "1.3**2 * AnisotropicRBF(scale_length=[0.4])", "kg_arbf_bare": "AnisotropicRBF(scale_length=[0.5, 0.2])",
          "kg_vk": "1.1**2 * VonKarman(length_scale=0.7)"}
    for tag, kern in kg.items():
        Xc = Xk[:, :1] if tag == "kg_arbf1d" else Xk
        K, dK = tg.eval_kernel(kern)(Xc, eval_gradient=True)
        out.update({tag + "_kernel": kern, tag + "_K": K, tag + "_dK": dK})
    out["kg_X"] = Xk
    out["kg_tags"] = np.array(list(kg))
    np.savez(os.path.join(OUT, "g14_loglik_grad.npz"), **out)


def main():
    tg = load_reference()
    GP = tg.GPInterpolation
    if "--only-g14" in sys.argv:
        make_g14(tg)
        return

    # ---------------- G1: config 1 ------------------------------------------------
    rng = np.random.default_rng(20240613)
    N, M = 512, 1024
    X = rng.uniform(-10, 10, (N, 1))
    kern = "1.0**2 * AnisotropicRBF(scale_length=[2.0])"
    noise = 0.1
    y = sine_field(rng, X / 20.0) + noise * rng.standard_normal(N)
    y_err = noise * np.ones(N)
    Xs = np.linspace(-10, 10, M).reshape(M, 1)
    gp = GP(kernel=kern, optimizer="none", normalize=True, white_noise=0.0)
    gp.initialize(X, y, y_err=y_err)
    yp = gp.predict(Xs)
    np.savez(os.path.join(OUT, "g1_c1_rbf1d.npz"), kernel=kern, X=X, y=y, y_err=y_err, Xs=Xs,
             y_pred=yp, alpha=gp._alpha, mean=gp._mean, logL=gp.return_log_likelihood())

    # ---------------- G2: sheared 2-D AnisotropicRBF -------------------------------
    rng = np.random.default_rng(7)
    N, M = 1024, 2048
    X = rng.uniform(0, 1, (N, 2))
    invL = np.linalg.inv(corr_len_matrix(0.05, 0.2, 0.1))
    kern = "0.8**2 * AnisotropicRBF(invLam={0!r})".format(invL)
    y = 0.3 + sine_field(rng, X) + 0.03 * rng.standard_normal(N)
    y_err = 0.03 * rng.uniform(0.8, 1.2, N)
    Xs = rng.uniform(0, 1, (M, 2))
    gp = GP(kernel=kern, optimizer="none", normalize=True, white_noise=0.01)
    gp.initialize(X, y, y_err=y_err)
    yp, cov = gp.predict(Xs[:256], return_cov=True)
    yp_all = gp.predict(Xs)
    invL = gp.kernel_template.k2.invLam      # the repr() in the kernel string truncated it
    np.savez(os.path.join(OUT, "g2_aniso2d.npz"), kernel=kern, invLam=invL, amp=0.8 ** 2, X=X, y=y,
             y_err=y_err, white_noise=0.01, Xs=Xs, y_pred=yp_all, alpha=gp._alpha, mean=gp._mean,
             cov256=cov, y_err_eff=gp._y_err)

    # ---------------- G3: von Karman kernels, with coincident prediction points -----
    rng = np.random.default_rng(11)
    N, M = 600, 500
    X = rng.uniform(0, 1, (N, 2))
    y = sine_field(rng, X) + 0.05 * rng.standard_normal(N)
    y_err = 0.05 * rng.uniform(0.8, 1.2, N)
    Xs = rng.uniform(0, 1, (M, 2))
    Xs[:50] = X[:50]                                  # exercises kernels.py:274-276 / 379-381
    out = dict(X=X, y=y, y_err=y_err, Xs=Xs)
    invL = np.linalg.inv(corr_len_matrix(0.3, 0.2, -0.1))
    for tag, kern in (("vk", "1.3**2 * VonKarman(length_scale=0.4)"),
                      ("avk", "1.3**2 * AnisotropicVonKarman(invLam={0!r})".format(invL))):
        gp = GP(kernel=kern, optimizer="none", normalize=True, white_noise=0.0)
        gp.initialize(X, y, y_err=y_err)
        yp, cov = gp.predict(Xs[:200], return_cov=True)
        if tag == "avk":
            invL = gp.kernel_template.k2.invLam
        out.update({tag + "_kernel": kern, tag + "_y_pred": gp.predict(Xs), tag + "_alpha": gp._alpha,
                    tag + "_cov200": cov, tag + "_logL": gp.return_log_likelihood()})
    out["avk_invLam"] = invL
    np.savez(os.path.join(OUT, "g3_vonkarman.npz"), **out)

    # ---------------- G4: kernel tables ------------------------------------------------
    rng = np.random.default_rng(3)
    n = 64
    # distances spanning ~1e-7 .. 1e3 so that Bessel arguments cover 1e-6 .. > 698 (exact 0)
    rad = 10.0 ** rng.uniform(-7, 2.3, n)
    ang = rng.uniform(0, 2 * np.pi, n)
    X = np.array([rad * np.cos(ang), rad * np.sin(ang)]).T
    Y = np.vstack([X[:16], rng.uniform(-3, 3, (32, 2))])      # 16 coincident rows
    invL = np.linalg.inv(corr_len_matrix(1.7, 0.3, -0.2))
    kerns = {
        "rbf": "2.0**2 * RBF(0.45)",
        "arbf": "0.5**2 * AnisotropicRBF(invLam={0!r})".format(invL),
        "vk": "1.5**2 * VonKarman(length_scale=3.0)",
        "avk": "0.7**2 * AnisotropicVonKarman(invLam={0!r})".format(invL),
        "vk_noamp": "VonKarman(length_scale=0.02)",
    }
    out = dict(X=X, Y=Y, invLam=invL)
    for tag, s in kerns.items():
        k = tg.eval_kernel(s)
        out[tag + "_str"] = s
        out[tag + "_self"] = k(X)
        out[tag + "_cross"] = k(Y, Y=X)
        out[tag + "_theta"] = k.theta
        if hasattr(getattr(k, "k2", None), "invLam"):
            out[tag + "_invLam"] = k.k2.invLam      # as parsed back from the repr() string
    X1 = np.sort(rng.uniform(-10, 10, 48)).reshape(-1, 1)
    out["X1"] = X1
    for tag, s in (("rbf1d", "2.0**2 * RBF(2.0)"), ("vk1d", "2.0**2 * VonKarman(2.0)"),
                   ("arbf1d", "1.0**2 * AnisotropicRBF(scale_length=[2.0])")):
        k = tg.eval_kernel(s)
        out[tag + "_str"] = s
        out[tag + "_self"] = k(X1)
        out[tag + "_cross"] = k(X1[:10] + 0.5, Y=X1)
    np.savez(os.path.join(OUT, "g4_kernels.npz"), **out)

    # ---------------- G5: log-likelihood at several thetas -----------------------------
    rng = np.random.default_rng(5)
    N = 400
    X = rng.uniform(0, 1, (N, 2))
    y = sine_field(rng, X) + 0.05 * rng.standard_normal(N)
    y_err = 0.05 * rng.uniform(0.8, 1.2, N)
    invL = np.linalg.inv(corr_len_matrix(0.2, 0.1, 0.1))
    kern = "1.0**2 * AnisotropicRBF(invLam={0!r})".format(invL)
    gp = GP(kernel=kern, optimizer="none", normalize=True)
    gp.initialize(X, y, y_err=y_err)
    th0 = gp.kernel.theta.copy()
    invL = gp.kernel_template.k2.invLam
    thetas = np.array([th0, th0 + 0.1, th0 - 0.2, th0 * 1.1, th0 + np.array([0.5, -0.1, 0.1, 0.3])])
    lls = np.array([gp.return_log_likelihood(theta=t) for t in thetas])
    # a theta that makes K numerically singular with zero noise -> -inf (log_likelihood.py:38-39)
    gp2 = GP(kernel="1.0**2 * AnisotropicRBF(scale_length=[50., 50.])", optimizer="none", normalize=False)
    gp2.initialize(X, y, y_err=np.zeros(N))
    ll_bad = gp2.return_log_likelihood()
    np.savez(os.path.join(OUT, "g5_loglike.npz"), kernel=kern, invLam=invL, X=X, y=y, y_err=y_err, thetas=thetas,
             logL=lls, mean=gp._mean, logL_singular=ll_bad)

    # ---------------- G6: mean-function (meanify output) case ---------------------------
    fx = read_fits_bintable_row0(os.path.join(REF, "tests", "inputs", "mean_gp_stat_mean.fits"))
    X0 = fx["COORDS0"].reshape(2500, 2)             # TDIM (2,2500): fastest axis first
    y0 = fx["PARAMS0"]
    rng = np.random.default_rng(13)
    N, M = 800, 600
    X = rng.uniform(0, 2048, (N, 2))
    avg = 0.02 + 5e-8 * (X[:, 0] - 1024) ** 2 + 5e-8 * (X[:, 1] - 1024) ** 2
    y = avg + 0.03 * sine_field(rng, X / 2048.0) + 0.003 * rng.standard_normal(N)
    y_err = 0.003 * rng.uniform(0.8, 1.2, N)
    Xs = rng.uniform(0, 2048, (M, 2))
    invL = np.linalg.inv(corr_len_matrix(300.0, 0.2, 0.1))
    kern = "0.03**2 * AnisotropicRBF(invLam={0!r})".format(invL)
    gp = GP(kernel=kern, optimizer="none", normalize=True, n_neighbors=4)
    gp._X0, gp._y0 = X0, y0                      # what gp_interp.py:100-107 does with fitsio
    gp.initialize(X, y, y_err=y_err)
    yp = gp.predict(Xs)
    np.savez(os.path.join(OUT, "g6_meanify.npz"), kernel=kern, invLam=gp.kernel_template.k2.invLam, X0=X0, y0=y0, X=X, y=y, y_err=y_err, Xs=Xs,
             spatial_average=gp._spatial_average, mean=gp._mean, y_pred=yp, alpha=gp._alpha,
             spatial_average_Xs=gp._build_average_meanify(Xs))

    # ---------------- G7: host-side scalars ----------------------------------------------
    # two_pcf.py cannot be imported (treecorr), so these restate its few NumPy/SciPy lines
    # verbatim in behaviour: mask two_pcf.py:311-321, n_bootstrap :375-383, rng :264-281.
    from scipy import optimize
    out = {}
    for nb in (15, 20, 21):
        mask = np.ones((nb, nb), dtype=bool)
        nmask = int((nb / 2) + nb % 2)
        mask[nmask:, :] = False
        mask[nmask - 1][nmask:] = (nb % 2 == 0)
        npix = int(mask.sum())

        def f_bias(x, npixel=npix):
            return ((x - 1.0) / (x - npixel - 2.0)) - 2.0
        out["npix_%d" % nb] = npix
        out["nboot_%d" % nb] = int(optimize.fsolve(f_bias, npix + 10)[0])
        out["mask_%d" % nb] = mask.reshape(-1)
    r = np.random.default_rng(610639139)
    out["boot_n10"] = np.stack([r.integers(0, 9, size=10) for _ in range(4)])
    r = np.random.default_rng(610639139)
    out["boot_n1000"] = np.stack([r.integers(0, 999, size=1000) for _ in range(3)])
    np.savez(os.path.join(OUT, "g7_host_scalars.npz"), **out)

    # ---------------- G8: the reference's own GP test problems (tests/test_gp_interp.py) ---
    out = {}
    np.random.seed(42)
    npts = 40
    x = np.random.uniform(-10, 10, npts).reshape((npts, 1))
    for tag, kern in (("rbf", "1.000000**2 * RBF(2.000000)"), ("vk", "2.000000**2 * VonKarman(2.000000)")):
        K = tg.eval_kernel(kern)(x)
        np.random.seed(43)
        yy = np.random.multivariate_normal(np.zeros(npts), K) + np.random.normal(scale=0.1, size=npts)
        gp = GP(kernel=kern, optimizer="none", white_noise=0.0)
        gp.initialize(x, yy, y_err=0.1 * np.ones(npts))
        yp, cov = gp.predict(x, return_cov=True)
        new_x = np.linspace(np.max(x) + 12.0, np.max(x) + 14.0, npts).reshape((npts, 1))
        gpb = GP(kernel=kern, optimizer="none", normalize=False, white_noise=0.0)
        gpb.initialize(x, yy, y_err=0.1 * np.ones(npts))
        ypb, covb = gpb.predict(new_x, return_cov=True)
        out.update({tag + "_kernel": kern, tag + "_y": yy, tag + "_y_pred": yp, tag + "_cov": cov,
                    tag + "_new_x": new_x, tag + "_y_far": ypb, tag + "_cov_far": covb})
    out["x"] = x
    np.savez(os.path.join(OUT, "g8_reftests.npz"), **out)
    # ---------------- G9: the reference's own exact pair binner (treegp/utils.py:5-107) ----
    ut = load_reference_utils()
    out = {}
    # (a) vector field, default-like log bins scaled to a unit field
    rng = np.random.default_rng(99)
    n = 1500
    x, y = rng.uniform(0, 1, n), rng.uniform(0, 1, n)
    dx = np.sin(3 * x) + 0.1 * rng.standard_normal(n)
    dy = np.cos(2 * y) * np.sin(x) + 0.1 * rng.standard_normal(n)
    kw = dict(rmin=0.01, rmax=0.8, dlogr=0.2)
    logr, xip, xim, xix, xiz2 = ut.vcorr(x, y, dx, dy, **kw)
    xie, xib, logr_eb = ut.comp_eb(x, y, dx, dy, **kw)
    out.update(a_x=x, a_y=y, a_dx=dx, a_dy=dy, a_rmin=kw["rmin"], a_rmax=kw["rmax"], a_dlogr=kw["dlogr"],
               a_logr=logr, a_xiplus=xip, a_ximinus=xim, a_xicross=xix, a_xiz2=xiz2, a_xie=xie, a_xib=xib)
    # (b) the reference's default arguments (rmin 5/3600, rmax 1.5, dlogr 0.05: 140 bins, some empty -> nan)
    n = 700
    x, y = rng.uniform(0, 2, n), rng.uniform(0, 2, n)
    dx, dy = rng.standard_normal(n), rng.standard_normal(n)
    with np.errstate(invalid="ignore", divide="ignore"):
        logr, xip, xim, xix, xiz2 = ut.vcorr(x, y, dx, dy)
    out.update(b_x=x, b_y=y, b_dx=dx, b_dy=dy, b_logr=logr, b_xiplus=xip, b_ximinus=xim, b_xicross=xix, b_xiz2=xiz2)
    # (c) scalar field through the same binner: dy = 0 makes xi+ = <k_i k_j> and logr = <log r> per bin,
    #     i.e. what KKCorrelation(min_sep, max_sep, nbins) with exact binning and unit weights returns as
    #     xi / meanlogr.  dlogr is chosen so that vcorr's bins ARE that log grid (checked).
    n = 2000
    x, y = rng.uniform(0, 1, n), rng.uniform(0, 1, n)
    kf = sine_field(rng, np.array([x, y]).T)
    kf = kf - kf.mean()
    min_sep, max_sep, nbins = 0.02, 0.5, 20
    dlogr = np.log(max_sep / min_sep) / nbins
    assert int(np.ceil(np.log(max_sep / min_sep) / dlogr)) == nbins
    logr, xip, xim, xix, xiz2 = ut.vcorr(x, y, kf, np.zeros(n), rmin=min_sep, rmax=max_sep, dlogr=dlogr)
    assert len(logr) == nbins
    out.update(c_x=x, c_y=y, c_k=kf, c_min_sep=min_sep, c_max_sep=max_sep, c_nbins=nbins, c_dlogr=dlogr,
               c_logr=logr, c_xiplus=xip, c_ximinus=xim, c_xicross=xix, c_xiz2=xiz2)
    # (f) per-point weights through the same unweighted binner: with dx = w k the bin averages are <w_i w_j k_i k_j> and with
    #     dx = w they are <w_i w_j>; their ratio is the WEIGHTED KK xi (the pair counts cancel), <w_i w_j> times the pair count
    #     is KK's weight.  Same points and bins as (c).
    wts = 1.0 / (0.03 * rng.uniform(0.8, 1.2, n)) ** 2
    f1 = ut.vcorr(x, y, wts * kf, np.zeros(n), rmin=min_sep, rmax=max_sep, dlogr=dlogr)
    f2 = ut.vcorr(x, y, wts, np.zeros(n), rmin=min_sep, rmax=max_sep, dlogr=dlogr)
    out.update(f_w=wts, f_xiplus_wk=f1[1], f_xiplus_w=f2[1])
    # (d) the subsampling branch (utils.py:28-35): legacy global RandomState, seeded here
    n = 900
    x, y = rng.uniform(0, 1, n), rng.uniform(0, 1, n)
    dx, dy = rng.standard_normal(n), rng.standard_normal(n)
    np.random.seed(1234)
    logr, xip, xim, xix, xiz2 = ut.vcorr(x, y, dx, dy, rmin=0.02, rmax=0.7, dlogr=0.3, maxpts=400)
    out.update(d_x=x, d_y=y, d_dx=dx, d_dy=dy, d_seed=1234, d_maxpts=400, d_rmin=0.02, d_rmax=0.7, d_dlogr=0.3,
               d_logr=logr, d_xiplus=xip, d_ximinus=xim, d_xicross=xix, d_xiz2=xiz2)
    # (e) xiB on its own (utils.py:77-86)
    lr = np.sort(rng.uniform(-5, 0, 30))
    p_, m_ = rng.standard_normal(30), rng.standard_normal(30)
    out.update(e_logr=lr, e_xiplus=p_, e_ximinus=m_, e_xib=ut.xiB(lr, p_, m_))
    np.savez(os.path.join(OUT, "g9_vcorr.npz"), **out)
    # ---------------- G10: kernel trees outside the parametrised device kernels -------------------
    rng = np.random.default_rng(31)
    N, M = 700, 300
    X = rng.uniform(0, 1, (N, 2))
    y = 0.2 + sine_field(rng, X) + 0.05 * rng.standard_normal(N)
    y_err = 0.05 * rng.uniform(0.8, 1.2, N)
    Xs = rng.uniform(0, 1, (M, 2))
    out = dict(X=X, y=y, y_err=y_err, Xs=Xs)
    for tag, kern in (("sumwhite", "1.0**2 * RBF(0.15) + WhiteKernel(1e-3)"),
                      ("matern", "0.9**2 * Matern(length_scale=0.2, nu=1.5)"),
                      ("rq", "1.1**2 * RationalQuadratic(length_scale=0.2, alpha=0.7)"),
                      ("sum2", "0.7**2 * RBF(0.1) + 0.5**2 * AnisotropicRBF(scale_length=[0.4, 0.25])")):
        gp = GP(kernel=kern, optimizer="none", normalize=True, white_noise=0.0)
        gp.initialize(X, y, y_err=y_err)
        yp, cov = gp.predict(Xs[:128], return_cov=True)
        out.update({tag + "_kernel": kern, tag + "_y_pred": gp.predict(Xs), tag + "_alpha": gp._alpha, tag + "_cov128": cov,
                    tag + "_logL": gp.return_log_likelihood()})
    np.savez(os.path.join(OUT, "g10_sklearn_kernels.npz"), **out)
    # ---------------- G11: the reference's maximum-likelihood fits -------------------------------------
    out = {}
    np.random.seed(42)
    x1 = np.random.uniform(-10, 10, 100).reshape((100, 1))
    for tag, kern, X in (("rbf1d", "1.000000**2 * RBF(0.500000)", x1),):
        K = tg.eval_kernel(kern)(X)
        np.random.seed(43)
        yy = np.random.multivariate_normal(np.zeros(len(X)), K) + np.random.normal(scale=0.01, size=len(X))
        gp = GP(kernel="0.7**2 * RBF(0.8)", optimizer="log-likelihood", normalize=True)
        gp.initialize(X, yy, y_err=0.01 * np.ones(len(X)))
        gp.solve()
        out.update({tag + "_X": X, tag + "_y": yy, tag + "_kernel0": "0.7**2 * RBF(0.8)", tag + "_theta": gp.kernel.theta,
                    tag + "_logL": gp._optimizer._logL, tag + "_theta_true": tg.eval_kernel(kern).theta})
    rng = np.random.default_rng(77)
    X2 = rng.uniform(-10, 10, (300, 2))
    invL = np.linalg.inv(corr_len_matrix(2.0, 0.2, -0.1))
    ktrue = tg.eval_kernel("1.5**2 * AnisotropicRBF(invLam={0!r})".format(invL))
    yy = rng.multivariate_normal(np.zeros(300), ktrue(X2)) + 0.02 * rng.standard_normal(300)
    k0 = "1.0**2 * AnisotropicRBF(invLam={0!r})".format(np.linalg.inv(corr_len_matrix(1.5, 0.0, 0.0)))
    gp = GP(kernel=k0, optimizer="log-likelihood", normalize=True)
    gp.initialize(X2, yy, y_err=0.02 * np.ones(300))
    gp.solve()
    out.update(arbf2d_X=X2, arbf2d_y=yy, arbf2d_kernel0=k0, arbf2d_theta=gp.kernel.theta, arbf2d_logL=gp._optimizer._logL,
               arbf2d_theta_true=ktrue.theta)
    np.savez(os.path.join(OUT, "g11_ml_fit.npz"), **out)

    # ---------------- G12: pure host pieces of treegp/two_pcf.py ------------------------------------------
    tp = load_reference_two_pcf()
    out = {}
    pars = np.array([[0.05, 0.2, 0.1], [1.7, -0.3, 0.25], [3000.0, 0.0, 0.0], [0.4, 0.0, -0.6]])
    out["clm_params"] = pars
    out["clm"] = np.stack([tp.get_correlation_length_matrix(*q) for q in pars])
    nb, mx = 11, 0.9
    c = (np.arange(nb) + 0.5) * 2 * mx / nb - mx
    xx, yv = np.meshgrid(c, c)
    px, py = xx.ravel(), yv.ravel()
    rng = np.random.default_rng(123)
    mask = np.ones(nb * nb, dtype=bool)
    mask[nb * nb // 2 + 1:] = False                                # half plane, like two_pcf.py:311-321
    nm = int(mask.sum())
    A = rng.standard_normal((nm, nm))
    W = A @ A.T / nm + np.eye(nm)                                  # positive-definite weights of the kept pixels (two_pcf.py:384-387)
    out.update(fit_x=px, fit_y=py, fit_W=W, fit_mask=mask)
    trial = np.array([[0.35, 0.1, -0.05], [0.5, 0.0, 0.0], [0.2, -0.3, 0.3], [0.6, 1.2, 0.0], [np.nan, 0.0, 0.0]])
    out["fit_trial"] = trial
    for tag, kstr in (("arbf", "1.0**2 * AnisotropicRBF(invLam=array([[1., 0.], [0., 1.]]))"),
                      ("avk", "AnisotropicVonKarman(invLam=array([[1., 0.], [0., 1.]]))")):
        kern = tg.eval_kernel(kstr)
        kcls = tp.get_kernel_class(kern)
        truth = 1.3 ** 2 * kcls(invLam=np.linalg.inv(tp.get_correlation_length_matrix(0.35, 0.1, -0.05)))
        data = truth(np.array([px, py]).T, Y=np.zeros((nb * nb, 2)))[:, 0] + 0.03 + 0.01 * rng.standard_normal(nb * nb)
        fit = tp.robust_2dfit(kern, data, px, py, W, mask=mask)
        chi2, alpha, model = [], [], []
        for q in trial:
            v = fit.chi2(q)
            chi2.append(v)
            ok = np.isfinite(v)
            alpha.append(np.ravel(fit.alpha) if ok else [np.nan, np.nan])
            m = fit._model_skl(1.0, q[0], q[1], q[2]) if np.isfinite(q[0]) else None
            model.append(m if m is not None else np.full(nb * nb, np.nan))
        out.update({tag + "_kernel": kstr, tag + "_data": data, tag + "_chi2": np.array(chi2, dtype=float), tag + "_alpha": np.array(alpha, dtype=float),
                    tag + "_model": np.array(model, dtype=float)})
    rng = np.random.default_rng(5)
    n = 57
    Xb = rng.uniform(0, 1, (n, 2)); yb = rng.standard_normal(n); eb = rng.uniform(0.1, 0.2, n)
    obj = tp.two_pcf(Xb, yb, eb, 0.0, 0.3, nbins=7, anisotropic=True)
    draws = [obj.resample_bootstrap() for _ in range(3)]
    out.update(boot_X=Xb, boot_y=yb, boot_yerr=eb, boot_u=np.stack([d[0] for d in draws]), boot_v=np.stack([d[1] for d in draws]),
               boot_yr=np.stack([d[2] for d in draws]), boot_er=np.stack([d[3] for d in draws]))
    x1d = rng.uniform(-3, 3, (9, 1))
    obj1 = tp.two_pcf(x1d, yb[:9], eb[:9], 0.1, 1.0, nbins=5)
    out.update(pad_X1=x1d, pad_X=obj1.X)
    np.savez(os.path.join(OUT, "g12_two_pcf_host.npz"), **out)
    # ---------------- G13: meanify run by the reference --------------------------------------------------
    mf = load_reference_meanify()
    rng = np.random.default_rng(2024)
    fields = []
    for _ in range(3):
        npt = 1500
        c = np.array([rng.uniform(0, 2048, npt), rng.uniform(0, 2048, npt)]).T
        c = c[(c[:, 0] - 600) ** 2 + (c[:, 1] - 1500) ** 2 > 250 ** 2]         # a hole: empty bins -> nan -> filtered
        pv = 0.02 + 5e-8 * (c[:, 0] - 1024) ** 2 + 5e-8 * (c[:, 1] - 1024) ** 2 + 0.01 * rng.standard_normal(len(c))
        fields.append((c, pv))
    out = {"nfields": len(fields)}
    for i, (c, pv) in enumerate(fields):
        out["coords%d" % i] = c
        out["params%d" % i] = pv
    for stat in ("mean", "median"):
        for tag, lim in (("auto", {}), ("lim", dict(lu_min=100.0, lu_max=1900.0, lv_min=0.0, lv_max=2048.0))):
            m = mf.meanify(bin_spacing=120.0, statistics=stat)
            for c, pv in fields:
                m.add_field(c, pv)
            m.meanify(**lim)
            key = stat + "_" + tag
            out.update({key + "_average": m._average, key + "_coords0": m.coords0, key + "_params0": m.params0, key + "_wrms0": m.wrms0,
                        key + "_xedge": m._xedge, key + "_yedge": m._yedge, key + "_u0": m._u0, key + "_v0": m._v0})
    np.savez(os.path.join(OUT, "g13_meanify.npz"), **out)
    make_g14(tg)
    print("golden vectors written to", OUT)


if __name__ == "__main__":
    main()

"""E/B decomposition of a vector field's 2-point correlation; mirrors ``treegp/utils.py:5-155``.

``vcorr`` accumulates the pair sums on the GPU (``tgp_vcorr``: all pairs, exact log binning on
the edges ``np.histogram`` would build) instead of materialising ``np.triu_indices`` index arrays
(4.5e8 pairs at the reference's ``maxpts`` = 30 000); the few O(bins) lines after it are NumPy as in
the reference.  ``comp_eb_treecorr`` keeps its name and signature: the reference runs
``treecorr.VVCorrelation`` there (``utils.py:108-134``); here the same kernel supplies xi+ / xi- on
TreeCorr's bin grid (nominal ``logr`` centres) with exact binning (``bin_slop = 0``).
"""
import numpy as np

from . import ops


def _log_edges(rmin, rmax, dlogr):
    logrmin = np.log(rmin)
    bins = int(np.ceil(np.log(rmax / rmin) / dlogr))
    # the uniform edges np.histogram(bins=bins, range=(lo, hi)) bins against (utils.py:52-55)
    return np.linspace(logrmin, logrmin + bins * dlogr, bins + 1), bins


def vcorr(x, y, dx, dy, rmin=5.0 / 3600.0, rmax=1.5, dlogr=0.05, maxpts=30000):
    """Angle-averaged 2-point correlation functions of the vector field (dx, dy) at (x, y), brute
    force over all pairs (utils.py:5-74).

    Returns logr (mean log separation per bin), xi_+ = <vx1 vx2 + vy1 vy2>, xi_- and xi_x (real and
    imaginary part of <v1 v2 conj(d)^2/|d|^2>), xi_z2 = <v1 v2> (complex)."""
    x, y, dx, dy = (np.asarray(t) for t in (x, y, dx, dy))
    if len(x) > maxpts:
        # subsample to about maxpts points (utils.py:28-35)
        rate = float(maxpts) / len(x)
        use = np.random.random(len(x)) <= rate
        x, y, dx, dy = x[use], y[use], dx[use], dy[use]
    edges, _ = _log_edges(rmin, rmax, dlogr)
    acc = ops.vcorr_sums(x, y, dx, dy, edges)
    counts = acc[0]
    logr = acc[1] / counts
    xiplus = acc[2] / counts
    xiz2 = (acc[3] + 1j * acc[4]) / counts
    ximinus = acc[5] / counts
    xicross = acc[6] / counts
    return logr, xiplus, ximinus, xicross, xiz2


def xiB(logr, xiplus, ximinus):
    """Estimate of the pure B-mode correlation function (utils.py:77-86):
    (xi+ - xi-)/2 + integral of xi- d(log r) from r outwards (central-difference bin widths, end bins 0)."""
    width = np.zeros_like(logr)
    width[1:-1] = (logr[2:] - logr[:-2]) * 0.5
    outward = np.flip(np.cumsum(np.flip(np.asarray(ximinus) * width)))
    return (xiplus - ximinus) * 0.5 + outward


def comp_eb(u, v, du, dv, **kwargs):
    """E/B decomposition of the correlation function of the vector field (du, dv) at (u, v)
    (utils.py:89-105).  Returns xie, xib, logr."""
    logr, xi_plus, xi_minus = vcorr(u, v, du, dv, **kwargs)[:3]
    b_mode = xiB(logr, xi_plus, xi_minus)
    return xi_plus - b_mode, b_mode, logr


class compEbTreecorr:
    """utils.py:108-134 with the pair loop on the GPU: xi+ / xi- on TreeCorr's log-bin grid
    (nbins = ceil(ln(rmax/rmin)/dlogr) bins of width dlogr from rmin, ``logr`` = nominal centres)."""

    def __init__(self, x, y, dx, dy, rmin=5.0 / 3600.0, rmax=1.5, dlogr=0.05):
        self._data = tuple(np.asarray(t, dtype=float) for t in (x, y, dx, dy))
        self._edges, self._bins = _log_edges(rmin, rmax, dlogr)
        self.logr = np.log(rmin) + (np.arange(self._bins) + 0.5) * dlogr
        self.xip = self.xim = None

    def vcorr(self):
        acc = ops.vcorr_sums(*self._data, self._edges)
        nz = acc[0] != 0
        safe = np.where(nz, acc[0], 1.0)
        self.npairs = acc[0]
        self.xip = np.where(nz, acc[2] / safe, 0.0)      # TreeCorr leaves empty bins at 0
        self.xim = np.where(nz, acc[5] / safe, 0.0)

    xiB = staticmethod(xiB)

    def comp_eb(self):
        self.vcorr()
        xib = self.xiB(self.logr, self.xip, self.xim)
        xie = self.xip - xib
        return xie, xib, self.logr


def comp_eb_treecorr(u, v, du, dv, **kwargs):
    """Same as comp_eb on TreeCorr's bin grid (utils.py:137-155).  Returns xie, xib, logr."""
    cebt = compEbTreecorr(u, v, du, dv, **kwargs)
    return cebt.comp_eb()

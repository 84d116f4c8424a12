"""``meanify``: build the mean function (binned spatial average of many fields) and write it as
the FITS table ``GPInterpolation(average_fits=...)`` reads.  Mirrors ``treegp/meanify.py:10-165``.

The binned statistic (``scipy.stats.binned_statistic_2d`` in the reference, :76-107) runs on the
GPU (``tgp_binned_stat_2d``); the O(bins) bookkeeping around it stays NumPy.  The file is
written by the minimal FITS writer of ``fits_io`` (fitsio is not a dependency), with the
reference's column names and shapes (:145-165).

One deliberate difference: in the reference the "weighted" statistic stops with a NameError
(``xedge`` / ``yedge`` are only bound in the other branch, :108-119); here it returns the weighted
mean and weighted rms that branch computes, on the same bin edges.
"""
import copy

import numpy as np

from . import ops
from .fits_io import write_bintable_row


class meanify(object):
    """Take data, build a spatial average, and write output average.

    :param bin_spacing: Bin size, resolution of the mean function. (default=120.)
    :param statistics:  "mean", "median" or "weighted". (default=mean)
    """

    def __init__(self, bin_spacing=120.0, statistics="mean"):
        self.bin_spacing = bin_spacing
        if statistics not in ["mean", "median", "weighted"]:
            raise ValueError("%s is not a suported statistic (only mean, weighted, and median are currently suported)"
                             % (statistics))
        self.stat_used = statistics
        self.coords = []
        self.params = []
        self.params_err = []

    def add_field(self, coord, param, params_err=None):
        """Add the (n, 2) coordinates and (n,) values (and errors, for "weighted") of one field."""
        if np.shape(coord)[1] != 2:
            raise ValueError("meanify is supported only in 2d for the moment.")
        self.coords.append(coord)
        self.params.append(param)
        if self.stat_used == "weighted":
            if params_err is None:
                raise ValueError("Need an associated error to params")
            self.params_err.append(params_err)

    def meanify(self, lu_min=None, lu_max=None, lv_min=None, lv_max=None):
        """Compute the mean function on a regular grid over the data (meanify.py:49-137)."""
        params = np.concatenate(self.params)
        coords = np.concatenate(self.coords, axis=0)
        params_err = np.concatenate(self.params_err) if self.stat_used == "weighted" else None

        if lu_min is None:
            lu_min = np.min(coords[:, 0])
        if lu_max is None:
            lu_max = np.max(coords[:, 0])
        if lv_min is None:
            lv_min = np.min(coords[:, 1])
        if lv_max is None:
            lv_max = np.max(coords[:, 1])

        nbin_u = int((lu_max - lu_min) / self.bin_spacing)
        nbin_v = int((lv_max - lv_min) / self.bin_spacing)
        xedge = np.linspace(lu_min, lu_max, nbin_u)
        yedge = np.linspace(lv_min, lv_max, nbin_v)

        average, wrms, _ = ops.binned_stat_2d(coords[:, 0], coords[:, 1], params, xedge, yedge,
                                              statistic=self.stat_used, err=params_err)
        average = average.T
        wrms = wrms.T
        self._average = copy.deepcopy(average)
        self._wrms = wrms
        average = average.reshape(-1)
        wrms = wrms.reshape(-1)
        keep = np.isfinite(average) & np.isfinite(wrms)

        # centre of each bin
        u0 = xedge[:-1] + (xedge[1] - xedge[0]) / 2.0
        v0 = yedge[:-1] + (yedge[1] - yedge[0]) / 2.0
        u0, v0 = np.meshgrid(u0, v0)
        self._u0 = u0
        self._v0 = v0
        self._xedge = xedge
        self._yedge = yedge
        coords0 = np.array([u0.reshape(-1), v0.reshape(-1)]).T

        # bins without data (nan) are dropped
        self.coords0 = coords0[keep]
        self.params0 = average[keep]
        self.wrms0 = wrms[keep]

    def save_results(self, name_output="mean_gp.fits"):
        """Write the mean function as a one-row binary table, extension "average_solution"."""
        write_bintable_row(name_output, {
            "COORDS0": self.coords0, "PARAMS0": self.params0, "WRMS0": self.wrms0,
            "_AVERAGE": self._average, "_WRMS": self._wrms, "_U0": self._u0, "_V0": self._v0,
        }, extname="average_solution")

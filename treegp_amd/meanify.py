"""``meanify``: the mean function of many exposures -- a binned spatial average of all their
stars -- and the FITS table ``GPInterpolation(average_fits=...)`` reads it from.

Public behaviour follows ``treegp/meanify.py:10-165`` of the reference (constructor arguments,
``add_field`` / ``meanify`` / ``save_results``, the attributes they leave behind, the column names
of the file).  The binned statistic itself (``scipy.stats.binned_statistic_2d`` there, :76-107)
is one call into the GPU library (``tgp_binned_stat_2d``, seam S6); the file is produced by this
package's minimal FITS writer instead of fitsio.

One deliberate difference: the reference's "weighted" statistic ends in a NameError (its edge
arrays are only bound in the other branch, :108-119).  Here it works, on the same grid, and yields
the weighted mean and weighted rms that branch computes.
"""
import numpy as np

from . import ops
from .fits_io import write_bintable_row

_SUPPORTED = ("mean", "median", "weighted")


def _grid_edges(lo, hi, spacing):
    """``int((hi - lo) / spacing)`` equally spaced edges from lo to hi (meanify.py:67-72)."""
    return np.linspace(lo, hi, int((hi - lo) / spacing))


class meanify(object):
    """Accumulate fields, then average them on a regular grid.

    :param bin_spacing: size of a grid cell, i.e. the resolution of the mean function. (default=120.)
    :param statistics:  "mean", "median" or "weighted" (inverse-variance weighted mean). (default=mean)
    """

    def __init__(self, bin_spacing=120.0, statistics="mean"):
        if statistics not in _SUPPORTED:
            raise ValueError("%s is not a suported statistic (only mean, weighted, and median are currently suported)"
                             % (statistics))
        self.bin_spacing = bin_spacing
        self.stat_used = statistics
        self.coords, self.params, self.params_err = [], [], []

    def add_field(self, coord, param, params_err=None):
        """One more field: coordinates (n, 2), values (n,), and their errors when weighting."""
        if np.shape(coord)[1] != 2:
            raise ValueError("meanify is supported only in 2d for the moment.")
        if self.stat_used == "weighted" and params_err is None:
            raise ValueError("Need an associated error to params")
        self.coords.append(coord)
        self.params.append(param)
        if self.stat_used == "weighted":
            self.params_err.append(params_err)

    def meanify(self, lu_min=None, lu_max=None, lv_min=None, lv_max=None):
        """Average everything added so far.  The grid spans the data unless limits are given.

        Leaves ``coords0`` / ``params0`` / ``wrms0`` (cells that hold data, flattened v-major) and
        the full grids ``_average`` / ``_wrms`` / ``_u0`` / ``_v0`` / ``_xedge`` / ``_yedge``."""
        uv = np.concatenate(self.coords, axis=0)
        values = np.concatenate(self.params)
        errors = np.concatenate(self.params_err) if self.stat_used == "weighted" else None
        u, v = uv[:, 0], uv[:, 1]
        limits = [(lu_min, np.min, u), (lu_max, np.max, u), (lv_min, np.min, v), (lv_max, np.max, v)]
        u_lo, u_hi, v_lo, v_hi = [given if given is not None else pick(axis) for given, pick, axis in limits]
        self._xedge = _grid_edges(u_lo, u_hi, self.bin_spacing)
        self._yedge = _grid_edges(v_lo, v_hi, self.bin_spacing)

        stat, rms, _ = ops.binned_stat_2d(u, v, values, self._xedge, self._yedge, statistic=self.stat_used, err=errors)
        # the library indexes [u cell][v cell]; the mean-function grid is stored v-major (meanify.py:105-106)
        self._average = np.array(stat.T)
        self._wrms = np.array(rms.T)

        half_u = (self._xedge[1] - self._xedge[0]) / 2.0
        half_v = (self._yedge[1] - self._yedge[0]) / 2.0
        self._u0, self._v0 = np.meshgrid(self._xedge[:-1] + half_u, self._yedge[:-1] + half_v)

        flat_avg, flat_rms = self._average.ravel(), self._wrms.ravel()
        has_data = np.isfinite(flat_avg) & np.isfinite(flat_rms)        # empty cells are nan
        centres = np.column_stack([self._u0.ravel(), self._v0.ravel()])
        self.coords0 = centres[has_data]
        self.params0 = flat_avg[has_data]
        self.wrms0 = flat_rms[has_data]

    def save_results(self, name_output="mean_gp.fits"):
        """Write the mean function: one table row, extension "average_solution" (meanify.py:139-165)."""
        columns = [("COORDS0", self.coords0), ("PARAMS0", self.params0), ("WRMS0", self.wrms0),
                   ("_AVERAGE", self._average), ("_WRMS", self._wrms), ("_U0", self._u0), ("_V0", self._v0)]
        write_bintable_row(name_output, dict(columns), extname="average_solution")

#!/usr/bin/env python3
"""cProfile of the anisotropic two-pcf fit at config 3's size (host side of the 444-resample bootstrap path)."""
import cProfile
import pstats
import sys

import numpy as np

sys.path.insert(0, __file__.rsplit("/tools/", 1)[0])
import treegp_amd as treegp  # noqa: E402
from treegp_amd.synthetic import star_field, headline_invlam  # noqa: E402

iL = headline_invlam()
X, y, ye, Xs = star_field(32768, 32768)
kern = "1.0**2 * AnisotropicVonKarman(invLam=array(%s))" % np.array2string(iL, separator=",", precision=17)


def run():
    gp = treegp.GPInterpolation(kernel=kern, optimizer="anisotropic", nbins=21, min_sep=0.0, max_sep=0.15, normalize=True)
    gp.initialize(X, y, y_err=ye)
    gp.solve()


run()
pr = cProfile.Profile()
pr.enable()
run()
pr.disable()
pstats.Stats(pr).sort_stats("cumulative").print_stats(28)

#!/usr/bin/env python3
"""cProfile of the isotropic two-pcf fit at config 3's size (N = 32768, von Karman)."""
import cProfile
import pstats
import sys

sys.path.insert(0, __file__.rsplit("/tools/", 1)[0])
import treegp_amd as treegp  # noqa: E402
from treegp_amd.synthetic import star_field  # noqa: E402

X, y, ye, Xs = star_field(32768, 16)


def run():
    gp = treegp.GPInterpolation(kernel="1.0**2 * VonKarman(length_scale=0.1)", optimizer="two-pcf", nbins=20, normalize=True)
    gp.initialize(X, y, y_err=ye)
    gp.solve()


run()
pr = cProfile.Profile()
pr.enable()
run()
pr.disable()
pstats.Stats(pr).sort_stats("cumulative").print_stats(22)

#!/usr/bin/env python3
"""cProfile of one maximum-likelihood fit (host side).  usage: profile_ml_fit.py [N=600] [parallel=1]"""
import cProfile
import os
import pstats
import sys

import numpy as np

sys.path.insert(0, __file__.rsplit("/tools/", 1)[0])
import treegp_amd as treegp  # noqa: E402
from treegp_amd.synthetic import star_field, headline_invlam  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 600
os.environ["TGP_ML_PARALLEL"] = sys.argv[2] if len(sys.argv) > 2 else "1"
iL = headline_invlam()
kern = "1.0**2 * AnisotropicRBF(invLam=array(%s))" % np.array2string(iL * 1.3, separator=",", precision=17)
X, y, ye, _ = star_field(n, 16)


def run():
    gp = treegp.GPInterpolation(kernel=kern, optimizer="log-likelihood", normalize=True)
    gp.initialize(X, y, y_err=ye)
    gp.solve()


run()
pr = cProfile.Profile()
pr.enable()
run()
pr.disable()
pstats.Stats(pr).sort_stats("cumulative").print_stats(30)

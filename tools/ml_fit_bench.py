#!/usr/bin/env python3
"""Wall-clock of a maximum-likelihood fit (optimizer="log-likelihood") at a few sizes, with the concurrent
finite-difference evaluations on and off.  usage: ml_fit_bench.py [N ...]"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, __file__.rsplit("/tools/", 1)[0])
import treegp_amd as treegp  # noqa: E402
from treegp_amd.synthetic import star_field, headline_invlam  # noqa: E402

sizes = [int(a) for a in sys.argv[1:]] or [600, 2048, 4096, 8192]
iL = headline_invlam()
kern = "1.0**2 * AnisotropicRBF(invLam=array(%s))" % np.array2string(iL * 1.3, separator=",", precision=17)
for n in sizes:
    X, y, ye, _ = star_field(n, 16)
    for par in ("1", "0"):
        os.environ["TGP_ML_PARALLEL"] = par
        best = None
        for rep in range(2):
            gp = treegp.GPInterpolation(kernel=kern, optimizer="log-likelihood", normalize=True)
            gp.initialize(X, y, y_err=ye)
            t0 = time.perf_counter()
            gp.solve()
            dt = time.perf_counter() - t0
            best = dt if best is None else min(best, dt)
        print("N=%5d  concurrent FD %s: fit %.1f ms, logL %.6f" % (n, par, best * 1e3, gp._optimizer._logL), flush=True)

#!/usr/bin/env python3
"""Final log-likelihood of maximum-likelihood fits on a set of seeded problems (the finite-difference L-BFGS-B path is
sensitive to rounding in the likelihood: this compares how two builds / switches fare).  usage: ml_fit_paths.py [N]"""
import sys
import time

import numpy as np

sys.path.insert(0, __file__.rsplit("/tools/", 1)[0])
import treegp_amd as treegp  # noqa: E402
from treegp_amd.synthetic import star_field, headline_invlam  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
iL = headline_invlam()
out = []
for seed in range(8):
    X, y, ye, _ = star_field(n, 16, seed=seed)
    scale = 1.0 + 0.15 * seed
    kern = "1.0**2 * AnisotropicRBF(invLam=array(%s))" % np.array2string(iL * scale, separator=",", precision=17)
    gp = treegp.GPInterpolation(kernel=kern, optimizer="log-likelihood", normalize=True)
    gp.initialize(X, y, y_err=ye)
    t0 = time.perf_counter()
    gp.solve()
    out.append((gp._optimizer._logL, time.perf_counter() - t0))
print("N=%d  final logL: %s" % (n, " ".join("%.3f" % v for v, _ in out)))
print("N=%d  fit time ms: %s   total %.0f ms" % (n, " ".join("%.0f" % (t * 1e3) for _, t in out), 1e3 * sum(t for _, t in out)))

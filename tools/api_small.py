#!/usr/bin/env python3
"""Wall-clock of the public API at config 1's size (1-D, N = 512, M = 1024) and its host-side profile."""
import cProfile
import pstats
import sys
import time

import numpy as np

sys.path.insert(0, __file__.rsplit("/tools/", 1)[0])
import treegp_amd as treegp  # noqa: E402

rng = np.random.default_rng(1)
X = rng.uniform(-10, 10, (512, 1)); y = np.sin(X[:, 0]) + 0.1 * rng.standard_normal(512); ye = np.full(512, 0.1)
Xs = np.linspace(-10, 10, 1024)[:, None]
kern = "1.0**2 * AnisotropicRBF(scale_length=[2.0])"


def run(opt):
    gp = treegp.GPInterpolation(kernel=kern, optimizer=opt, normalize=True)
    gp.initialize(X, y, y_err=ye)
    gp.solve()
    return gp.predict(Xs, return_cov=False)


for opt in ("none", "log-likelihood"):
    run(opt)
    t0 = time.perf_counter()
    for _ in range(10):
        run(opt)
    print("optimizer=%-14s construct + initialize + solve + predict(1024): %.2f ms" % (opt, (time.perf_counter() - t0) / 10 * 1e3), flush=True)
pr = cProfile.Profile()
pr.enable()
for _ in range(10):
    run("none")
pr.disable()
pstats.Stats(pr).sort_stats("cumulative").print_stats(18)

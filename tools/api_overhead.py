#!/usr/bin/env python3
"""Where the API route (GPInterpolation.predict from host arrays) spends its time beyond the device phases, at the headline size."""
import cProfile
import pstats
import sys
import time

sys.path.insert(0, __file__.rsplit("/tools/", 1)[0])
import treegp_amd  # noqa: E402
from treegp_amd import _lib  # noqa: E402
from treegp_amd.synthetic import star_field, headline_kernel_string  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
if len(sys.argv) > 2 and sys.argv[2] == "profiling":      # bench.py's mode: per-launch events in the factorisation
    _lib.load_library().tgp_set_profiling(_lib.get_ctx(), 1)
X, y, ye, Xs = star_field(n, 4 * n)
for rep in range(3):
    gp = treegp_amd.GPInterpolation(kernel=headline_kernel_string(), optimizer="none", normalize=True, white_noise=0.0)
    t0 = time.perf_counter()
    gp.initialize(X, y, y_err=ye)
    t1 = time.perf_counter()
    if rep == 2:
        pr = cProfile.Profile()
        pr.enable()
    yp = gp.predict(Xs)
    if rep == 2:
        pr.disable()
    t2 = time.perf_counter()
    tm = _lib.timings(_lib.get_ctx())
    print("rep %d: initialize %.1f ms, predict %.1f ms wall; last call's device phases %s" % (rep, (t1 - t0) * 1e3, (t2 - t1) * 1e3, ["%.2f" % v for v in tm]), flush=True)
pstats.Stats(pr).sort_stats("cumulative").print_stats(18)

#!/usr/bin/env python3
"""Likelihood-only evaluation (augmented row, padding-block skip) against the full solve at sizes that are not
multiples of 256, up to the groups-of-four schedule.  usage: check_large_odd.py [N ...]"""
import sys
import time

import numpy as np

sys.path.insert(0, __file__.rsplit("/tools/", 1)[0])
from treegp_amd import _lib, ops  # noqa: E402
from treegp_amd.synthetic import star_field, headline_invlam  # noqa: E402

iL = headline_invlam()
spec = ops.KernelSpec(_lib.TGP_ARBF, amp=1.0, a=iL[0, 0], b=iL[0, 1], c=iL[1, 1])
for n in [int(a) for a in sys.argv[1:]] or [9001, 23000, 30000, 33100]:
    X, y, ye, _ = star_field(n, 16)
    y = y - y.mean()
    t0 = time.perf_counter(); _, ld, ya, _ = ops.gp_solve(spec, X, y, ye); t1 = time.perf_counter()
    _, ld2, c2, _ = ops.gp_solve(spec, X, y, ye, want_alpha=False); t2 = time.perf_counter()
    print("N=%d  full solve %.1f ms, likelihood only %.1f ms; rel diff chi2 %.2e logdet %.2e" %
          (n, (t1 - t0) * 1e3, (t2 - t1) * 1e3, abs(c2 - ya) / abs(ya), abs(ld2 - ld) / abs(ld)), flush=True)

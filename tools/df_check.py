#!/usr/bin/env python3
"""Cholesky time (best of 5) and the solve's residual ||(K + D) alpha - y|| / ||y|| at a list of sizes, for A/B runs of
the schedules (TGP_CHOL_MODE, TGP_DF_T are read once per process): TGP_CHOL_MODE=4 python tools/df_check.py 4096 8192"""
import sys
import numpy as np
sys.path.insert(0, __file__.rsplit("/tools/", 1)[0])
from treegp_amd import _lib, ops
from treegp_amd.synthetic import star_field, headline_invlam
iL = headline_invlam()
spec = ops.KernelSpec(_lib.TGP_ARBF, amp=1.0, a=iL[0, 0], b=iL[0, 1], c=iL[1, 1])
for n in [int(v) for v in sys.argv[1:]] or (4096, 8192, 16384):
    X, y, ye, _ = star_field(n, 16)
    y = y - y.mean()
    best, lik = 1e9, 1e9
    for it in range(5):
        alpha, logdet, ydota, _ = ops.gp_solve(spec, X, y, ye)
        tm = _lib.timings(_lib.get_ctx())
        best = min(best, tm[1])
    Ka = ops.gp_predict(spec, X, alpha, X)
    res = np.linalg.norm(Ka + ye ** 2 * alpha - y) / np.linalg.norm(y)
    print("%6d chol %.3f ms  %.2f TF  trsv %.3f ms  residual %.2e  logdet %.10e" % (n, best, n ** 3 / 3 / best / 1e9, tm[2], res, logdet), flush=True)

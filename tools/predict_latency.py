#!/usr/bin/env python3
"""Wall-clock of tgp_gp_predict (host boundary) over a grid of small (N, M).  Development aid."""
import sys
import time

import numpy as np

sys.path.insert(0, __file__.rsplit("/tools/", 1)[0])
from treegp_amd import _lib, ops  # noqa: E402
from treegp_amd.synthetic import star_field, headline_invlam  # noqa: E402

iL = headline_invlam()
spec = ops.KernelSpec(_lib.TGP_ARBF, amp=1.0, a=iL[0, 0], b=iL[0, 1], c=iL[1, 1])
ctx = _lib.get_ctx()
for n in (256, 512, 1024, 2048):
    X, y, ye, Xs = star_field(n, 16384)
    a = ops.gp_solve(spec, X, y - y.mean(), ye)[0]
    row = []
    for m in (512, 1024, 2048, 4096, 8192, 16384):
        ops.gp_predict(spec, X, a, Xs[:m])
        t0 = time.perf_counter()
        for _ in range(20):
            ops.gp_predict(spec, X, a, Xs[:m])
        row.append("%d:%.3f" % (m, (time.perf_counter() - t0) / 20 * 1e3))
    t0 = time.perf_counter()
    for _ in range(20):
        ops.gp_solve(spec, X, y - y.mean(), ye)
    print("N=%5d  predict ms by M  %s   | solve with alpha %.3f ms" % (n, "  ".join(row), (time.perf_counter() - t0) / 20 * 1e3), flush=True)

print("alternating solve(with alpha) + predict, ms per pair:")
for n in (256, 512, 1024):
    X, y, ye, Xs = star_field(n, 16384)
    yc = y - y.mean()
    row = []
    for m in (512, 1024, 2048, 4096, 8192):
        for _ in range(3):
            a = ops.gp_solve(spec, X, yc, ye)[0]
            ops.gp_predict(spec, X, a, Xs[:m])
        t0 = time.perf_counter()
        for _ in range(20):
            a = ops.gp_solve(spec, X, yc, ye)[0]
            ops.gp_predict(spec, X, a, Xs[:m])
        row.append("%d:%.3f" % (m, (time.perf_counter() - t0) / 20 * 1e3))
    print("N=%5d  %s" % (n, "  ".join(row)), flush=True)

#!/usr/bin/env python3
"""Throughput of K concurrent likelihood evaluations at one size, contexts with one stream each (tgp_set_lookahead 0: what the
ML fit's concurrent finite differences use) against contexts that keep their look-ahead stream.  usage: concurrency_modes.py [N=8192]"""
import sys
import time
from concurrent.futures import ThreadPoolExecutor

sys.path.insert(0, __file__.rsplit("/tools/", 1)[0])
from treegp_amd import _lib, ops  # noqa: E402
from treegp_amd.synthetic import star_field, headline_invlam  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
iL = headline_invlam()
spec = ops.KernelSpec(_lib.TGP_ARBF, amp=1.0, a=iL[0, 0], b=iL[0, 1], c=iL[1, 1])
X, y, ye, _ = star_field(n, 16)
prob = ops.ResidentProblem(X, y - y.mean(), ye)
lib = _lib.load_library()
reps = 8
for look in (0, 1):
    ctxs = [_lib.new_ctx(0) for _ in range(5)]
    for c in ctxs:
        lib.tgp_set_lookahead(c, look)

    def loop(i):
        for _ in range(reps):
            ops.gp_solve_resident(spec, prob, ctx=ctxs[i])

    for K in (1, 2, 3, 4, 5):
        with ThreadPoolExecutor(K) as pool:
            list(pool.map(loop, range(K)))
            t0 = time.perf_counter()
            list(pool.map(loop, range(K)))
            dt = time.perf_counter() - t0
        print("N=%d lookahead=%d K=%d: %.3f ms per evaluation overall (%.0f evaluations/s)" % (n, look, K, dt / (K * reps) * 1e3, K * reps / dt), flush=True)
    for c in ctxs:
        lib.tgp_destroy(c)

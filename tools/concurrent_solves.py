#!/usr/bin/env python3
"""K concurrent host-boundary solves (one context / stream each, as the ML fit issues them) checked against the same
solve done alone.  usage: concurrent_solves.py [N=4096] [K=4]"""
import sys
import time
from concurrent.futures import ThreadPoolExecutor

import numpy as np

sys.path.insert(0, __file__.rsplit("/tools/", 1)[0])
from treegp_amd import _lib, ops  # noqa: E402
from treegp_amd.synthetic import star_field, headline_invlam  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
K = int(sys.argv[2]) if len(sys.argv) > 2 else 4
iL = headline_invlam()
specs = [ops.KernelSpec(_lib.TGP_ARBF, amp=1.0, a=iL[0, 0] * (1 + 0.01 * i), b=iL[0, 1], c=iL[1, 1]) for i in range(K)]
X, y, ye, _ = star_field(n, 16)
y = y - y.mean()
ref = [ops.gp_solve(s, X, y, ye, want_alpha=False)[1:3] for s in specs]
ctxs = [_lib.new_ctx(0) for _ in range(K)]


def one(i):
    try:
        return ops.gp_solve(specs[i], X, y, ye, want_alpha=False, ctx=ctxs[i])[1:3]
    except Exception as e:  # noqa: BLE001
        return repr(e)


with ThreadPoolExecutor(K) as pool:
    for rep in range(3):
        t0 = time.perf_counter()
        got = list(pool.map(one, range(K)))
        dt = time.perf_counter() - t0
        bad = [(i, g, r) for i, (g, r) in enumerate(zip(got, ref)) if isinstance(g, str) or not np.allclose(g, r, rtol=1e-9)]
        print("round %d: %d concurrent solves in %.2f ms, mismatches: %s" % (rep, K, dt * 1e3, bad if bad else "none"), flush=True)

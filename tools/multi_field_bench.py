#!/usr/bin/env python3
"""Several fields at shared positions (README.rst:28): k independent solves, as a treegp user writes them today, against one
factorisation + tgp_factor_solve; and the dense route (caller-evaluated K) against the parametrised one.
usage: multi_field_bench.py [N=16384] [k=8]"""
import sys
import time

import numpy as np

sys.path.insert(0, __file__.rsplit("/tools/", 1)[0])
from treegp_amd import _lib, ops  # noqa: E402
from treegp_amd.synthetic import star_field, headline_invlam  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 16384
k = int(sys.argv[2]) if len(sys.argv) > 2 else 8
X, y, y_err, _ = star_field(n, 8)
iL = headline_invlam()
spec = ops.KernelSpec(_lib.TGP_ARBF, amp=1.0, a=iL[0, 0], b=iL[0, 1], c=iL[1, 1])
rng = np.random.default_rng(0)
Y = np.stack([y - y.mean()] + [rng.standard_normal(n) for _ in range(k - 1)])
ops.gp_solve(spec, X, Y[0], y_err)                                     # warm-up
t0 = time.perf_counter()
ref = np.stack([ops.gp_solve(spec, X, Y[v], y_err)[0] for v in range(k)])
t_indep = time.perf_counter() - t0
t0 = time.perf_counter()
_, _, _, f = ops.gp_solve(spec, X, Y[0], y_err, keep=True, want_alpha=False)
t_factor = time.perf_counter() - t0
t0 = time.perf_counter()
got = ops.factor_solve(f, Y)
t_solve = time.perf_counter() - t0
dev_ms = _lib.timings(f._ctx)[2]
t0 = time.perf_counter()
ops.factor_solve(f, Y)
t_solve2 = time.perf_counter() - t0
f.free()
print("N=%d, %d fields: %d independent solves %.1f ms; one factorisation %.1f ms + factor_solve %.1f ms (device %.2f ms incl. the slab "
      "build; second call %.1f ms); max rel diff %.1e" % (n, k, k, t_indep * 1e3, t_factor * 1e3, t_solve * 1e3, dev_ms, t_solve2 * 1e3,
                                                          np.abs(got - ref).max() / np.abs(ref).max()))
if n <= 16384:
    K = ops.kernel_matrix(spec, X)                 # a caller-evaluated matrix (here: the library's own S1 output)
    ops.gp_solve_dense(K, Y[0], y_err)
    t0 = time.perf_counter()
    a = ops.gp_solve_dense(K, Y[0], y_err)[0]
    t_dense = time.perf_counter() - t0
    tm = _lib.timings(_lib.get_ctx())
    print("dense route: %.1f ms wall for an (n, n) host matrix (pack %.2f ms, Cholesky %.2f ms, solves %.2f ms on the device; the rest is "
          "the %.0f MB upload); max rel diff vs parametrised %.1e" % (t_dense * 1e3, tm[0], tm[1], tm[2], K.nbytes / 1e6,
                                                                      np.abs(a - ref[0]).max() / np.abs(ref[0]).max()))

#!/usr/bin/env python3
"""Wall-clock latency of one host-boundary solve (tgp_gp_solve: what each ML-fit likelihood
evaluation costs) at the small sizes treegp is typically used at.  Development aid."""
import sys
import time

import numpy as np

sys.path.insert(0, __file__.rsplit("/tools/", 1)[0])
from treegp_amd import _lib, ops  # noqa: E402
from treegp_amd.synthetic import star_field, headline_invlam  # noqa: E402


def main():
    sizes = [int(s) for s in sys.argv[1:]] or [256, 512, 1024, 2048, 4096, 8192]
    ctx = _lib.get_ctx()
    iL = headline_invlam()
    spec = ops.KernelSpec(_lib.TGP_ARBF, amp=1.0, a=iL[0, 0], b=iL[0, 1], c=iL[1, 1])
    for n in sizes:
        X, y, ye, Xs = star_field(n, 4 * n)
        y = y - y.mean()
        ops.gp_solve(spec, X, y, ye, want_alpha=False)
        reps = 20
        t0 = time.perf_counter()
        for _ in range(reps):
            ops.gp_solve(spec, X, y, ye, want_alpha=False)
        dt = (time.perf_counter() - t0) / reps
        tm = _lib.timings(ctx)
        prob = ops.ResidentProblem(X, y, ye)
        ops.gp_solve_resident(spec, prob)
        t0 = time.perf_counter()
        for _ in range(reps):
            ops.gp_solve_resident(spec, prob)
        dtr = (time.perf_counter() - t0) / reps
        prob.close()
        a = ops.gp_solve(spec, X, y, ye)[0]          # warm-up: first use of a size grows the library's scratch buffers
        ops.gp_predict(spec, X, a, Xs)
        t0 = time.perf_counter()
        for _ in range(reps):
            a = ops.gp_solve(spec, X, y, ye)[0]
            ops.gp_predict(spec, X, a, Xs)
        dt2 = (time.perf_counter() - t0) / reps
        print("N=%5d  likelihood eval %.3f ms (%.0f solves/s; device: kbuild %.3f chol %.3f trsv %.3f ms), data resident %.3f ms (%.0f solves/s)   solve+predict(4N) %.3f ms"
              % (n, dt * 1e3, 1.0 / dt, tm[0], tm[1], tm[2], dtr * 1e3, 1.0 / dtr, dt2 * 1e3), flush=True)


if __name__ == "__main__":
    main()

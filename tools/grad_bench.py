#!/usr/bin/env python3
"""Cost of the analytic likelihood gradient (tgp_gp_loglik_grad) beside the solve it belongs to, and a whole maximum-likelihood
fit with it against the finite-difference fit (serial and side by side).  One JSON line per size / fit."""
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import treegp_amd as treegp
from treegp_amd import _lib, ops
from treegp_amd.kernels import kernel_to_spec

KERN = "1.0**2 * AnisotropicRBF(invLam=array([[120., 30.], [30., 90.]]))"


def problem(n, seed=3):
    rng = np.random.default_rng(seed)
    X = rng.uniform(0, 1, (n, 2))
    y = np.sin(6 * X[:, 0]) * np.cos(4 * X[:, 1]) + 0.05 * rng.standard_normal(n)
    return X, y, 0.05 * rng.uniform(0.8, 1.2, n)


def main():
    lib, ctx = _lib.load_library(), _lib.get_ctx()
    spec = kernel_to_spec(treegp.eval_kernel(KERN))
    for n in [int(v) for v in os.environ.get("SIZES", "600,2048,4096,8192,16384,32768").split(",")]:
        X, y, e = problem(n)
        best_s, best_g, best_gd = 1e9, 1e9, 1e9
        for _ in range(4):
            t0 = time.perf_counter()
            alpha, _, _, fac = ops.gp_solve(spec, X, y, e, keep=True)
            t1 = time.perf_counter()
            ops.gp_loglik_grad(spec, fac, X, alpha)
            t2 = time.perf_counter()
            ms = _lib.timings(ctx)
            fac.free()
            best_s, best_g, best_gd = min(best_s, t1 - t0), min(best_g, t2 - t1), min(best_gd, ms[3] * 1e-3)
        flops = 2.0 / 3.0 * n ** 3
        print(json.dumps({"n": n, "solve_ms": round(best_s * 1e3, 3), "grad_ms": round(best_g * 1e3, 3),
                          "grad_device_ms": round(best_gd * 1e3, 3), "grad_over_solve": round(best_g / best_s, 2),
                          "grad_tflops_of_2n3_3": round(flops / best_gd / 1e12, 2)}), flush=True)
    for n in [int(v) for v in os.environ.get("FIT_SIZES", "600,2048,4096").split(",")]:
        X, y, e = problem(n, seed=11)
        out = {"fit_n": n}
        for mode, env in (("fd_serial", {"TGP_ML_PARALLEL": "0", "TGP_ML_GRADIENT": "fd"}),
                          ("fd_side_by_side", {"TGP_ML_PARALLEL": "1", "TGP_ML_GRADIENT": "fd"}),
                          ("analytic", {"TGP_ML_GRADIENT": "analytic"})):
            os.environ.update(env)
            best = 1e9
            for _ in range(3):
                gp = treegp.GPInterpolation(kernel="0.7**2 * AnisotropicRBF(invLam=array([[60., 0.], [0., 60.]]))",
                                            optimizer="log-likelihood", normalize=True)
                gp.initialize(X, y, y_err=e)
                t0 = time.perf_counter()
                gp.solve()
                best = min(best, time.perf_counter() - t0)
            out[mode + "_ms"] = round(best * 1e3, 1)
            out[mode + "_logL"] = round(float(gp._optimizer._logL), 6)
        print(json.dumps(out), flush=True)


if __name__ == "__main__":
    main()

#!/usr/bin/env python3
"""Quick phase timings of the device-resident GP solve/predict (development aid)."""
import ctypes as C
import sys
import time

import numpy as np

sys.path.insert(0, __file__.rsplit("/tools/", 1)[0])
from treegp_amd import _lib, ops  # noqa: E402


def main():
    sizes = [int(s) for s in sys.argv[1:]] or [8192, 16384]
    lib = _lib.load_library()
    ctx = _lib.get_ctx()
    lib.tgp_set_profiling(ctx, 1)
    for n in sizes:
        m = 4 * n
        rng = np.random.default_rng(20240613)
        X = rng.uniform(0, 1, (n, 2))
        y = np.sin(7 * X[:, 0]) * np.cos(5 * X[:, 1]) + 0.03 * rng.standard_normal(n)
        yerr = 0.03 * rng.uniform(0.8, 1.2, n)
        Xs = rng.uniform(0, 1, (m, 2))
        e = (1 - 0.2236) / (1 + 0.2236)
        from numpy.linalg import inv
        phi = 0.5 * np.arctan2(0.1, 0.2)
        rot = np.array([[np.cos(phi), np.sin(phi)], [-np.sin(phi), np.cos(phi)]])
        iL = inv(rot.T @ np.diag([0.05 ** 2, (0.05 * e) ** 2]) @ rot)
        spec = ops.KernelSpec(_lib.TGP_ARBF, amp=1.0, a=iL[0, 0], b=iL[0, 1], c=iL[1, 1])
        dX = ops.DeviceBuffer.from_array(ctx, X); dy = ops.DeviceBuffer.from_array(ctx, y)
        de = ops.DeviceBuffer.from_array(ctx, yerr); dXs = ops.DeviceBuffer.from_array(ctx, Xs)
        da = ops.DeviceBuffer(ctx, n * 8); dys = ops.DeviceBuffer(ctx, m * 8)
        ld, yd = C.c_double(), C.c_double()
        for it in range(2):
            t0 = time.perf_counter()
            rc = lib.tgp_d_gp_solve(ctx, C.byref(spec.to_c()), dX.ptr, n, dy.ptr, de.ptr, da.ptr, C.byref(ld), C.byref(yd), None)
            t1 = time.perf_counter()
            assert rc == 0 or int(__import__('os').environ.get('TGP_SYRK_VARIANT', '0')) >= 10, rc
            tm = _lib.timings(ctx)
            rc = lib.tgp_d_gp_predict(ctx, C.byref(spec.to_c()), dX.ptr, n, da.ptr, dXs.ptr, m, dys.ptr)
            t2 = time.perf_counter()
            tp = _lib.timings(ctx)[3]
            chol_tf = (n ** 3 / 3) / (tm[1] * 1e-3) / 1e12
            syrk_tf = tm[7] / (tm[5] * 1e-3) / 1e12 if tm[5] > 0 else 0
            print("N=%d it%d solve %.1f ms (kbuild %.2f ms = %.0f GB/s, chol %.1f ms = %.2f TF, syrk kernels %.1f ms = %.2f TF over %d launches, trsv %.2f ms) "
                  "predict M=%d %.2f ms = %.2e pairs/s  logdet %.6f" %
                  (n, it, (t1 - t0) * 1e3, tm[0], tm[8] / tm[0] / 1e6, tm[1], chol_tf, tm[5], syrk_tf, int(tm[6]), tm[2], m, tp, n * m / (tp * 1e-3), ld.value), flush=True)
        for b in (dX, dy, de, dXs, da, dys):
            b.free()


if __name__ == "__main__":
    main()

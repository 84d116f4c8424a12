#!/usr/bin/env python3
"""The dataflow factorisation (TGP_CHOL_DATAFLOW=1, csrc/pchol.hip) against the launch schedule on the same problems: alpha and
log-determinant of both, the relative residual of the dataflow solve, Cholesky times (best of 4).  usage: dataflow_check.py [N ...]"""
import os
import subprocess
import sys

import numpy as np

sys.path.insert(0, __file__.rsplit("/tools/", 1)[0])


def run(n):
    from treegp_amd import _lib, ops
    from treegp_amd.synthetic import star_field, headline_invlam
    iL = headline_invlam()
    spec = ops.KernelSpec(_lib.TGP_ARBF, amp=1.0, a=iL[0, 0], b=iL[0, 1], c=iL[1, 1])
    X, y, ye, _ = star_field(n, 16)
    y = y - y.mean()
    best = 1e9
    for _ in range(4):
        alpha, logdet, ydota, _f = ops.gp_solve(spec, X, y, ye)
        best = min(best, _lib.timings(_lib.get_ctx())[1])
    r = ops.gp_predict(spec, X, alpha, X) + ye ** 2 * alpha - y
    return alpha, logdet, best, float(np.linalg.norm(r) / np.linalg.norm(y))


if __name__ == "__main__":
    if len(sys.argv) > 2 and sys.argv[1] == "--child":
        n = int(sys.argv[2])
        alpha, logdet, ms, res = run(n)
        np.save(sys.argv[3], np.concatenate([alpha, [logdet, ms, res]]))
        sys.exit(0)
    for n in [int(v) for v in sys.argv[1:]] or [2048, 4096, 8192]:
        out = {}
        for mode in ("0", "1"):
            f = "/tmp/_dfc_%s.npy" % mode
            r = subprocess.run([sys.executable, __file__, "--child", str(n), f], env=dict(os.environ, TGP_CHOL_DATAFLOW=mode),
                               capture_output=True, text=True, timeout=300)
            if r.returncode != 0:
                print("N=%d TGP_CHOL_DATAFLOW=%s FAILED: %s" % (n, mode, r.stderr[-600:]), flush=True)
                out = None
                break
            out[mode] = np.load(f)
        if out is None:
            continue
        a0, a1 = out["0"], out["1"]
        da = np.abs(a1[:-3] - a0[:-3]).max() / np.abs(a0[:-3]).max()
        print("N=%5d: launches %.3f ms, dataflow %.3f ms (%.2fx); alpha differs by %.2e of its scale, logdet by %.2e relative; "
              "residual of the dataflow solve %.2e" % (n, a0[-2], a1[-2], a0[-2] / a1[-2], da, abs(a1[-3] - a0[-3]) / abs(a0[-3]), a1[-1]), flush=True)

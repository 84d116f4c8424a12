#!/usr/bin/env python3
"""Triangular sweeps on one factor: the 128-block chain (trsv.hip) against the big-step sweeps with inverse slabs
(trsv_big.hip), same right-hand side; prints times, 8 Np^2-byte rates and the difference of the two solutions.
usage: trsv_bench.py [N ...]      env TGP_POTRS_STEP=512|1024"""
import ctypes as C
import json
import os
import sys

import numpy as np

sys.path.insert(0, __file__.rsplit("/tools/", 1)[0])
from treegp_amd import _lib, ops  # noqa: E402
from treegp_amd.synthetic import star_field, headline_invlam  # noqa: E402

lib = _lib.load_library()
ctx = _lib.get_ctx()


def run(n, reps=3):
    X, y, y_err, _ = star_field(n, 16)
    iL = headline_invlam()
    spec = ops.KernelSpec(_lib.TGP_ARBF, amp=1.0, a=iL[0, 0], b=iL[0, 1], c=iL[1, 1])
    Np = lib.tgp_padded_n(n)
    dX = ops.DeviceBuffer.from_array(ctx, X); de = ops.DeviceBuffer.from_array(ctx, y_err)
    dA = ops.DeviceBuffer(ctx, lib.tgp_panel_elems(Np) * 8)
    dW = ops.DeviceBuffer(ctx, Np * 128 * 8)
    _lib.check(ctx, lib.tgp_d_kbuild_lower(ctx, C.byref(spec.to_c()), dX.ptr, n, de.ptr, dA.ptr), "kbuild")
    assert lib.tgp_d_potrf(ctx, dA.ptr, Np, dW.ptr) == 0
    rhs = np.zeros(Np); rhs[:n] = y - y.mean()
    res = {}
    for tag, env in (("chain128", {"TGP_POTRS_BIG_FROM": "0"}), ("big", {"TGP_POTRS_BIG_FROM": "256"})):
        os.environ.update(env)
        best = 1e9
        for _ in range(reps):
            db = ops.DeviceBuffer.from_array(ctx, rhs)
            _lib.check(ctx, lib.tgp_d_potrs(ctx, dA.ptr, dW.ptr, Np, db.ptr), "potrs")
            best = min(best, _lib.timings(ctx)[2])
            sol = db.to_array(Np)
            db.free()
        res[tag] = (best, sol)
    os.environ.pop("TGP_POTRS_BIG_FROM", None)
    a, b = res["chain128"][1], res["big"][1]
    bytes_alg = 8.0 * Np * Np
    print(json.dumps(dict(n=n, step=int(os.environ.get("TGP_POTRS_STEP", "512" if Np < 12288 else "1024")), chain128_ms=res["chain128"][0], big_ms=res["big"][0],
                          chain128_GBps=bytes_alg / res["chain128"][0] / 1e6, big_GBps=bytes_alg / res["big"][0] / 1e6,
                          big_frac_hbm=bytes_alg / res["big"][0] / 1e6 / 8000, max_rel_diff=float(np.abs(a - b).max() / np.abs(a).max()))),
          flush=True)
    for buf in (dX, de, dA, dW):
        buf.free()


if __name__ == "__main__":
    for n in [int(a) for a in sys.argv[1:]] or [1000, 2048, 2300, 8192, 32768, 65536]:
        run(n)

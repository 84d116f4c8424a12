#!/usr/bin/env python3
"""Task trace of one dataflow factorisation (TGP_CHOL_DATAFLOW=1 TGP_PCHOL_TRACE=1): per task type count / busy time, the span,
worker utilisation, and per panel when its chain tasks ran.  usage: dataflow_trace.py [N=8192] [panels to list=40]"""
import ctypes as C
import os
import sys

import numpy as np

os.environ["TGP_CHOL_DATAFLOW"] = "1"
os.environ["TGP_PCHOL_TRACE"] = "1"
sys.path.insert(0, __file__.rsplit("/tools/", 1)[0])
from treegp_amd import _lib, ops  # noqa: E402
from treegp_amd.synthetic import star_field, headline_invlam  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
npan = int(sys.argv[2]) if len(sys.argv) > 2 else 40
iL = headline_invlam()
spec = ops.KernelSpec(_lib.TGP_ARBF, amp=1.0, a=iL[0, 0], b=iL[0, 1], c=iL[1, 1])
X, y, ye, _ = star_field(n, 16)
for _ in range(3):
    ops.gp_solve(spec, X, y - y.mean(), ye, want_alpha=False)
print("chol ms", _lib.timings(_lib.get_ctx())[1])
raw = C.CDLL(_lib.LIB_PATH)
cap = 1 << 16
buf = (C.c_ulonglong * (4 * cap))()
m = raw.tgp_debug_pchol_trace(buf, cap)
print("records", m, "dataflow launches so far", raw.tgp_debug_pchol_launches(), "last error", _lib.load_library().tgp_last_error(_lib.get_ctx()))
if m <= 0:
    sys.exit(1)
r = np.array(buf[:4 * m], dtype=np.uint64).reshape(m, 4)
task, wg = r[:, 0].astype(np.int64), r[:, 1].astype(np.int64)
t0 = (r[:, 2] - r[:, 2].min()).astype(np.float64) / 100.0
t1 = (r[:, 3] - r[:, 2].min()).astype(np.float64) / 100.0
typ, k, i = task >> 24, (task >> 16) & 255, (task >> 8) & 255
names = ["D0", "A1", "A2", "D1", "R12", "R3", "U"]
span = t1.max()
print("span %.1f us" % span)
for t in range(7):
    s = typ == t
    if s.any():
        d = t1[s] - t0[s]
        print("%-4s x %5d  busy %9.1f us  mean %6.1f  min %6.1f  max %6.1f" % (names[t], s.sum(), d.sum(), d.mean(), d.min(), d.max()))
workers = wg < 0x10000
print("worker busy fraction: %.3f of %d workgroups x span" % ((t1[workers] - t0[workers]).sum() / (len(set(wg[workers])) * span), len(set(wg[workers]))))
print("panel:  D0 start-end | A1 first-last end | A2 last end | D1 start-end | R12 first start - last end | R3 last end | U first start - last end")
for p in range(min(npan, int(k.max()) + 1)):
    def sel(t):
        return (typ == t) & (k == p)
    def rng(t):
        s = sel(t)
        return (t0[s].min(), t1[s].max()) if s.any() else (float("nan"), float("nan"))
    d0, a1, a2, d1, r12, r3, u = [rng(t) for t in range(7)]
    print("%3d: %7.1f-%7.1f | %7.1f-%7.1f | %7.1f | %7.1f-%7.1f | %7.1f-%7.1f | %7.1f | %7.1f-%7.1f"
          % (p, d0[0], d0[1], a1[0], a1[1], a2[1], d1[0], d1[1], r12[0], r12[1], r3[1], u[0], u[1]))

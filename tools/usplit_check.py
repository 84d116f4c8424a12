#!/usr/bin/env python3
"""Split panel boundaries (TGP_USPLIT): the weights of a solve with and without, per size -- relative difference and residual."""
import os, subprocess, sys, numpy as np
here = os.path.dirname(os.path.abspath(__file__))
code = r'''
import sys, numpy as np
sys.path.insert(0, %r)
from treegp_amd import _lib, ops
from treegp_amd.synthetic import star_field, headline_invlam
iL = headline_invlam(); spec = ops.KernelSpec(_lib.TGP_ARBF, amp=1.0, a=iL[0,0], b=iL[0,1], c=iL[1,1])
out = {}
for n in [int(v) for v in sys.argv[2:]]:
    X, y, ye, _ = star_field(n, 16, seed=n)
    y = y - y.mean()
    a, ld, yd, _ = ops.gp_solve(spec, X, y, ye)
    r = ops.gp_predict(spec, X, a, X) + ye ** 2 * a - y
    out["a%%d" %% n] = a; out["l%%d" %% n] = ld
    print(n, "residual %%.2e logdet %%.9f" %% (np.linalg.norm(r) / np.linalg.norm(y), ld), flush=True)
np.savez(sys.argv[1], **out)
''' % os.path.dirname(here)
sizes = sys.argv[1:] or ["700", "1300", "2048", "3000", "5000", "8192", "12288", "24000"]
res = []
for us in ("0", "1"):
    fn = "/tmp/usplit_%s.npz" % us
    r = subprocess.run([sys.executable, "-c", code, fn] + sizes, env=dict(os.environ, TGP_USPLIT=us), capture_output=True, text=True, timeout=600)
    print("TGP_USPLIT=" + us); print(r.stdout.strip())
    if r.returncode != 0:
        print(r.stderr[-1500:]); sys.exit(1)
    res.append(np.load(fn))
worst = 0.0
for n in sizes:
    a0, a1 = res[0]["a" + n], res[1]["a" + n]
    d = np.linalg.norm(a1 - a0) / np.linalg.norm(a0)
    worst = max(worst, d)
    print("N=%s: |alpha1 - alpha0| / |alpha0| = %.2e, logdet diff %.2e" % (n, d, abs(float(res[0]["l" + n]) - float(res[1]["l" + n]))))
sys.exit(0 if worst < 1e-9 else 1)

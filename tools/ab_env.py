#!/usr/bin/env python3
"""A/B of environment switches with the boxes' run-to-run spread averaged out: every configuration is run in a fresh process
(the switches are read once per process), the configurations in turn, REPS times over; per configuration and size the minimum
and the median of the per-process best-of-8 Cholesky times.
usage: python tools/ab_env.py REPS N[,N...] "K=V K2=V2" "K=V" ...      ("-" = no switch)"""
import os, subprocess, sys, statistics
reps = int(sys.argv[1]); sizes = sys.argv[2].split(","); cfgs = sys.argv[3:]
here = os.path.dirname(os.path.abspath(__file__))
code = r'''
import sys, numpy as np
sys.path.insert(0, %r)
from treegp_amd import _lib, ops
from treegp_amd.synthetic import star_field, headline_invlam
iL = headline_invlam(); spec = ops.KernelSpec(_lib.TGP_ARBF, amp=1.0, a=iL[0,0], b=iL[0,1], c=iL[1,1])
for n in [int(v) for v in sys.argv[1:]]:
    X, y, ye, _ = star_field(n, 16)
    best = 1e9
    for it in range(9):
        ops.gp_solve(spec, X, y - y.mean(), ye)
        if it: best = min(best, _lib.timings(_lib.get_ctx())[1])
    print(n, best, flush=True)
''' % os.path.dirname(here)
res = {c: {n: [] for n in sizes} for c in cfgs}
for r in range(reps):
    for c in cfgs:
        env = dict(os.environ)
        if c != "-":
            for kv in c.split():
                k, v = kv.split("=", 1); env[k] = v
        out = subprocess.run([sys.executable, "-c", code] + sizes, env=env, capture_output=True, text=True, timeout=280)
        if out.returncode != 0:
            print("FAILED", c, out.stderr[-400:]); sys.exit(1)
        for line in out.stdout.split("\n"):
            p = line.split()
            if len(p) == 2 and p[0] in res[c]: res[c][p[0]].append(float(p[1]))
    print("rep", r + 1, "done", flush=True)
print("%-58s" % "configuration" + "".join("%22s" % ("N=%s min / med" % n) for n in sizes))
for c in cfgs:
    print("%-58s" % c + "".join("%12.3f / %7.3f" % (min(res[c][n]), statistics.median(res[c][n])) for n in sizes))

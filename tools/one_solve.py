#!/usr/bin/env python3
"""Three host-boundary solves at one size (for a kernel trace of the small-N schedule).  usage: one_solve.py N"""
import sys
sys.path.insert(0, __file__.rsplit("/tools/", 1)[0])
from treegp_amd import _lib, ops  # noqa: E402
from treegp_amd.synthetic import star_field, headline_invlam  # noqa: E402
n = int(sys.argv[1])
iL = headline_invlam()
spec = ops.KernelSpec(_lib.TGP_ARBF, amp=1.0, a=iL[0, 0], b=iL[0, 1], c=iL[1, 1])
X, y, ye, Xs = star_field(n, 16)
for _ in range(3):
    ops.gp_solve(spec, X, y - y.mean(), ye, want_alpha=False)
print(_lib.timings(_lib.get_ctx())[:3])

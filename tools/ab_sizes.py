#!/usr/bin/env python3
"""Cholesky time and rate at a list of sizes (best of 4), for A/B runs of two builds: TGP_LIB_PATH=... python tools/ab_sizes.py [N ...]"""
import sys, time, numpy as np
sys.path.insert(0, __file__.rsplit("/tools/", 1)[0])
from treegp_amd import _lib, ops
from treegp_amd.synthetic import star_field, headline_invlam
iL = headline_invlam(); spec = ops.KernelSpec(_lib.TGP_ARBF, amp=1.0, a=iL[0,0], b=iL[0,1], c=iL[1,1])
for n in [int(v) for v in sys.argv[1:]] or (22528, 24576, 32768, 49152):
    X, y, ye, _ = star_field(n, 16)
    best = 1e9
    for it in range(4):
        ops.gp_solve(spec, X, y - y.mean(), ye)
        best = min(best, _lib.timings(_lib.get_ctx())[1])
    print(n, "chol ms %.3f  TF %.2f" % (best, n**3/3/best/1e9), flush=True)

#!/usr/bin/env python3
"""sha1 of the weights of one solve per size -- to see whether a build or switch changes any bit: python tools/solve_digest.py [N ...]"""
import hashlib, sys, numpy as np
sys.path.insert(0, __file__.rsplit("/tools/", 1)[0])
from treegp_amd import _lib, ops
from treegp_amd.synthetic import star_field, headline_invlam
iL = headline_invlam(); spec = ops.KernelSpec(_lib.TGP_ARBF, amp=1.0, a=iL[0,0], b=iL[0,1], c=iL[1,1])
for n in [int(v) for v in sys.argv[1:]] or (3000, 8192):
    X, y, ye, _ = star_field(n, 16)
    out = ops.gp_solve(spec, X, y - y.mean(), ye)
    alpha = np.ascontiguousarray(out[0] if isinstance(out, tuple) else out.alpha)
    print(n, hashlib.sha1(alpha.tobytes()).hexdigest()[:16], flush=True)

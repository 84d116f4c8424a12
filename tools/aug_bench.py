import sys, numpy as np
sys.path.insert(0, "/root/repo")
from treegp_amd import _lib, ops
from treegp_amd.synthetic import star_field, headline_invlam
iL = headline_invlam(); spec = ops.KernelSpec(_lib.TGP_ARBF, amp=1.0, a=iL[0,0], b=iL[0,1], c=iL[1,1])
for n in (4000, 8000, 16000, 32000, 65000):
    X, y, ye, _ = star_field(n, 16); y = y - y.mean()
    best = (1e9, 0, 0)
    for it in range(4):
        ops.gp_solve(spec, X, y, ye)
        tm = _lib.timings(_lib.get_ctx())
        if tm[1] + tm[2] < best[0]: best = (tm[1] + tm[2], tm[1], tm[2])
    print(n, "chol+trsv %.3f ms (chol %.3f trsv %.3f) sweeps %g" % (best + (tm[10],)), flush=True)

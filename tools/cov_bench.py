import sys, time, numpy as np
sys.path.insert(0, sys.argv[1])
from treegp_amd import _lib, ops
from treegp_amd.synthetic import star_field, headline_invlam
iL = headline_invlam(); spec = ops.KernelSpec(_lib.TGP_ARBF, amp=1.0, a=iL[0,0], b=iL[0,1], c=iL[1,1])
for n, m in ((8192, 4096), (2048, 2048), (16384, 8192)):
    X, y, ye, Xs = star_field(n, m)
    a, ld, yd, fac = ops.gp_solve(spec, X, y - y.mean(), ye, keep=True)
    ts = []
    for it in range(4):
        t0 = time.perf_counter(); cov = ops.gp_predict_cov(spec, fac, X, Xs); ts.append((time.perf_counter()-t0)*1e3)
        tm = _lib.timings(_lib.get_ctx()); dev = (tm[3], tm[9])
    print("cov N=%d M=%d: %s ms wall; last call: device compute %.2f ms, D2H of the result %.2f ms; min diag %.3e" % (n, m, ["%.1f" % t for t in ts], dev[0], dev[1], cov.diagonal().min()))
    fac.free()

import sys, numpy as np, time
sys.path.insert(0, sys.argv[1])
from treegp_amd import _lib, ops
from treegp_amd.synthetic import star_field, headline_invlam
iL = headline_invlam(); spec = ops.KernelSpec(_lib.TGP_ARBF, amp=1.0, a=iL[0,0], b=iL[0,1], c=iL[1,1])
for n in (24576, 28672, 28800, 29000, 33011, 40000):
    X, y, ye, _ = star_field(n, 16)
    y = y - y.mean()
    t0 = time.perf_counter(); alpha, logdet, ydota, _ = ops.gp_solve(spec, X, y, ye); dt = time.perf_counter() - t0
    r = ops.gp_predict(spec, X, alpha, X) + ye ** 2 * alpha - y
    print("N=%d: solve %.1f ms, relative residual %.2e, logdet %.6f" % (n, dt * 1e3, np.linalg.norm(r) / np.linalg.norm(y), logdet), flush=True)

import os, sys, ctypes as C, numpy as np
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from treegp_amd import _lib, ops
from treegp_amd.synthetic import star_field, headline_invlam
lib, ctx = _lib.load_library(), _lib.get_ctx()
iL = headline_invlam()
spec = ops.KernelSpec(_lib.TGP_ARBF, amp=1.0, a=iL[0, 0], b=iL[0, 1], c=iL[1, 1])
rng = np.random.default_rng(123)
sizes = [2049, 2303, 2304, 2305, 2560, 2816, 3071, 3072, 3073, 3329, 4095, 4097, 5121, 6400, 7000, 7425, 10241, 13057] + list(rng.integers(2048, 9000, 6))
worst = 0.0
for n in sizes:
    n = int(n)
    X, y, ye, Xs = star_field(n, 64, seed=n)
    os.environ["TGP_POTRS_BIG_FROM"] = "0"
    a0, ld0, yd0, _ = ops.gp_solve(spec, X, y - y.mean(), ye)
    os.environ["TGP_POTRS_BIG_FROM"] = "256"
    a1, ld1, yd1, f = ops.gp_solve(spec, X, y - y.mean(), ye, keep=True)
    B = rng.standard_normal((5, n)); B[0] = y - y.mean()
    S = ops.factor_solve(f, B)
    os.environ["TGP_COV_BIG"] = "0"; c0 = ops.gp_predict_cov(spec, f, X, Xs)
    os.environ.pop("TGP_COV_BIG"); c1 = ops.gp_predict_cov(spec, f, X, Xs)
    f.free()
    d = max(np.abs(a1 - a0).max() / np.abs(a0).max(), np.abs(S[0] - a0).max() / np.abs(a0).max(), np.abs(c1 - c0).max() / np.abs(c0).max(), abs(yd1 - yd0) / abs(yd0))
    worst = max(worst, d)
    print(n, "%.1e" % d, flush=True)
    assert d < 1e-10, (n, d)
print("worst", worst)

import os, sys, time, numpy as np
sys.path.insert(0, "/root/repo" if os.path.exists("/root/repo/treegp_amd") else ".")
import treegp_amd as treegp
def problem(n, seed=11):
    rng = np.random.default_rng(seed)
    X = rng.uniform(0, 1, (n, 2))
    y = np.sin(6 * X[:, 0]) * np.cos(4 * X[:, 1]) + 0.05 * rng.standard_normal(n)
    return X, y, 0.05 * rng.uniform(0.8, 1.2, n)
os.environ["TGP_ML_GRADIENT"] = "fd"
for n in (600, 1024, 2048, 4096):
    X, y, e = problem(n)
    for cap in (5, 4, 3):
        os.environ["TGP_ML_MAX_CONCURRENT"] = str(cap)
        best = 1e9
        for _ in range(4):
            gp = treegp.GPInterpolation(kernel="0.7**2 * AnisotropicRBF(invLam=array([[60., 0.], [0., 60.]]))", optimizer="log-likelihood", normalize=True)
            gp.initialize(X, y, y_err=e)
            t0 = time.perf_counter(); gp.solve(); best = min(best, time.perf_counter() - t0)
        print("n=%d cap=%d fit %.1f ms logL %.6f" % (n, cap, best * 1e3, gp._optimizer._logL), flush=True)

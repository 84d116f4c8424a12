import os, sys, numpy as np
sys.path.insert(0, '/root/repo' if os.path.exists('/root/repo/treegp_amd') else os.environ.get('GRAFT_REPO_ROOT','.'))
from treegp_amd import _lib, ops
rng = np.random.default_rng(0)
for n, ell, noise in ((3000, 0.05, 3e-2), (3000, 0.1, 1e-3), (3000, 0.2, 1e-4), (5000, 0.3, 1e-4), (5000, 0.3, 1e-5)):
    X = rng.uniform(0, 1, (n, 2)); y = np.sin(5*X[:,0])*np.cos(3*X[:,1]) + noise*rng.standard_normal(n)
    ye = np.full(n, noise)
    spec = ops.KernelSpec(_lib.TGP_RBF, amp=1.0, a=1/ell**2, b=0.0, c=1/ell**2)
    K = ops.kernel_matrix(spec, X) + np.diag(ye**2)           # the library's own K (seam S1); SciPy solves it below
    res = {}
    for tag, env in (("chain", "0"), ("big", "256")):
        os.environ["TGP_POTRS_BIG_FROM"] = env
        try:
            a = ops.gp_solve(spec, X, y, ye)[0]
        except np.linalg.LinAlgError as e:
            print(n, ell, noise, tag, "not PD"); a = None
        res[tag] = a
    if res["chain"] is None: continue
    ev = np.linalg.eigvalsh(K); cond = ev[-1]/ev[0]
    r = {t: np.linalg.norm(K @ a - y)/np.linalg.norm(y) for t, a in res.items()}
    import scipy.linalg as sl
    aref = sl.cho_solve(sl.cho_factor(K, lower=False), y)
    d = {t: np.abs(a - aref).max()/np.abs(aref).max() for t, a in res.items()}
    print("n=%d ell=%.2f noise=%.0e cond=%.1e  residual chain %.1e big %.1e | alpha rel diff vs scipy: chain %.1e big %.1e | big vs chain %.1e" % (n, ell, noise, cond, r["chain"], r["big"], d["chain"], d["big"], np.abs(res["big"]-res["chain"]).max()/np.abs(res["chain"]).max()))

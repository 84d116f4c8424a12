"""Bootstrap of the TwoD pair correlation at the API's anisotropic-fit size: pair-list path vs per-resample path.
usage: python tools/boot_bench.py [n] [n_boot]"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from treegp_amd import _lib, ops
from treegp_amd.synthetic import star_field

n = int(sys.argv[1]) if len(sys.argv) > 1 else 32768
nboot = int(sys.argv[2]) if len(sys.argv) > 2 else 444
X, y, ye, _ = star_field(n, 16); y = y - y.mean()
rng = np.random.default_rng(610639139)
idx = np.stack([rng.integers(0, n - 1, size=n) for _ in range(nboot)])
res = {}
for mode in ("1", "0"):
    os.environ["TGP_BOOT_LISTS"] = mode
    for it in range(3):
        t0 = time.perf_counter(); xi = ops.kk_twod_bootstrap(X[:, 0], X[:, 1], y, ye, idx, 0.0, 0.15, 21); t1 = time.perf_counter()
        print("lists=%s  wall %.1f ms, device (events incl. copies) %.1f ms" % (mode, (t1 - t0) * 1e3, _lib.timings(_lib.get_ctx())[4]), flush=True)
    res[mode] = xi
d = np.abs(res["1"] - res["0"]).max() / np.abs(res["0"]).max()
print("max |lists - per-resample| / max|xi| = %.3e" % d)

"""The reference's own CPU call sequence for the hot path, issued call for call.

TEST INFRASTRUCTURE / REPORTED BASELINE ONLY (see oracle/__init__.py).  Unlike
``gp_oracle.py`` (explicit formulas), this file makes the very SciPy/NumPy calls the
reference makes, in its order, so that timing it on the GPU box's host cores is what a
treegp user gets today:

  kernels.py:117-121   pdist(mahalanobis, VI) -> exp(-.5 d^2) -> squareform -> fill_diagonal
  sklearn Product      np.full(amp) * K                      (sklearn kernels.py:931-966)
  gp_interp.py:180     + np.eye(N) * y_err**2
  gp_interp.py:181-182 cholesky(overwrite_a=True, lower=False) ; cho_solve
  kernels.py:125-126   cdist(mahalanobis, VI) -> exp           (gp_interp.py:177)
  gp_interp.py:183     HT @ alpha
"""
import time

import numpy as np
from scipy.linalg import cho_solve, cholesky
from scipy.spatial.distance import cdist, pdist, squareform


def anisotropic_rbf_self(X, invLam, amp):
    dists = pdist(X, metric="mahalanobis", VI=invLam)
    K = np.exp(-0.5 * dists ** 2)
    K = squareform(K)
    np.fill_diagonal(K, 1)
    return np.full((len(X), len(X)), amp) * K


def anisotropic_rbf_cross(Xs, X, invLam, amp):
    dists = cdist(Xs, X, metric="mahalanobis", VI=invLam)
    K = np.exp(-0.5 * dists ** 2)
    return np.full(K.shape, amp) * K


def solve_predict(X, y, y_err, Xs, invLam, amp, timings=None):
    """alpha, y_pred exactly as GPInterpolation.return_gp_predict computes them
    (gp_interp.py:168-183), with per-phase wall times appended to ``timings``."""
    t = [time.perf_counter()]
    K = anisotropic_rbf_self(X, invLam, amp) + np.eye(len(y)) * y_err ** 2
    t.append(time.perf_counter())
    factor = (cholesky(K, overwrite_a=True, lower=False), False)
    t.append(time.perf_counter())
    alpha = cho_solve(factor, y, overwrite_b=False)
    t.append(time.perf_counter())
    HT = anisotropic_rbf_cross(Xs, X, invLam, amp)
    t.append(time.perf_counter())
    y_pred = np.dot(HT, alpha.reshape((len(alpha), 1))).T[0]
    t.append(time.perf_counter())
    if timings is not None:
        timings.update(kbuild=t[1] - t[0], cholesky=t[2] - t[1], cho_solve=t[3] - t[2], cross_kernel=t[4] - t[3],
                       matvec=t[5] - t[4], total=t[5] - t[0])
    return alpha, y_pred


# ---------------------------------------------------------------------------------------------------
# timing protocol of BASELINE.md section 3 (bench.py's cpu_baseline leg)
# ---------------------------------------------------------------------------------------------------
def host_cpus():
    """What the host offers this process: logical CPUs, physical cores (distinct (socket, core) pairs of
    /proc/cpuinfo), the scheduler affinity and the cgroup CPU quota if one is set."""
    import os
    info = {"logical": os.cpu_count() or 1, "physical": None, "affinity": None, "cgroup_quota": None, "model": "unknown CPU"}
    try:
        info["affinity"] = len(os.sched_getaffinity(0))
    except (AttributeError, OSError):
        pass
    try:
        cores, phys, core = set(), None, None
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name") and info["model"] == "unknown CPU":
                info["model"] = line.split(":", 1)[1].strip()
            elif line.startswith("physical id"):
                phys = line.split(":", 1)[1].strip()
            elif line.startswith("core id"):
                core = line.split(":", 1)[1].strip()
            elif not line.strip():
                if phys is not None and core is not None:
                    cores.add((phys, core))
                phys = core = None
        info["physical"] = len(cores) or None
    except OSError:
        pass
    for path in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us"):
        try:
            txt = open(path).read().split()
            if path.endswith("cpu.max"):
                if txt[0] != "max":
                    info["cgroup_quota"] = float(txt[0]) / float(txt[1])
            else:
                q = float(txt[0])
                if q > 0:
                    info["cgroup_quota"] = q / float(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            break
        except (OSError, ValueError, IndexError):
            continue
    return info


def pick_blas_threads(candidates, n=4096):
    """dpotrf GFLOP/s at order n for each BLAS thread count (best of 2 after a warm-up); the fastest is what the
    baseline runs with.  Returns (best, {threads: GFLOP/s})."""
    from threadpoolctl import threadpool_limits
    rng = np.random.default_rng(0)
    A = rng.standard_normal((n, 64))
    K = A @ A.T + n * np.eye(n)
    rates = {}
    for t in candidates:
        with threadpool_limits(limits=int(t), user_api="blas"):
            cholesky(K.copy(), overwrite_a=True, lower=False)
            best = np.inf
            for _ in range(2):
                B = K.copy()
                t0 = time.perf_counter()
                cholesky(B, overwrite_a=True, lower=False)
                best = min(best, time.perf_counter() - t0)
        rates[int(t)] = n ** 3 / 3.0 / best / 1e9
    return max(rates, key=rates.get), rates


def timed_passes(X, y, y_err, Xs, invLam, amp, threads, passes=5, budget_s=90.0):
    """1 warm-up + up to ``passes`` timed passes of solve_predict with BLAS limited to ``threads``; stops early (never
    below 3 timed passes) when the time budget is spent -- configs[1] in full is ~10 s a pass on 16 threads, 5 passes fit.  Returns ({phase: median seconds}, number of timed passes)."""
    from threadpoolctl import threadpool_limits
    rows = []
    t_start = time.perf_counter()
    with threadpool_limits(limits=int(threads), user_api="blas"):
        solve_predict(X, y, y_err, Xs, invLam, amp, timings={})            # warm-up: first touch, thread start-up
        for i in range(passes):
            tm = {}
            solve_predict(X, y, y_err, Xs, invLam, amp, timings=tm)
            rows.append(tm)
            if i >= 2 and time.perf_counter() - t_start > budget_s:
                break
    med = {k: float(np.median([r[k] for r in rows])) for k in rows[0]}
    return med, len(rows)

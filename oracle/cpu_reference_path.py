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
from scipy import special
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


# ---------------------------------------------------------------------------------------------------
# the other BASELINE.json configs of BASELINE.md section 3
# ---------------------------------------------------------------------------------------------------
def config0_passes(threads, passes=5):
    """configs[0] in full -- 1-D AnisotropicRBF, N = 512 training / 1024 prediction points, the reference's own
    CPU-runnable case (tests/test_gp_interp.py style: X ~ U(-10, 10), scale_length 2, noise 0.1, X* = linspace):
    1 warm-up + `passes` timed passes of the same call sequence, median per phase."""
    from threadpoolctl import threadpool_limits
    rng = np.random.default_rng(20240613)
    n, m = 512, 1024
    X = rng.uniform(-10.0, 10.0, (n, 1))
    y = np.sin(X[:, 0]) + 0.1 * rng.standard_normal(n)
    y_err = 0.1 * rng.uniform(0.8, 1.2, n)
    Xs = np.linspace(-10.0, 10.0, m)[:, None]
    invLam = np.array([[1.0 / 2.0 ** 2]])                   # scale_length=[2.0]  (kernels.py:95-112)
    rows = []
    with threadpool_limits(limits=int(threads), user_api="blas"):
        solve_predict(X, y - y.mean(), y_err, Xs, invLam, 1.0, timings={})
        for _ in range(passes):
            tm = {}
            solve_predict(X, y - y.mean(), y_err, Xs, invLam, 1.0, timings=tm)
            rows.append(tm)
    med = {k: float(np.median([r[k] for r in rows])) for k in rows[0]}
    return {"n_train": n, "m_predict": m, "passes": passes, "phases_s": med, "value": (n + m) / med["total"], "unit": "points/s"}


def von_karman_self(X, length_scale):
    """VonKarman.__call__(X) call for call (kernels.py:249-262): pdist(euclidean), scipy.special.kv on the
    non-zero separations, squareform, the limit on the diagonal."""
    dists = pdist(X, metric="euclidean")
    Filter = dists != 0.0
    K = np.zeros_like(dists)
    K[Filter] = (dists[Filter] / length_scale) ** (5.0 / 6.0) * special.kv(5.0 / 6.0, 2 * np.pi * dists[Filter] / length_scale)
    K = squareform(K)
    lim0 = special.gamma(5.0 / 6.0) / (2 * (np.pi ** (5.0 / 6.0)))
    np.fill_diagonal(K, lim0)
    K /= lim0
    return K


def config2_sample(threads, n_s=2048, n_full=32768, passes=5, length_scale=0.1):
    """configs[2] SAMPLED: the von Karman K build (single-threaded pdist + kv + squareform), dpotrf and cho_solve at
    n_s points of the same star-field recipe, median of `passes` after a warm-up; per-element / per-flop rates and a
    LABELLED extrapolation to N = n_full (K build by N^2, dpotrf by N^3 at the measured GFLOP/s of the larger
    configs[1] factorisation when the caller supplies it)."""
    from threadpoolctl import threadpool_limits
    rng = np.random.default_rng(20240613)
    X = rng.uniform(0.0, 1.0, (n_s, 2))
    y = rng.standard_normal(n_s)
    y_err = 0.03 * rng.uniform(0.8, 1.2, n_s)
    rows = []
    with threadpool_limits(limits=int(threads), user_api="blas"):
        for it in range(passes + 1):
            t0 = time.perf_counter()
            K = von_karman_self(X, length_scale) + np.eye(n_s) * y_err ** 2
            t1 = time.perf_counter()
            factor = (cholesky(K, overwrite_a=True, lower=False), False)
            t2 = time.perf_counter()
            cho_solve(factor, y, overwrite_b=False)
            t3 = time.perf_counter()
            if it:
                rows.append({"kbuild": t1 - t0, "cholesky": t2 - t1, "cho_solve": t3 - t2})
    med = {k: float(np.median([r[k] for r in rows])) for k in rows[0]}
    ns_per_element = med["kbuild"] / (n_s * (n_s - 1) / 2.0) * 1e9
    return {"n_sample": n_s, "passes": passes, "phases_s": med,
            "kbuild_ns_per_element": ns_per_element, "kbuild_elements_per_sec": 1e9 / ns_per_element,
            "dpotrf_gflops_at_sample": n_s ** 3 / 3.0 / med["cholesky"] / 1e9,
            "extrapolated_n%d" % n_full: {
                "label": "EXTRAPOLATED, not measured: K build x (N/%d)^2 (single-threaded pdist + scipy.special.kv + "
                         "squareform), cho_solve x (N/%d)^2" % (n_s, n_s),
                "kbuild_s": med["kbuild"] * (n_full / n_s) ** 2, "cho_solve_s": med["cho_solve"] * (n_full / n_s) ** 2}}

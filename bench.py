#!/usr/bin/env python3
"""Headline benchmark: GP solve + predict on the synthetic 2-D star field (BASELINE.json).

A step = one pass of the hot path with inputs resident in HBM:
    K build (AnisotropicRBF + y_err^2 diagonal) -> blocked fp64 Cholesky -> triangular solves
    (+ logdet, y.alpha) -> fused predict of M points.
N=1 workload: the configuration the metric is quoted on, 2-D AnisotropicRBF N=65 536
(configs[3]'s problem on one GPU; the packed factor is 17 GB), M = 4 N prediction points.
With --gpus G > 1 (launched by torch.distributed.run) the SAME problem is factorised with the
row-block-cyclic distributed Cholesky (treegp_amd/dist.py) and the prediction points are
sharded: strong scaling.

Prints ONE JSON line (rank 0).  `roofline` is for the dominant kernel, the trailing-update
syrk on v_mfma_f64_16x16x4_f64: algorithmic flops per launch = 256 m (m+1) for a trailing
matrix of order m (N^3/3 in total), time from hipEvents around every launch
on the library's stream inside the timed region.
"""
import argparse
import ctypes as C
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

if int(os.environ.get("WORLD_SIZE", "1")) > 1:
    # A rank of a multi-GPU run keeps five streams busy (bulk, panel chain, keep copies, RCCL's, torch's default) and parks
    # stream wait-values on one of them; the runtime multiplexes streams onto 4 hardware queues by default, and two streams
    # that share a queue run in order (round 4 saw the bulk and the chain share one: chain 75 -> 93 ms).  Eight queues cost
    # nothing measurable on one GPU (profiles/r04_hw_queues_ab.txt); must be set before the HIP runtime starts.
    os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")

FP64_MFMA_PEAK_TFLOPS = 78.6     # MI355X fp64 matrix peak


def cpu_baseline(n_s, m_s, n_headline):
    """The reference's own SciPy call sequence (oracle/cpu_reference_path.py) on the host cores, protocol of
    BASELINE.md section 3: BLAS threads = the fastest of {cgroup quota, physical cores, half, quarter, 16, 32}
    on a dpotrf probe, 1 warm-up + 5 timed passes (fewer, never below 3, only if a 90 s budget is spent), median per phase; sample =
    configs[1] in full (N=8192 / M=32768).  `extrapolated_n65536` scales the measured phases to the headline size
    (dpotrf by N^3, pair loops by N^2 and M N) and is labelled as such."""
    from oracle import cpu_reference_path as R
    from treegp_amd.synthetic import star_field, headline_invlam
    X, y, y_err, Xs = star_field(n_s, m_s)
    iL = headline_invlam()
    host = R.host_cpus()
    phys = host["physical"] or host["logical"]
    cand = {phys, max(phys // 2, 1), max(phys // 4, 1), 16, 32}
    if host["cgroup_quota"]:
        cand.add(max(int(round(host["cgroup_quota"])), 1))
    cand = sorted(c for c in cand if 1 <= c <= host["logical"])
    threads, probe = R.pick_blas_threads(cand)
    med, npass = R.timed_passes(X, y - y.mean(), y_err, Xs, iL, 1.0, threads)
    blas = "unknown BLAS"
    try:
        from threadpoolctl import threadpool_info
        blas = ", ".join(sorted({"%s %s" % (p.get("internal_api", "?"), p.get("version", "")) for p in threadpool_info()
                                 if p.get("user_api") == "blas"})) or blas
    except Exception:
        pass
    gf = n_s ** 3 / 3 / med["cholesky"] / 1e9
    # cores the baseline really had: a cgroup quota below the thread count is what binds (16 CPUs on this pool's boxes)
    eff = min([float(threads)] + [float(v) for v in (host["cgroup_quota"], host["affinity"]) if v])
    r3, r2 = (n_headline / n_s) ** 3, (n_headline / n_s) ** 2
    m_h = 4 * n_headline
    ext = {"kbuild": med["kbuild"] * r2, "cholesky": med["cholesky"] * r3, "cho_solve": med["cho_solve"] * r2,
           "cross_kernel": med["cross_kernel"] * (m_h * n_headline) / (m_s * n_s),
           "matvec": med["matvec"] * (m_h * n_headline) / (m_s * n_s)}
    ext_total = sum(ext.values())
    # BASELINE.md section 3's other sizes: configs[0] in full, configs[2] sampled (von Karman K build per element + dpotrf)
    c0 = R.config0_passes(threads)
    c2 = R.config2_sample(threads)
    key32 = [k for k in c2 if k.startswith("extrapolated_n")][0]
    c2[key32]["cholesky_s_at_configs1_gflops"] = 32768 ** 3 / 3.0 / (gf * 1e9)      # dpotrf at N = 32768 at configs[1]'s measured rate
    # the pair binning of configs[2] has NO reference CPU baseline here: it lives in TreeCorr (treegp/two_pcf.py:297-305, 330-334),
    # absent from this image.  What can be timed is the oracle's brute-force restatement -- not the reference -- on a sample.
    from oracle import gp_oracle as O
    import time as _time
    n_pb = 4096
    Xp, yp, ep, _ = star_field(n_pb, 16)
    t0 = _time.perf_counter()
    O.kk_log(Xp[:, 0], Xp[:, 1], yp - yp.mean(), 1.0 / ep ** 2, 1.0 / np.sqrt(n_pb), 0.7, 20)
    t_pb = _time.perf_counter() - t0
    c2["pair_binning"] = ("no CPU baseline: TreeCorr absent; oracle brute force %.2f s at N = %d = %.2e pairs/s on one core "
                          "(restatement, not the reference)" % (t_pb, n_pb, n_pb * (n_pb - 1) / 2.0 / t_pb))
    return {
        "configs": {"configs[0]": dict(c0, sample="in full: 1-D AnisotropicRBF N=512 / M=1024, the reference's call sequence, "
                                                  "1 warm-up + 5 passes, median per phase"),
                    "configs[2]": dict(c2, sample="SAMPLED at N=%d of 32768: von Karman K build (pdist + scipy.special.kv + squareform, "
                                                  "single-threaded), dpotrf, cho_solve; the N=32768 figures are labelled "
                                                  "extrapolations" % c2["n_sample"])},
        "value": (n_s + m_s) / med["total"], "unit": "points/s", "cores": max(int(round(eff)), 1), "blas_threads": int(threads),
        "kind": "port",
        "sample": "configs[1] in full: same star-field recipe at N=%d train / M=%d predict; the reference's SciPy calls "
                  "(oracle/cpu_reference_path.py), 1 warm-up + %d timed passes, median per phase: pdist+exp+squareform "
                  "%.2fs (single-thread), dpotrf %.3fs = %.0f GFLOP/s on %d BLAS threads (%g effective cores), cho_solve %.3fs, cdist+exp "
                  "%.2fs (single-thread); host: %s, %d logical / %s physical CPUs, cgroup quota %s, %s; the O(N^3) "
                  "factorisation makes points/s size-dependent -- compare with the GPU at the same N, not with `value`"
                  % (n_s, m_s, npass, med["kbuild"], med["cholesky"], gf, threads, eff, med["cho_solve"], med["cross_kernel"],
                     host["model"], host["logical"], host["physical"], host["cgroup_quota"], blas),
        "passes": npass, "blas_threads_probe_gflops": {str(k): round(v, 1) for k, v in probe.items()},
        "dpotrf_gflops": gf,
        "phases_s": {k: round(v, 4) for k, v in med.items()},
        "extrapolated_n%d" % n_headline: {
            "label": "EXTRAPOLATED, not measured: dpotrf time x (N/%d)^3, pair loops x N^2 resp. M N, M = 4 N" % n_s,
            "phases_s": {k: round(v, 2) for k, v in ext.items()}, "total_s": round(ext_total, 1),
            "value": (n_headline + m_h) / ext_total, "unit": "points/s"},
    }


def _csrc_digest():
    """sha256 over the HIP sources: a PMC file is only quoted for the code it was collected on."""
    import hashlib
    h = hashlib.sha256()
    d = os.path.join(ROOT, "treegp_amd", "csrc")
    for fn in sorted(os.listdir(d)):
        if fn.endswith((".hip", ".h")):
            h.update(fn.encode())
            h.update(open(os.path.join(d, fn), "rb").read())
    return h.hexdigest()[:16]


def measured_traffic(n, kernel_key):
    """HBM-side bytes per launch of a kernel from the newest profiles/r*_pmc_bench_n<N>.json (separate rocprofv3
    --pmc passes of this very command, tools/pmc_bench.sh, corrected as the microarchitecture guide's HBM section
    prescribes).  Returns (bytes or None, provenance string): None when no file was collected on the shipped
    sources -- a stale number is not quoted."""
    import glob
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_pmc_bench_n%d.json" % n)))
    if not files:
        return None, "no PMC file for this workload"
    rec = json.load(open(files[-1]))
    der = rec.get("_derived", {})
    name = os.path.relpath(files[-1], ROOT)
    if der.get("csrc_digest") != _csrc_digest():
        return None, "%s was collected on other sources (digest %s, now %s): not quoted" % (name, der.get("csrc_digest"), _csrc_digest())
    return der.get(kernel_key), "%s (csrc digest %s)" % (name, der["csrc_digest"])


def measured_pipe():
    """MFMA-pipe occupancy and effective clock of the dominant kernel from the newest profiles/r*_pmc_sq.json (tools/pmc_sq.sh:
    SQ_VALU_MFMA_BUSY_CYCLES and GRBM_GUI_ACTIVE passes on the shipped sources); None when collected on other sources."""
    import glob
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_pmc_sq.json")))
    if not files:
        return None
    rec = json.load(open(files[-1]))
    if rec.get("csrc_digest") != _csrc_digest():
        return None
    return {"mfma_busy_frac": rec.get("mfma_busy_frac"), "effective_clock_ghz_profiled": rec.get("effective_clock_ghz"),
            "wave_cycles_split": rec.get("wave_cycles_split"), "source": os.path.relpath(files[-1], ROOT)}


def self_launch(args):
    """`python bench.py --gpus N` with N > 1 and no launcher around it: start the N ranks as a child
    torch.distributed.run (fresh processes; this parent has not touched the GPU and never will), pass their single
    JSON line through and exit with their code."""
    import socket
    import subprocess
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus),
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("OMP_NUM_THREADS", "8")
    return subprocess.call(cmd, env=env)


def configs_measured(lib, ctx, ops, _lib):
    """configs[1] and configs[2] of BASELINE.json on this GPU, right after the headline's timed region (a few seconds in
    all): the numbers README / DESIGN quote for them, now on the driver's own line."""
    import treegp_amd
    from treegp_amd.synthetic import star_field, headline_invlam
    out = []
    iL = headline_invlam()
    # (i) configs[1]: 2-D AnisotropicRBF N = 8192, predict 32768 -- the ML-fit regime
    n, m, reps = 8192, 32768, 5
    X, y, y_err, Xs = star_field(n, m)
    spec = ops.KernelSpec(_lib.TGP_ARBF, amp=1.0, a=iL[0, 0], b=iL[0, 1], c=iL[1, 1])
    kc = spec.to_c()
    bufs = [ops.DeviceBuffer.from_array(ctx, a) for a in (X, y - y.mean(), y_err, Xs)]
    da, dys = ops.DeviceBuffer(ctx, n * 8), ops.DeviceBuffer(ctx, m * 8)
    ld, yd = C.c_double(), C.c_double()
    acc, samples = {}, {}
    # (no per-launch events here: they are for the headline's roofline and cost 3 % of a factorisation of this size --
    # 5.30 against 5.15 ms, profiles/r05_kernel_bench.jsonl `ms_with_per_launch_events`; the phases are medians of `reps`)
    lib.tgp_set_profiling(ctx, 0)
    for it in range(reps + 1):
        t0 = time.perf_counter()
        rc = lib.tgp_d_gp_solve(ctx, C.byref(kc), bufs[0].ptr, n, bufs[1].ptr, bufs[2].ptr, da.ptr, C.byref(ld), C.byref(yd), None)
        if rc != 0:
            raise RuntimeError("configs[1] solve rc=%d" % rc)
        tm = _lib.timings(ctx)
        lib.tgp_d_gp_predict(ctx, C.byref(kc), bufs[0].ptr, n, da.ptr, bufs[3].ptr, m, dys.ptr)
        tp = _lib.timings(ctx)[3]
        wall = (time.perf_counter() - t0) * 1e3
        if it == 0:
            continue                                       # warm-up (workspace allocation)
        for k, v in (("kbuild_ms", tm[0]), ("chol_ms", tm[1]), ("trsv_ms", tm[2]), ("predict_ms", tp), ("wall_ms", wall), ("sweeps", tm[10])):
            samples.setdefault(k, []).append(v)
    acc = {k: float(np.median(v)) for k, v in samples.items()}
    lib.tgp_set_profiling(ctx, 1)
    # likelihood-only evaluations (no alpha): what one step of an ML fit costs
    t0 = time.perf_counter()
    for _ in range(reps):
        lib.tgp_d_gp_solve(ctx, C.byref(kc), bufs[0].ptr, n, bufs[1].ptr, bufs[2].ptr, None, C.byref(ld), C.byref(yd), None)
    lik_ms = (time.perf_counter() - t0) * 1e3 / reps
    # the same evaluations four at a time, one context (one stream) each on the same resident data: what the finite-difference
    # gradient of the ML fit issues (treegp_amd/log_likelihood.py: parallel_fd; the reference takes them one after the other)
    from concurrent.futures import ThreadPoolExecutor
    ctxs = [_lib.new_ctx(int(os.environ.get("TGP_DEVICE", "0"))) for _ in range(4)]
    for c in ctxs:
        lib.tgp_set_lookahead(c, 0)

    def lik_loop(c):
        l2, y2 = C.c_double(), C.c_double()
        for _ in range(reps):
            lib.tgp_d_gp_solve(c, C.byref(kc), bufs[0].ptr, n, bufs[1].ptr, bufs[2].ptr, None, C.byref(l2), C.byref(y2), None)

    with ThreadPoolExecutor(4) as pool:
        list(pool.map(lik_loop, ctxs))                     # warm-up: each context's workspace
        t0 = time.perf_counter()
        list(pool.map(lik_loop, ctxs))
        lik4_ms = (time.perf_counter() - t0) * 1e3 / (4 * reps)
    for c in ctxs:
        lib.tgp_destroy(c)
    npad = (n + 255) // 256 * 256
    chol_tf = n ** 3 / 3.0 / (acc["chol_ms"] * 1e-3) / 1e12
    out.append({"config": "configs[1]: 2-D AnisotropicRBF N=8192, predict 32768, one GPU", "reps": reps, "phases": "medians, no per-launch events",
                "likelihood_evaluations_per_sec_4_contexts": 1e3 / lik4_ms,
                "ms": acc["wall_ms"], "phases_ms": {k: acc[k] for k in ("kbuild_ms", "chol_ms", "trsv_ms", "predict_ms")},
                "gp_solves_per_sec": 1e3 / (acc["kbuild_ms"] + acc["chol_ms"] + acc["trsv_ms"]),
                "likelihood_evaluations_per_sec": 1e3 / lik_ms,
                "cholesky_tflops_fp64": chol_tf, "cholesky_frac_mfma_peak": chol_tf / FP64_MFMA_PEAK_TFLOPS,
                "trsv_frac_hbm": (acc["sweeps"] * 4.0 * npad * (npad + 1)) / (acc["trsv_ms"] * 1e-3) / 8e12,      # L read once per sweep that ran
                "sweeps_per_solve": int(round(acc["sweeps"])),
                "predict_pairs_per_sec": float(m) * n / (acc["predict_ms"] * 1e-3),
                "predict_points_per_sec": m / (acc["predict_ms"] * 1e-3)})
    for b in bufs + [da, dys]:
        b.free()
    # the same configuration through the drop-in API (host arrays in, NumPy out)
    from treegp_amd.synthetic import headline_kernel_string
    api1 = api_route(X, y, y_err, Xs, headline_kernel_string(), passes=5)
    out[-1]["api_route_ms"] = api1
    # (0) configs[0]: the reference's own CPU-runnable case on the GPU, through the API (plumbing; the device is idle most of it)
    rng0 = np.random.default_rng(20240613)
    X0 = rng0.uniform(-10.0, 10.0, (512, 1))
    y0 = np.sin(X0[:, 0]) + 0.1 * rng0.standard_normal(512)
    e0 = 0.1 * rng0.uniform(0.8, 1.2, 512)
    api0 = api_route(X0, y0, e0, np.linspace(-10.0, 10.0, 1024)[:, None], "1.0**2 * AnisotropicRBF(scale_length=[2.0])", passes=5)
    out.insert(0, {"config": "configs[0]: 1-D AnisotropicRBF N=512, predict 1024, through the API", "api_route_ms": api0,
                   "points_per_sec": 1536.0 / (api0["total"] * 1e-3)})
    # (ii) configs[2]: 2-D VonKarman N = 32768 with the two-pcf hyper-parameter fit
    n = 32768
    X, y, y_err, Xs = star_field(n, n)
    vk = ops.KernelSpec(_lib.TGP_VK, amp=1.0, ell=0.1)
    dX, de = ops.DeviceBuffer.from_array(ctx, X), ops.DeviceBuffer.from_array(ctx, y_err)
    dA = ops.DeviceBuffer(ctx, lib.tgp_panel_elems(n) * 8)
    best = 1e9
    for _ in range(4):
        _lib.check(ctx, lib.tgp_d_kbuild_lower(ctx, C.byref(vk.to_c()), dX.ptr, n, de.ptr, dA.ptr), "kbuild")
        best = min(best, _lib.timings(ctx)[0])
    for b in (dX, de, dA):
        b.free()
    VK_CEILING = 1.745e11                                    # profiles/r02_vk_ceiling.txt
    elems = n * (n + 1) / 2.0
    gp = treegp_amd.GPInterpolation(kernel="1.0**2 * VonKarman(length_scale=0.1)", optimizer="two-pcf", nbins=20, normalize=True)
    gp.initialize(X, y, y_err)
    gp.solve()                                              # warm-up
    gp.initialize(X, y, y_err)
    t0 = time.perf_counter()
    gp.solve()
    fit_ms = (time.perf_counter() - t0) * 1e3
    t0 = time.perf_counter()
    gp.predict(Xs)
    solve_predict_ms = (time.perf_counter() - t0) * 1e3
    # the fused von Karman predict alone (random alpha: the arithmetic does not depend on it), best of 4, against the evaluator's
    # register-only ceiling with the table in LDS (profiles/r02_vk_ceiling.txt) -- one evaluation per (query, training point) pair
    alpha_r = np.random.default_rng(3).standard_normal(n)
    vk_pred_ms = 1e9
    for _ in range(4):
        ops.gp_predict(vk, X, alpha_r, Xs)
        vk_pred_ms = min(vk_pred_ms, _lib.timings(ctx)[3])
    vk_pairs = float(n) * float(len(Xs)) / (vk_pred_ms * 1e-3)
    roofline_predict_vk = {"bound": "valu", "achieved": vk_pairs, "peak": VK_CEILING, "unit": "pairs/s", "frac": vk_pairs / VK_CEILING,
                           "predict_ms": vk_pred_ms, "note": "N=32768 training, M=32768 query points; peak = register-only K_5/6 "
                           "evaluations/s of the same evaluator (tools/probes/vk_ceiling.hip), itself a measured figure of one box: the predict sits "
                           "on it to within the box-to-box spread of the clocks (+-2 %), so frac may read a hair over 1"}
    k = y - y.mean()
    w = 1.0 / y_err ** 2
    ops.kk_log(X[:, 0], X[:, 1], k, w, 1.0 / np.sqrt(n), 0.7, 20)
    kk_out = ops.kk_log(X[:, 0], X[:, 1], k, w, 1.0 / np.sqrt(n), 0.7, 20)
    kk_ms = _lib.timings(ctx)[4]
    ops.kk_twod(X[:, 0], X[:, 1], k, w, 0.0, 0.15, 21)
    kk2_out = ops.kk_twod(X[:, 0], X[:, 1], k, w, 0.0, 0.15, 21)
    kk2_ms = _lib.timings(ctx)[4]
    # pairs/s against a ceiling that bounds the shipped kernels.  Counted are the pairs that REACH a bin (sum of the
    # kernel's own pair counts; culled tiles and out-of-range pairs cost less and are left out: no ratio above 1 by construction).
    # VALU ceiling: VALU instructions per pair iteration on the common path (tools/kk_isa_count.py: Log 152, TwoD 52) at 4 clocks
    # per wave64 instruction on 256 x 4 SIMDs at 2.4 GHz.  LDS-atomic ceiling: 1.5e12 fp64 LDS atomics/s
    # (profiles/r02_lds_atomic_ceiling.txt: 5.0e11 pairs/s at three sums per pair) over the atomics a binned pair issues -- Log: five,
    # for the pairs below the four top bins only (those are summed in registers); TwoD: six (both mirrored pixels).
    VALU_SLOTS = 256 * 4 * 64 * 2.4e9 / 4.0
    LDS_ATOMICS = 1.5e12
    log_np = np.asarray(kk_out[4], dtype=np.float64)
    log_in = float(log_np.sum())
    log_low = float(log_np[:-4].sum()) / max(log_in, 1.0)
    log_rate = log_in / (kk_ms * 1e-3)
    log_ceil = min(VALU_SLOTS / 152.0, LDS_ATOMICS / max(5.0 * log_low, 1e-9))
    twod_in = float(np.asarray(kk2_out[2]).sum()) / 2.0                 # every pair lands in two (mirrored) pixels
    twod_rate = twod_in / (kk2_ms * 1e-3)
    twod_ceil = min(VALU_SLOTS / 52.0, LDS_ATOMICS / 6.0)
    roofline_kk = {
        "log": {"bound": "valu" if VALU_SLOTS / 152.0 <= LDS_ATOMICS / max(5.0 * log_low, 1e-9) else "lds_atomics",
                "achieved": log_rate, "peak": log_ceil, "unit": "binned pairs/s", "frac": log_rate / log_ceil,
                "binned_pairs": log_in, "ms_incl_copies": kk_ms, "fraction_below_the_register_bins": log_low,
                "valu_instructions_per_pair": 152, "note": "N=32768, 20 log bins in [1/sqrt(N), 0.7)"},
        "twod": {"bound": "valu" if VALU_SLOTS / 52.0 <= LDS_ATOMICS / 6.0 else "lds_atomics",
                 "achieved": twod_rate, "peak": twod_ceil, "unit": "binned pairs/s", "frac": twod_rate / twod_ceil,
                 "binned_pairs": twod_in, "ms_incl_copies": kk2_ms, "valu_instructions_per_pair": 52,
                 "note": "N=32768, 21 x 21 pixels, max_sep 0.15 (the anisotropic fit's binning); parity of this binning is "
                         "unpinned (TreeCorr absent)"}}
    rng = np.random.default_rng(610639139)
    idx = np.stack([rng.integers(0, n - 1, size=n) for _ in range(444)])
    ops.kk_twod_bootstrap(X[:, 0], X[:, 1], y, y_err, idx, 0.0, 0.15, 21)
    t0 = time.perf_counter()
    ops.kk_twod_bootstrap(X[:, 0], X[:, 1], y, y_err, idx, 0.0, 0.15, 21)
    boot_wall = (time.perf_counter() - t0) * 1e3
    boot_dev = _lib.timings(ctx)[4]
    out.append({"config": "configs[2]: 2-D VonKarman N=32768, two-pcf fit (nbins=20), one GPU",
                "kbuild_ms": best, "kbuild_elements_per_sec": elems / (best * 1e-3),
                "kbuild_frac_of_vk_ceiling": elems / (best * 1e-3) / VK_CEILING,
                "two_pcf_fit_ms": fit_ms, "solve_plus_predict_32768_ms": solve_predict_ms,
                "kk_log_ms_incl_copies": kk_ms, "kk_log_pairs_per_sec": n * (n - 1) / 2.0 / (kk_ms * 1e-3),
                # (round 4: the ratio to the one-LDS-atomic-per-sum rate was dropped -- the kernel accumulates the crowded bins in
                # registers and issues far fewer atomics, so that rate does not bound it; pairs/s stands alone)
                "roofline_kk": roofline_kk, "roofline_predict_vk": roofline_predict_vk,
                "bootstrap_444_resamples_21x21_ms": {"device_incl_copies": boot_dev, "wall": boot_wall}})
    return out


def api_route(X, y, y_err, Xs, kernel_string, passes=5):
    """initialize() + predict() of treegp_amd.GPInterpolation from host arrays, `passes` timed passes after one warm-up (the
    first pass sizes the host-boundary buffers): median milliseconds per phase, plus the spread.  `host_tax_ms` = wall time
    minus the device phases OF THE SAME CALLS (K build, Cholesky, sweeps, predict by the library's own events): what the
    Python layer, the host copies and the launch round trips add.  (Comparing with the timed loop's step instead would
    also count the clock state of a GPU that has been under load for ten more seconds: 2 - 3 % at the headline size.)"""
    import treegp_amd
    from treegp_amd import _lib
    gp1 = treegp_amd.GPInterpolation(kernel=kernel_string, optimizer="none", normalize=True, white_noise=0.0, backend="single")
    rows = []
    # as a user runs it: without the per-launch events of the roofline accounting (in that mode the factorisation ends with a
    # host loop over ~200 event pairs, 10 - 30 ms at the headline size that no device phase shows; TGP_HOST_PHASES=1 prints it)
    lib, ctx0 = _lib.load_library(), _lib.get_ctx()
    lib.tgp_set_profiling(ctx0, 0)
    for it in range(passes + 1):
        t0 = time.perf_counter()
        gp1.initialize(X, y, y_err)
        t1 = time.perf_counter()
        if it == passes and os.environ.get("TGP_BENCH_PROFILE_API") == "1":      # development: where the host time of a pass goes
            import cProfile
            import pstats
            pr = cProfile.Profile()
            pr.enable()
            gp1.predict(Xs)
            pr.disable()
            pstats.Stats(pr, stream=sys.stderr).sort_stats("tottime").print_stats(12)
        else:
            gp1.predict(Xs)
        t2 = time.perf_counter()
        tm = _lib.timings(_lib.get_ctx())
        if it:
            rows.append(((t1 - t0) * 1e3, (t2 - t1) * 1e3, (t2 - t0) * 1e3, tm[0] + tm[1] + tm[2] + tm[3]))
    lib.tgp_set_profiling(ctx0, 1)
    rows = np.array(rows)
    return {"initialize": float(np.median(rows[:, 0])), "predict": float(np.median(rows[:, 1])), "total": float(np.median(rows[:, 2])),
            "total_min": float(rows[:, 2].min()), "total_max": float(rows[:, 2].max()), "passes": passes,
            "device_phases_ms": float(np.median(rows[:, 3])), "host_tax_ms": float(np.median(rows[:, 2] - rows[:, 3]))}


def collective_backend():
    """torch.distributed backend of this run and the version of the library under it (RCCL for "nccl" on ROCm)"""
    import torch
    import torch.distributed as dist
    rec = {"backend": dist.get_backend(), "torch": torch.__version__, "hip": getattr(torch.version, "hip", None)}
    if rec["backend"] == "nccl":
        try:
            rec["rccl_version"] = ".".join(str(v) for v in torch.cuda.nccl.version())
        except Exception as ex:          # noqa: BLE001  (a version query must not cost the line)
            rec["rccl_version"] = "unavailable: %s" % type(ex).__name__
    return rec


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--ntrain", dest="n", type=int, default=65536, help="training points (default: headline 65536)")
    ap.add_argument("--mpredict", dest="m", type=int, default=0, help="prediction points (default 4 n)")
    ap.add_argument("--cpu-sample", type=int, default=8192, help="N of the CPU-baseline sample (0 = skip)")
    ap.add_argument("--meanify", action="store_true", help="configs[4]'s recipe: mean-function table (KNN-4) under the field")
    ap.add_argument("--api", action="store_true", help="one GPU: step through GPInterpolation too (always so with --gpus > 1)")
    ap.add_argument("--no-configs", action="store_true", help="skip the configs[1] / configs[2] measurements of the one-GPU line")
    args = ap.parse_args()
    n = args.n
    m = args.m or 4 * n
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if os.environ.get("TGP_ONE_DEVICE") == "1":
        local_rank = 0
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(self_launch(args))                   # before anything below initialises HIP in this process
    if args.gpus != world:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d" % (args.gpus, world))
    os.environ["TGP_DEVICE"] = str(local_rank)         # the process-wide context of this rank (KNN mean function, ...)

    from treegp_amd import _lib, ops
    from treegp_amd.synthetic import star_field, star_field_with_mean, headline_invlam, headline_kernel_string

    lib = _lib.load_library()
    ctx = _lib.get_ctx(device=local_rank)
    # TGP_BENCH_FORCE_DIST=1: the N > 1 code path (process group, engine, API route, per-rank diagnostics) with whatever
    # world size the launcher gave, including one rank -- a rehearsal of the driver's multi-GPU run on the real backend
    force_dist = os.environ.get("TGP_BENCH_FORCE_DIST") == "1" and "WORLD_SIZE" in os.environ
    use_dist = world > 1 or force_dist
    api = use_dist or args.api or args.meanify
    fits = None
    if args.meanify:
        import tempfile
        from treegp_amd.fits_io import write_bintable_row
        X, y, y_err, Xs, X0, y0 = star_field_with_mean(n, m)
        fits = os.path.join(tempfile.mkdtemp(), "mean_rank%d.fits" % rank)
        write_bintable_row(fits, {"COORDS0": X0, "PARAMS0": y0})      # the wire format of treegp/meanify.py:139-165
    else:
        X, y, y_err, Xs = star_field(n, m)
    iL = headline_invlam()
    spec = ops.KernelSpec(_lib.TGP_ARBF, amp=1.0, a=iL[0, 0], b=iL[0, 1], c=iL[1, 1])
    ymean = y.mean()

    engine = None
    if use_dist:
        import torch
        import torch.distributed as dist
        torch.cuda.set_device(local_rank)
        # TGP_DIST_BACKEND=gloo + TGP_ONE_DEVICE=1: rehearsal of this code path with several ranks
        # sharing one GPU (development boxes have one); the driver's runs use nccl = RCCL
        backend = os.environ.get("TGP_DIST_BACKEND", "nccl")
        if backend == "nccl":
            # the collectives sit on the factorisation's critical path (panel chain), the bulk update does not:
            # RCCL's own stream gets high priority like the look-ahead stream whose work it carries
            try:
                opts = dist.ProcessGroupNCCL.Options(is_high_priority_stream=True)
                dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank), pg_options=opts)
            except (AttributeError, TypeError):
                dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend)
        from treegp_amd import dist as tdist
        engine = tdist.enable(profile=True, min_n=0)

        def barrier():
            dist.barrier()
            torch.cuda.synchronize()
            lib.tgp_sync(ctx)
    else:
        def barrier():
            lib.tgp_sync(ctx)

    lib.tgp_set_profiling(ctx, 1)
    if not api:
        dX = ops.DeviceBuffer.from_array(ctx, X)
        dy = ops.DeviceBuffer.from_array(ctx, y - ymean)
        de = ops.DeviceBuffer.from_array(ctx, y_err)
        dXs = ops.DeviceBuffer.from_array(ctx, Xs)
        da = ops.DeviceBuffer(ctx, n * 8)
        dys = ops.DeviceBuffer(ctx, m * 8)
        ld, yd = C.c_double(), C.c_double()
        kc = spec.to_c()

        def step(acc):
            rc = lib.tgp_d_gp_solve(ctx, C.byref(kc), dX.ptr, n, dy.ptr, de.ptr, da.ptr, C.byref(ld), C.byref(yd), None)
            if rc != 0:
                raise RuntimeError("tgp_d_gp_solve rc=%d %s" % (rc, lib.tgp_last_error(ctx)))
            tm = _lib.timings(ctx)
            rc = lib.tgp_d_gp_predict(ctx, C.byref(kc), dX.ptr, n, da.ptr, dXs.ptr, m, dys.ptr)
            if rc != 0:
                raise RuntimeError("tgp_d_gp_predict rc=%d" % rc)
            tp = _lib.timings(ctx)[3]
            for k, v in (("kbuild_ms", tm[0]), ("chol_ms", tm[1]), ("trsv_ms", tm[2]), ("predict_ms", tp),
                         ("syrk_ms", tm[5]), ("syrk_launches", tm[6]), ("syrk_flops", tm[7]), ("kbuild_bytes", tm[8]),
                         ("sweeps", tm[10])):
                acc[k] = acc.get(k, 0.0) + v
    else:
        # the drop-in API itself (treegp/gp_interp.py:196-227, 143-194): initialize() takes the data -- mean function at
        # the stars, mean, noise -- and predict() builds K, factorises, solves and predicts; with --gpus > 1 on the
        # multi-GPU route (treegp_amd.dist: row-block-cyclic Cholesky, query points sharded), every rank the same calls
        import treegp_amd
        gp = treegp_amd.GPInterpolation(kernel=headline_kernel_string(), optimizer="none", normalize=True, white_noise=0.0,
                                        average_fits=fits, backend="dist" if use_dist else "single")

        def step(acc):
            t0 = time.perf_counter()
            gp.initialize(X, y, y_err)
            t1 = time.perf_counter()
            gp.predict(Xs)
            t2 = time.perf_counter()
            acc["api_initialize_ms"] = acc.get("api_initialize_ms", 0.0) + (t1 - t0) * 1e3
            acc["api_predict_ms"] = acc.get("api_predict_ms", 0.0) + (t2 - t1) * 1e3
            if engine is None:
                tm = _lib.timings(ctx)       # of the last calls on the process-wide context: the solve, then the predict
                for k, v in (("kbuild_ms", tm[0]), ("chol_ms", tm[1]), ("trsv_ms", tm[2]), ("predict_ms", tm[3]),
                             ("syrk_ms", tm[5]), ("syrk_launches", tm[6]), ("syrk_flops", tm[7]), ("kbuild_bytes", tm[8]),
                             ("sweeps", tm[10])):
                    acc[k] = acc.get(k, 0.0) + v

    for _ in range(args.warmup):
        step({})
    if engine is not None:
        engine.acc.clear()
    barrier()
    t0 = time.perf_counter()
    acc = {}
    for _ in range(args.steps):
        step(acc)
    barrier()
    dt = time.perf_counter() - t0
    per_rank = None
    if use_dist:
        import torch
        import torch.distributed as dist
        acc.update(engine.acc)
        dev = "cuda" if dist.get_backend() == "nccl" else "cpu"
        t = torch.tensor([dt], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
        seen = torch.ones(1, dtype=torch.float64, device=dev)
        dist.all_reduce(seen)
        ranks_seen = int(seen.item())
        # per-rank diagnostics (ms per step): where each rank's factorisation time went
        names = ("chol_ms", "syrk_ms", "chain_ms", "gather_wait_ms", "trsv_ms", "predict_ms", "bytes_received")
        mine = torch.zeros(world * len(names), dtype=torch.float64, device=dev)
        for i, k in enumerate(names):
            mine[rank * len(names) + i] = acc.get(k, 0.0) / args.steps
        dist.all_reduce(mine)
        tab = mine.cpu().numpy().reshape(world, len(names))
        per_rank = {("bulk_ms" if k == "syrk_ms" else k): [float(v) for v in tab[:, i]] for i, k in enumerate(names)}
    else:
        ranks_seen = 1

    if rank == 0:
        K = args.steps
        ms_step = dt / K * 1e3
        recipe = "configs[4] recipe (mean-function table by KNN-4 + y_err noise diagonal)" if args.meanify else \
                 "configs[3] problem (y_err noise diagonal)"
        out = {
            "metric": "GP solve+predict throughput, 2-D AnisotropicRBF N=%d" % n,
            "value": (n + m) * K / dt, "unit": "points/s", "n_gpus": world, "steps": K, "warmup": args.warmup,
            "ms_per_step": ms_step, "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
            "dtype": "f64", "data": "synthetic", "ranks_seen": ranks_seen,
            "config": {"workload": "%s: 2-D AnisotropicRBF star field, N=%d training points, solved end to end + fused "
                                   "predict of M=%d points; %s; %s"
                                   % (recipe, n, m, "one GPU" if not use_dist else ("row-block-cyclic over %d GPUs" % world if world > 1 else
                                                                                        "multi-GPU route rehearsed with one rank"),
                                      "through treegp_amd.GPInterpolation (initialize + predict)" if api else
                                      "device-resident C-ABI calls (tgp_d_gp_solve + tgp_d_gp_predict)"),
                       "n_train": n, "m_predict": m, "kernel": "1.0**2 * AnisotropicRBF(invLam=inv(L(0.05,0.2,0.1)))",
                       "parallelism": "single" if not use_dist else "rowcyclic%d" % world, "through_api": bool(api),
                       "meanify": bool(args.meanify)},
        }
        if per_rank is not None:
            np_ = (n + 255) // 256 * 256
            out["per_rank_ms_per_step"] = per_rank
            # bytes_received (per rank, per step) is COUNTED by the communicator from the tensors it was handed during the
            # factorisation (diagonal-block broadcasts, panel exchanges); the line below is the factorisation's ideal
            # volume 4 N^2 (G-1)/G -- the counted figure is larger by the padding of the gathered panels to the fullest rank's
            # block count and by the 768 KB broadcast per panel
            out["bytes_received_expected"] = 4.0 * np_ * np_ * (world - 1) / world if world > 1 else 0.0
            out["per_rank_note"] = ("chol_ms = factorisation wall on the rank's main stream; bulk_ms = its trailing-update kernels; "
                                    "chain_ms = its look-ahead stream (diagonal blocks, broadcasts, local solves, all-gathers, strips); "
                                    "gather_wait_ms = main stream stalled behind the chain between two bulk updates; "
                                    "bytes_received = counted from the tensors handed to the collectives")
            out["panel_chain"] = {"form": getattr(engine, "chain_form", "gather"),
                                  "note": "gather: the chain's strips wait for each all-gathered panel; bcast (TGP_DIST_CHAIN_BCAST=1): "
                                          "their column operand rides on the diagonal block's broadcast and the all-gathers feed the "
                                          "bulk update only -- the A/B for the first run on a real node"}
            out["gather_probe"] = getattr(engine.comm, "gather_probe", None)
            out["collective_backend"] = collective_backend()
            out["owner_map"] = "256-row blocks dealt block-cyclically, reflected every G blocks"
            out["cpu_baseline"] = "see n_gpus=1 line"
        if "api_predict_ms" in acc:
            out["api_ms_per_step"] = {"initialize": acc["api_initialize_ms"] / K, "predict": acc["api_predict_ms"] / K}
        if acc.get("syrk_ms", 0) > 0:
            ach = acc["syrk_flops"] / (acc["syrk_ms"] * 1e-3) / 1e12          # per GPU (rank 0's share when N > 1)
            launches = max(acc["syrk_launches"], 1)
            traffic, src = (None, "not collected for this configuration")
            alg_bytes = None
            if not use_dist:
                # HBM-side bytes per launch of this kernel for this exact workload, from separate rocprofv3 --pmc
                # passes on the shipped sources (FETCH_SIZE x2 for the 16-B/lane operand loads, WRITE_SIZE exact;
                # tools/pmc_bench.sh); algorithmic bytes per launch = C read + written once (16 B per updated element)
                # (per STEP in the PMC file -- the profiled run splits the launches the timed run fuses -- over this run's launches)
                traffic, src = measured_traffic(n, "syrk_hbm_bytes_per_step")
                if traffic is not None:
                    traffic = traffic / (launches / K)
                else:
                    traffic, src = measured_traffic(n, "syrk_hbm_bytes_per_launch")
                alg_bytes = 16.0 * (acc["syrk_flops"] / (2.0 * 1024.0)) / launches if n >= 18432 else None
            out["roofline"] = {"bound": "mfma", "achieved": ach, "peak": FP64_MFMA_PEAK_TFLOPS, "unit": "TFLOP/s",
                               "frac": ach / FP64_MFMA_PEAK_TFLOPS, "traffic": traffic, "traffic_source": src,
                               "algorithmic_bytes": alg_bytes,
                               "traffic_over_algorithmic": (traffic / alg_bytes) if (traffic and alg_bytes) else None,
                               "kernel": ("syrk_segs_kernel<4>" if n >= 18432 else "syrk_dtv_kernel<4, 2>") if not use_dist else "syrk_distn_kernel (rank 0)",
                               "launches": int(acc["syrk_launches"]),
                               "avg_launch_ms": acc["syrk_ms"] / launches}
            if not use_dist and n >= 18432:
                # `frac` is against the 2.4 GHz peak; the chip runs this kernel at ~2.3 GHz (DVFS) with the MFMA pipe
                # busy `mfma_busy_frac` of the cycles -- frac ~ busy x clock / 2.4 (PMC passes, profiles/r*_pmc_sq.json)
                out["roofline"]["pipe"] = measured_pipe()
        if "chol_ms" in acc:
            out["cholesky_tflops_fp64"] = (n ** 3 / 3.0) * K / (acc["chol_ms"] * 1e-3) / 1e12
            out["gp_solves_per_sec"] = K / ((acc["kbuild_ms"] + acc["chol_ms"] + acc["trsv_ms"]) * 1e-3)
            out["predict_points_per_sec"] = m * K / (acc["predict_ms"] * 1e-3)
            npad = (n + 255) // 256 * 256
            if not use_dist:
                kb_bytes = K * (8.0 * (npad * (npad + 1) / 2.0) + 16.0 * npad)
                out["kbuild_GBps"] = kb_bytes / (acc["kbuild_ms"] * 1e-3) / 1e9
                # K build is HBM-write bound, 8 Np(Np+1)/2 + 16 Np algorithmic bytes per launch
                kb_t, kb_src = measured_traffic(n, "kbuild_write_bytes")
                out["roofline_kbuild"] = {"bound": "hbm", "achieved": out["kbuild_GBps"], "peak": 8000.0, "unit": "GB/s",
                                          "frac": out["kbuild_GBps"] / 8000.0, "traffic": kb_t, "traffic_source": kb_src,
                                          "algorithmic_bytes": kb_bytes / K,
                                          "kernel": "kbuild_slab_kernel<GAUSS>", "avg_launch_ms": acc["kbuild_ms"] / K}
            # SURVEY 8(d) prices the solves at two reads of the packed factor (forward + backward sweep, ~8 Np^2 B).  When n is not a
            # multiple of 256 the right-hand side rides through the factorisation as an extra matrix row (L^-1 y comes out of it)
            # and only the backward sweep is left: ONE read of L.  The sweeps that really ran are counted by the library
            # (tgp_last_timings[10]); the multi-GPU route's replicated solve always runs both.
            sweeps = int(round(acc["sweeps"] / K)) if acc.get("sweeps", 0) > 0 else 2
            tr_bytes = sweeps * 8.0 * (npad * (npad + 1) / 2.0)
            tr_rate = tr_bytes * K / (acc["trsv_ms"] * 1e-3) / 1e9
            out["roofline_trsv"] = {"bound": "hbm", "achieved": tr_rate, "peak": 8000.0, "unit": "GB/s", "frac": tr_rate / 8000.0,
                                    "traffic": None, "algorithmic_bytes": tr_bytes, "sweeps_per_solve": sweeps,
                                    "kernel": "potrs: forward + backward sweep" if sweeps == 2 else "potrs: backward sweep (L^-1 y from the augmented row)",
                                    "avg_ms": acc["trsv_ms"] / K}
            # the fused predict is fp64-VALU-bound, 30 algorithmic flops per (query, training point) pair
            m_rank = m if world == 1 else -(-m // world)
            pr_tf = 30.0 * m_rank * n * K / (acc["predict_ms"] * 1e-3) / 1e12
            out["roofline_predict"] = {"bound": "valu", "achieved": pr_tf, "peak": 78.6, "unit": "TFLOP/s", "frac": pr_tf / 78.6,
                                       "pairs_per_sec": m_rank * n * K / (acc["predict_ms"] * 1e-3),
                                       "kernel": "predict_gauss_tab_kernel<256>", "avg_ms": acc["predict_ms"] / K,
                                       # 16.4 VALU instructions per pair in the loop (csrc/predict.hip): share of the issue slots
                                       # of 256 CUs x 4 SIMDs x 16 lanes at 2.4 GHz
                                       "valu_issue_frac_at_2p4ghz": m_rank * n * K / (acc["predict_ms"] * 1e-3) * 16.4 / (256 * 4 * 16 * 2.4e9)}
            out["phases_ms_per_step"] = {k: acc[k] / K for k in ("kbuild_ms", "chol_ms", "trsv_ms", "predict_ms", "syrk_ms")}
        if not use_dist and not api and not args.no_configs:
            # the same workloads once more through the drop-in API on this one GPU (host buffers in, NumPy out): what the
            # N > 1 lines, which always step through GPInterpolation, are to be compared with.  Medians of 5 passes after a
            # warm-up; `host_tax_ms` = wall minus the library's own device phases of the same calls (api_route).
            out["api_route_ms"] = {"headline": api_route(X, y, y_err, Xs, headline_kernel_string(), passes=5)}
            out["api_route_ms_per_step_one_gpu"] = {k: out["api_route_ms"]["headline"][k] for k in ("initialize", "predict", "total")}
        if not use_dist and not args.no_configs:
            out["configs_measured"] = configs_measured(lib, ctx, ops, _lib)
        if not use_dist and args.cpu_sample > 0:
            out["cpu_baseline"] = cpu_baseline(args.cpu_sample, 4 * args.cpu_sample, n)
        print(json.dumps(out), flush=True)
    if use_dist:
        import torch.distributed as dist
        dist.destroy_process_group()


if __name__ == "__main__":
    main()

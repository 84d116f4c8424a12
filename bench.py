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

FP64_MFMA_PEAK_TFLOPS = 78.6     # MI355X fp64 matrix peak


def cpu_baseline(n_s, m_s):
    """The reference's own SciPy call sequence on the host cores, on a bounded sample."""
    from oracle import cpu_reference_path as R
    from treegp_amd.synthetic import star_field, headline_invlam
    X, y, y_err, Xs = star_field(n_s, m_s)
    iL = headline_invlam()
    tm = {}
    R.solve_predict(X, y - y.mean(), y_err, Xs, iL, 1.0, timings=tm)       # single pass, first touch included
    threads = os.cpu_count()
    blas = "unknown BLAS"
    try:
        from threadpoolctl import threadpool_info
        pools = threadpool_info()
        threads = max([p.get("num_threads", 1) for p in pools] + [1])
        blas = ", ".join(sorted({"%s %s" % (p.get("internal_api", "?"), p.get("version", "")) for p in pools
                                 if p.get("user_api") == "blas"})) or blas
    except Exception:
        pass
    cpu = "unknown CPU"
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                cpu = line.split(":", 1)[1].strip()
                break
    except Exception:
        pass
    return {
        "value": (n_s + m_s) / tm["total"], "unit": "points/s", "cores": threads, "kind": "port",
        "sample": "same star field at N=%d train / M=%d predict, one pass of the reference's SciPy calls "
                  "(pdist+exp+squareform %.2fs, dpotrf %.2fs = %.1f GFLOP/s on %d BLAS threads, cho_solve %.3fs, "
                  "cdist+exp %.2fs single-thread); host: %s, %d logical CPUs, %s; the O(N^3) factorisation makes "
                  "points/s size-dependent"
                  % (n_s, m_s, tm["kbuild"], tm["cholesky"], n_s ** 3 / 3 / tm["cholesky"] / 1e9, threads,
                     tm["cho_solve"], tm["cross_kernel"], cpu, os.cpu_count() or 0, blas),
        "phases_s": {k: round(v, 4) for k, v in tm.items()},
    }


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--ntrain", dest="n", type=int, default=65536, help="training points (default: headline 65536)")
    ap.add_argument("--mpredict", dest="m", type=int, default=0, help="prediction points (default 4 n)")
    ap.add_argument("--cpu-sample", type=int, default=8192, help="N of the CPU-baseline sample (0 = skip)")
    args = ap.parse_args()
    n = args.n
    m = args.m or 4 * n
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if os.environ.get("TGP_ONE_DEVICE") == "1":
        local_rank = 0
    if args.gpus != world:
        if args.gpus > 1:
            raise SystemExit("launch with torch.distributed.run --nproc-per-node %d (WORLD_SIZE=%d)" % (args.gpus, world))

    from treegp_amd import _lib, ops
    from treegp_amd.synthetic import star_field, headline_invlam

    lib = _lib.load_library()
    ctx = _lib.get_ctx(device=local_rank)
    X, y, y_err, Xs = star_field(n, m)
    iL = headline_invlam()
    spec = ops.KernelSpec(_lib.TGP_ARBF, amp=1.0, a=iL[0, 0], b=iL[0, 1], c=iL[1, 1])
    ymean = y.mean()

    dist_solver = None
    if world > 1:
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
        from treegp_amd.dist import DistributedGP
        dist_solver = DistributedGP(ctx, spec, X, y - ymean, y_err, Xs, profile=True)

        def barrier():
            dist.barrier()
            torch.cuda.synchronize()
            lib.tgp_sync(ctx)
    else:
        def barrier():
            lib.tgp_sync(ctx)

    lib.tgp_set_profiling(ctx, 1)
    if world == 1:
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
                         ("syrk_ms", tm[5]), ("syrk_launches", tm[6]), ("syrk_flops", tm[7]), ("kbuild_bytes", tm[8])):
                acc[k] = acc.get(k, 0.0) + v
    else:
        def step(acc):
            dist_solver.step(acc)

    for _ in range(args.warmup):
        step({})
    barrier()
    t0 = time.perf_counter()
    acc = {}
    for _ in range(args.steps):
        step(acc)
    barrier()
    dt = time.perf_counter() - t0
    if world > 1:
        import torch
        import torch.distributed as dist
        t = torch.tensor([dt], dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())

    if rank == 0:
        K = args.steps
        ms_step = dt / K * 1e3
        out = {
            "metric": "GP solve+predict throughput, 2-D AnisotropicRBF N=%d" % n,
            "value": (n + m) * K / dt, "unit": "points/s", "n_gpus": world, "steps": K, "warmup": args.warmup,
            "ms_per_step": ms_step, "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
            "dtype": "f64", "data": "synthetic",
            "config": {"workload": "configs[3] problem (2-D AnisotropicRBF star field, N=%d training points, "
                                   "y_err noise diagonal) solved end to end + fused predict of M=%d points; "
                                   "%s" % (n, m, "one GPU" if world == 1 else "row-block-cyclic over %d GPUs" % world),
                       "n_train": n, "m_predict": m, "kernel": "1.0**2 * AnisotropicRBF(invLam=inv(L(0.05,0.2,0.1)))",
                       "parallelism": "single" if world == 1 else "rowcyclic%d" % world},
        }
        if acc.get("syrk_ms", 0) > 0:
            ach = acc["syrk_flops"] / (acc["syrk_ms"] * 1e-3) / 1e12          # per GPU (rank 0's share when N > 1)
            traffic = None
            pmc = os.path.join(ROOT, "profiles", "r01_v6_pmc_bench_n65536.json")
            if world == 1 and n == 65536 and os.path.exists(pmc):
                # HBM bytes per launch of this kernel for this exact workload, from separate rocprofv3 --pmc
                # passes (FETCH_SIZE, WRITE_SIZE; tools/pmc_bench.sh), calibration notes inside the file
                traffic = json.load(open(pmc))["_derived"]["syrk_hbm_bytes_per_launch"]
            out["roofline"] = {"bound": "mfma", "achieved": ach, "peak": FP64_MFMA_PEAK_TFLOPS, "unit": "TFLOP/s",
                               "frac": ach / FP64_MFMA_PEAK_TFLOPS, "traffic": traffic,
                               "kernel": ("syrk_segs_kernel<4>" if n >= 22528 else "syrk_dtv_kernel<4, 2>") if world == 1 else "syrk_distn_kernel (rank 0)",
                               "launches": int(acc["syrk_launches"]),
                               "avg_launch_ms": acc["syrk_ms"] / max(acc["syrk_launches"], 1)}
        if "chol_ms" in acc:
            out["cholesky_tflops_fp64"] = (n ** 3 / 3.0) * K / (acc["chol_ms"] * 1e-3) / 1e12
            out["gp_solves_per_sec"] = K / ((acc["kbuild_ms"] + acc["chol_ms"] + acc["trsv_ms"]) * 1e-3)
            out["predict_points_per_sec"] = m * K / (acc["predict_ms"] * 1e-3)
            if "kbuild_bytes" in acc:
                out["kbuild_GBps"] = acc["kbuild_bytes"] / (acc["kbuild_ms"] * 1e-3) / 1e9
            out["phases_ms_per_step"] = {k: acc[k] / K for k in ("kbuild_ms", "chol_ms", "trsv_ms", "predict_ms", "syrk_ms")}
        if world == 1 and args.cpu_sample > 0:
            out["cpu_baseline"] = cpu_baseline(args.cpu_sample, max(args.cpu_sample // 4, 1))
        print(json.dumps(out), flush=True)
    if world > 1:
        import torch.distributed as dist
        dist.destroy_process_group()


if __name__ == "__main__":
    main()

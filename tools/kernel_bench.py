#!/usr/bin/env python3
"""Per-kernel measurements at the BASELINE.json configuration sizes:
prints one JSON object per kernel family with its algorithmic work and achieved rate."""
import ctypes as C
import json
import sys
import time

import numpy as np

sys.path.insert(0, __file__.rsplit("/tools/", 1)[0])
from treegp_amd import _lib, ops  # noqa: E402
from treegp_amd.synthetic import star_field, headline_invlam  # noqa: E402

lib = _lib.load_library()
ctx = _lib.get_ctx()


def out(**kw):
    print(json.dumps(kw), flush=True)


def kbuild(kind, n, reps=12):
    X, y, y_err, _ = star_field(n, 16)
    iL = headline_invlam()
    if kind == "gauss":
        spec = ops.KernelSpec(_lib.TGP_ARBF, amp=1.0, a=iL[0, 0], b=iL[0, 1], c=iL[1, 1])
    elif kind == "vk":
        spec = ops.KernelSpec(_lib.TGP_VK, amp=1.0, ell=0.1)
    else:
        spec = ops.KernelSpec(_lib.TGP_AVK, amp=1.0, a=iL[0, 0], b=iL[0, 1], c=iL[1, 1])
    Np = lib.tgp_padded_n(n)
    dX = ops.DeviceBuffer.from_array(ctx, X); de = ops.DeviceBuffer.from_array(ctx, y_err)
    dA = ops.DeviceBuffer(ctx, lib.tgp_panel_elems(Np) * 8)
    best = 1e9
    for _ in range(reps):
        _lib.check(ctx, lib.tgp_d_kbuild_lower(ctx, C.byref(spec.to_c()), dX.ptr, n, de.ptr, dA.ptr), "kbuild")
        tm = _lib.timings(ctx)
        best = min(best, tm[0])
    elems = Np * (Np + 1) / 2
    out(kernel="kbuild_lower<%s>" % kind, n=n, ms=best, bytes=tm[8], GBps=tm[8] / best / 1e6, frac_hbm_8TBps=tm[8] / best / 1e6 / 8000,
        elements_per_s=elems / (best * 1e-3))
    for b in (dX, de, dA):
        b.free()


def solve_phases(n):
    X, y, y_err, Xs = star_field(n, 4 * n)
    iL = headline_invlam()
    spec = ops.KernelSpec(_lib.TGP_ARBF, amp=1.0, a=iL[0, 0], b=iL[0, 1], c=iL[1, 1])
    dX = ops.DeviceBuffer.from_array(ctx, X); dy = ops.DeviceBuffer.from_array(ctx, y - y.mean())
    de = ops.DeviceBuffer.from_array(ctx, y_err); dXs = ops.DeviceBuffer.from_array(ctx, Xs)
    da = ops.DeviceBuffer(ctx, n * 8); dys = ops.DeviceBuffer(ctx, 4 * n * 8)
    ld, yd = C.c_double(), C.c_double()
    # the factorisation's time as it ships (best of 4; the per-launch events of profiling mode sit on the bulk stream and cost a
    # chain-bound size ~8 %: 5.3 -> 5.8 ms at N = 8192), then one profiled pass for the bulk kernel's own time
    chol_ms = 1e9
    for _ in range(4):
        rc = lib.tgp_d_gp_solve(ctx, C.byref(spec.to_c()), dX.ptr, n, dy.ptr, de.ptr, da.ptr, C.byref(ld), C.byref(yd), None)
        assert rc == 0
        chol_ms = min(chol_ms, _lib.timings(ctx)[1])
    lib.tgp_set_profiling(ctx, 1)
    for _ in range(2):
        rc = lib.tgp_d_gp_solve(ctx, C.byref(spec.to_c()), dX.ptr, n, dy.ptr, de.ptr, da.ptr, C.byref(ld), C.byref(yd), None)
        assert rc == 0
        tm = _lib.timings(ctx)
        lib.tgp_d_gp_predict(ctx, C.byref(spec.to_c()), dX.ptr, n, da.ptr, dXs.ptr, 4 * n, dys.ptr)
        tp = _lib.timings(ctx)[3]
    lib.tgp_set_profiling(ctx, 0)
    Np = lib.tgp_padded_n(n)
    out(kernel="cholesky", n=n, ms=chol_ms, TFLOPs=n ** 3 / 3 / chol_ms / 1e9, frac_mfma_78_6_whole=n ** 3 / 3 / chol_ms / 1e9 / 78.6,
        ms_with_per_launch_events=tm[1], syrk_ms=tm[5], syrk_TFLOPs=tm[7] / tm[5] / 1e9, frac_mfma_78_6=tm[7] / tm[5] / 1e9 / 78.6)
    out(kernel="potrs(trsv)+logdet+dot", n=n, ms=tm[2], bytes_alg=8.0 * Np * Np, GBps=8.0 * Np * Np / tm[2] / 1e6,
        frac_hbm_8TBps=8.0 * Np * Np / tm[2] / 1e6 / 8000)
    pairs = 4.0 * n * n
    out(kernel="predict_partial<gauss>", n=n, m=4 * n, ms=tp, pairs_per_s=pairs / (tp * 1e-3), TFLOPs_30_per_pair=30 * pairs / tp / 1e9,
        frac_fp64_vector_78_6=30 * pairs / tp / 1e9 / 78.6, points_per_s=4 * n / (tp * 1e-3))
    for b in (dX, dy, de, dXs, da, dys):
        b.free()


def pair_binning(n, nboot):
    rng = np.random.default_rng(1)
    X = rng.uniform(0, 1, (n, 2)); yv = rng.standard_normal(n); yerr = rng.uniform(0.05, 0.1, n)
    k = yv - yv.mean(); w = 1 / yerr ** 2
    for name, f in (("kk_twod(nbins=21,max_sep=0.15)", lambda: ops.kk_twod(X[:, 0], X[:, 1], k, w, 0.0, 0.15, 21)),
                    ("kk_log(nbins=20)", lambda: ops.kk_log(X[:, 0], X[:, 1], k, w, 1.0 / np.sqrt(n), 0.7, 20))):
        f()
        f()
        ms = _lib.timings(ctx)[4]
        out(kernel=name, n=n, ms_incl_copies=ms, pairs_per_s=n * (n - 1) / 2 / (ms * 1e-3))
    idx = np.stack([np.random.default_rng(610639139 + i).integers(0, n - 1, n) for i in range(nboot)])
    t0 = time.perf_counter()
    ops.kk_twod_bootstrap(X[:, 0], X[:, 1], yv, yerr, idx, 0.0, 0.15, 21)
    ms = _lib.timings(ctx)[4]
    out(kernel="kk_twod_bootstrap(n_boot=%d)" % nboot, n=n, ms_incl_copies=ms, wall_ms=(time.perf_counter() - t0) * 1e3,
        pairs_per_s=nboot * n * (n - 1) / 2 / (ms * 1e-3))


def posterior_cov(n, m):
    X, y, y_err, Xs = star_field(n, m)
    iL = headline_invlam()
    spec = ops.KernelSpec(_lib.TGP_ARBF, amp=1.0, a=iL[0, 0], b=iL[0, 1], c=iL[1, 1])
    alpha, _, _, fac = ops.gp_solve(spec, X, y - y.mean(), y_err, keep=True)
    ops.gp_predict_cov(spec, fac, X, Xs)                                    # builds the factor's inverse slabs
    t0 = time.perf_counter()
    cov = ops.gp_predict_cov(spec, fac, X, Xs)
    dt = time.perf_counter() - t0
    tm = _lib.timings(fac._ctx)
    flops = 2.0 * m * n * (n / 2 + m)
    out(kernel="gp_predict_cov", n=n, m=m, wall_ms=dt * 1e3, device_ms=tm[3], transfer_ms=tm[9], device_TFLOPs=flops / tm[3] / 1e9,
        min_diag=float(np.diag(cov).min()))


if __name__ == "__main__":
    which = sys.argv[1:] or ["kbuild", "solve", "kk", "cov"]
    if "kbuild" in which:
        kbuild("gauss", 65536); kbuild("gauss", 32768); kbuild("vk", 32768); kbuild("avk", 32768)
    if "solve" in which:
        solve_phases(8192); solve_phases(16384); solve_phases(32768)
    if "kk" in which:
        pair_binning(32768, 444)
    if "cov" in which:
        posterior_cov(8192, 4096)
        posterior_cov(32768, 4096)

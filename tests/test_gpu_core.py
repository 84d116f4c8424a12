"""GPU parity tests of the C-ABI seams against the oracle (tests/ only use of oracle/)."""
import ctypes as C

import numpy as np
import pytest

from oracle import gp_oracle as O

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def tg():
    from treegp_amd import _lib, ops
    ctx = _lib.get_ctx()
    return _lib, ops, ctx


def _inv(size=0.05, e1=0.2, e2=0.1):
    return np.linalg.inv(O.correlation_length_matrix(size, e1, e2))


def _field(rng, n):
    X = rng.uniform(0, 1, (n, 2))
    y = np.sin(7 * X[:, 0]) * np.cos(5 * X[:, 1]) + 0.03 * rng.standard_normal(n)
    yerr = 0.03 * rng.uniform(0.8, 1.2, n)
    return X, y, yerr


def test_mfma_layout_and_tilemap(tg):
    """every lower-triangular tile is visited exactly once by the XCD-aware enumeration"""
    _lib, ops, ctx = tg
    lib = _lib.load_library()
    for T in (1, 2, 7, 8, 9, 30, 64, 100):
        cap = 4 * (T + 16) * (T + 16) + 4096
        ti = np.empty(cap, np.int32); tj = np.empty(cap, np.int32)
        g = lib.tgp_debug_tilemap(T, ti.ctypes.data_as(C.c_void_p), tj.ctypes.data_as(C.c_void_p), cap)
        assert g > 0
        ok = ti[:g] >= 0
        pairs = set(zip(ti[:g][ok].tolist(), tj[:g][ok].tolist()))
        assert len(pairs) == ok.sum() == T * (T + 1) // 2
        assert all(0 <= b <= a < T for a, b in pairs)


@pytest.mark.parametrize("kind", ["gauss", "vk", "avk"])
def test_kernel_matrix_vs_oracle(tg, kind):
    _lib, ops, ctx = tg
    rng = np.random.default_rng(1)
    X = rng.uniform(0, 1, (300, 2)); Y = rng.uniform(0, 1, (170, 2)); Y[:20] = X[:20]
    invL = _inv(0.3, 0.2, -0.1)
    kw = dict(amp=1.7, a=invL[0, 0], b=invL[0, 1], c=invL[1, 1], ell=0.4)
    spec = ops.KernelSpec({"gauss": _lib.TGP_ARBF, "vk": _lib.TGP_VK, "avk": _lib.TGP_AVK}[kind], **kw)
    np.testing.assert_allclose(ops.kernel_matrix(spec, X), O.kernel_matrix(kind, X, **kw), rtol=1e-12, atol=1e-13)
    np.testing.assert_allclose(ops.kernel_matrix(spec, Y, X), O.kernel_matrix(kind, Y, X, **kw), rtol=1e-12, atol=1e-13)


@pytest.mark.parametrize("n", [200, 512, 1000])
def test_kbuild_and_potrf_vs_numpy(tg, n):
    _lib, ops, ctx = tg
    lib = _lib.load_library()
    rng = np.random.default_rng(n)
    X, y, yerr = _field(rng, n)
    invL = _inv()
    spec = ops.KernelSpec(_lib.TGP_ARBF, amp=1.0, a=invL[0, 0], b=invL[0, 1], c=invL[1, 1])
    Np = lib.tgp_padded_n(n)
    dX = ops.DeviceBuffer.from_array(ctx, X); de = ops.DeviceBuffer.from_array(ctx, yerr)
    dA = ops.DeviceBuffer(ctx, lib.tgp_panel_elems(Np) * 8); dW = ops.DeviceBuffer(ctx, Np * 128 * 8)
    _lib.check(ctx, lib.tgp_d_kbuild_lower(ctx, C.byref(spec.to_c()), dX.ptr, n, de.ptr, dA.ptr), "kbuild")
    Kd = np.empty((n, n))
    _lib.check(ctx, lib.tgp_d_unpack_lower(ctx, dA.ptr, Np, n, Kd.ctypes.data_as(C.c_void_p)), "unpack")
    Kref = O.kernel_matrix("gauss", X, amp=1.0, a=invL[0, 0], b=invL[0, 1], c=invL[1, 1]) + np.diag(yerr ** 2)
    np.testing.assert_allclose(Kd, np.tril(Kref), rtol=1e-13, atol=1e-15)
    info = lib.tgp_d_potrf(ctx, dA.ptr, Np, dW.ptr)
    assert info == 0
    Ld = np.empty((n, n))
    _lib.check(ctx, lib.tgp_d_unpack_lower(ctx, dA.ptr, Np, n, Ld.ctypes.data_as(C.c_void_p)), "unpack")
    Lref = np.linalg.cholesky(Kref)
    np.testing.assert_allclose(Ld, Lref, rtol=0, atol=1e-11 * np.abs(Lref).max())
    np.testing.assert_allclose(Ld @ Ld.T, Kref, rtol=0, atol=1e-13 * n)
    # inverted diagonal blocks
    W = dW.to_array((Np // 128, 128, 128))
    L0 = Lref[:128, :128] if n >= 128 else None
    if L0 is not None:
        np.testing.assert_allclose(W[0] @ L0, np.eye(128), atol=1e-10)
    # triangular solves on the padded right-hand side
    b = np.zeros(Np); b[:n] = y
    db = ops.DeviceBuffer.from_array(ctx, b)
    _lib.check(ctx, lib.tgp_d_potrs(ctx, dA.ptr, dW.ptr, Np, db.ptr), "potrs")
    alpha = db.to_array((Np,))[:n]
    ref = np.linalg.solve(Kref, y)
    np.testing.assert_allclose(alpha, ref, rtol=0, atol=1e-9 * np.abs(ref).max())


@pytest.mark.parametrize("n,m", [(40, 33), (777, 1500), (2048, 4096)])
def test_gp_solve_predict_vs_oracle(tg, n, m):
    _lib, ops, ctx = tg
    rng = np.random.default_rng(n + m)
    X, y, yerr = _field(rng, n)
    Xs = rng.uniform(0, 1, (m, 2))
    invL = _inv()
    kw = dict(amp=0.9, a=invL[0, 0], b=invL[0, 1], c=invL[1, 1])
    spec = ops.KernelSpec(_lib.TGP_ARBF, **kw)
    alpha, logdet, ydota, _ = ops.gp_solve(spec, X, y, yerr)
    K = O.kernel_matrix("gauss", X, **kw)
    a_ref, ld_ref = O.gp_solve(K, y, yerr)
    np.testing.assert_allclose(alpha, a_ref, rtol=0, atol=1e-9 * np.abs(a_ref).max())
    np.testing.assert_allclose(logdet, ld_ref, rtol=1e-12)
    np.testing.assert_allclose(ydota, y @ a_ref, rtol=1e-10)
    yp = ops.gp_predict(spec, X, alpha, Xs)
    yp_ref = O.gp_predict(O.kernel_matrix("gauss", Xs, X, **kw), a_ref)
    # north-star tolerance: 1e-10 relative on predicted values
    np.testing.assert_allclose(yp, yp_ref, rtol=0, atol=1e-10 * np.abs(yp_ref).max())


def test_not_positive_definite_raises(tg):
    _lib, ops, ctx = tg
    rng = np.random.default_rng(0)
    X = rng.uniform(0, 1, (300, 2))
    spec = ops.KernelSpec(_lib.TGP_ARBF, amp=1.0, a=1e-4, b=0.0, c=1e-4)     # near-constant kernel, no noise
    with pytest.raises(np.linalg.LinAlgError):
        ops.gp_solve(spec, X, np.ones(300), np.zeros(300))


def test_golden_config1_and_aniso2d(tg, golden):
    _lib, ops, ctx = tg
    g = golden("g1_c1_rbf1d.npz")
    spec = ops.KernelSpec(_lib.TGP_ARBF, amp=1.0, a=0.25, b=0.0, c=0.0)
    mean = np.mean(g["y"])
    alpha, logdet, ydota, _ = ops.gp_solve(spec, g["X"], g["y"] - mean, g["y_err"])
    yp = ops.gp_predict(spec, g["X"], alpha, g["Xs"]) + mean
    np.testing.assert_allclose(yp, g["y_pred"], rtol=0, atol=1e-10 * np.abs(g["y_pred"]).max())
    ll = -0.5 * ydota - 0.5 * len(alpha) * np.log(2 * np.pi) - 0.5 * logdet
    np.testing.assert_allclose(ll, g["logL"], rtol=1e-11)
    g = golden("g2_aniso2d.npz")
    iL = g["invLam"]
    spec = ops.KernelSpec(_lib.TGP_ARBF, amp=float(g["amp"]), a=iL[0, 0], b=iL[0, 1], c=iL[1, 1])
    mean = np.mean(g["y"])
    alpha, _, _, _ = ops.gp_solve(spec, g["X"], g["y"] - mean, g["y_err_eff"])
    yp = ops.gp_predict(spec, g["X"], alpha, g["Xs"]) + mean
    np.testing.assert_allclose(yp, g["y_pred"], rtol=0, atol=1e-10 * np.abs(g["y_pred"]).max())


@pytest.mark.parametrize("env", [{"TGP_SMALL_T": "0", "TGP_SMALL_ROWS": "0"},
                                 {"TGP_CHOL_MODE": "0"}, {"TGP_CHOL_MODE": "1"}, {"TGP_CHOL_MODE": "2"},
                                 {"TGP_CHOL_MODE": "3", "TGP_QUAD_TAIL_TILES": "0"}, {"TGP_CHOL_MODE": "3", "TGP_QUAD_TAIL_TILES": "8"},
                                 {"TGP_CHOL_MODE": "3", "TGP_SMALL_T": "0", "TGP_SMALL_ROWS": "0"}, {"TGP_PREDICT_GENERIC": "1"},
                                 {"TGP_QUEUE_T": "0"}, {"TGP_QUEUE_T": "200"}, {"TGP_STRIP64_T": "0"}, {"TGP_NO_AUGMENT": "1"},
                                 {"TGP_QUEUE_T1": "0"}, {"TGP_QUEUE_T1": "200", "TGP_QUEUE_T2": "0"}, {"TGP_QUEUE_T1": "200", "TGP_QUEUE_T2": "200"},
                                 {"TGP_U2A_SPLIT_T": "0"}, {"TGP_STRIP32_T": "0"}, {"TGP_U2A_SPLIT_T": "200", "TGP_STRIP32_T": "200"},
                                 {"TGP_U2A_SPLIT_T": "200", "TGP_SYNC_EVENTS": "1"}, {"TGP_U2A_SPLIT_T": "200", "TGP_PANEL_MID": "0"},
                                 {"TGP_PREDICT_EXP": "0"}, {"TGP_PREDICT_EXP": "32"}, {"TGP_PREDICT_EXP": "64"},
                                 {"TGP_NO_AUGMENT_ALPHA": "1"}, {"TGP_SYNC_EVENTS": "1"}, {"TGP_SYNC_EVENTS": "0"},
                                 {"TGP_SYNC_EVENTS": "1", "TGP_CHOL_MODE": "3", "TGP_QUAD_TAIL_TILES": "8"},
                                 {"TGP_CHOL_MODE": "3", "TGP_QUAD_TAIL_TILES": "8", "TGP_FLAG_SEQ_START": "4294967274"},
                                 {"TGP_FLAG_SEQ_START": "4294967274"}])      # the hand-off sequence numbers wrap during this solve
def test_alternative_kernel_paths_agree(env):
    """The A/B switches kept in the library (schedules, tile thresholds, hand-off mechanism) must stay correct:
    each one solves and predicts the same problem in a fresh process (the switches are read once per process)."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = r'''
import sys, numpy as np
sys.path.insert(0, %r)
from treegp_amd import _lib, ops
from oracle import gp_oracle as O
rng = np.random.default_rng(3)
n, m = 3000, 500
X = rng.uniform(0, 1, (n, 2)); y = np.sin(5 * X[:, 0]) + 0.1 * rng.standard_normal(n); e = rng.uniform(0.05, 0.1, n)
Xs = rng.uniform(0, 1, (m, 2))
spec = ops.KernelSpec(_lib.TGP_ARBF, amp=1.3, a=60.0, b=8.0, c=45.0)
kw = dict(amp=1.3, a=60.0, b=8.0, c=45.0)
alpha, logdet, _, _ = ops.gp_solve(spec, X, y, e)
a_ref, ld_ref = O.gp_solve(O.kernel_matrix("gauss", X, **kw), y, e)
yp = ops.gp_predict(spec, X, alpha, Xs)
yp_ref = O.kernel_matrix("gauss", Xs, X, **kw) @ a_ref
assert np.abs(alpha - a_ref).max() <= 1e-10 * np.abs(a_ref).max(), np.abs(alpha - a_ref).max()
assert abs(logdet - ld_ref) <= 1e-11 * abs(ld_ref)
assert np.abs(yp - yp_ref).max() <= 1e-10 * np.abs(yp_ref).max()
print("OK")
''' % root
    r = subprocess.run([sys.executable, "-c", code], env=dict(os.environ, **env), capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and "OK" in r.stdout, (env, r.stdout[-500:], r.stderr[-1500:])


def test_concurrent_contexts_agree_with_single_solves():
    """Several contexts factorising at once (the ML fit's concurrent finite differences).  The bulk update of chain-bound
    steps is a persistent grid whose workgroups leave the compute units it keeps clear for the panel chain; with other
    contexts' kernels on the chip it must still get every tile done (it once did not: all its workgroups could land on
    the units another context kept clear)."""
    from concurrent.futures import ThreadPoolExecutor
    from treegp_amd import _lib, ops
    from treegp_amd.synthetic import star_field, headline_invlam
    n, K = 4096, 5
    iL = headline_invlam()
    specs = [ops.KernelSpec(_lib.TGP_ARBF, amp=1.0, a=iL[0, 0] * (1 + 0.01 * i), b=iL[0, 1], c=iL[1, 1]) for i in range(K)]
    X, y, ye, _ = star_field(n, 16)
    y = y - y.mean()
    ref = [ops.gp_solve(s, X, y, ye, want_alpha=False)[1:3] for s in specs]
    ctxs = [_lib.new_ctx(0) for _ in range(K)]
    with ThreadPoolExecutor(K) as pool:
        for _ in range(6):
            got = list(pool.map(lambda i: ops.gp_solve(specs[i], X, y, ye, want_alpha=False, ctx=ctxs[i])[1:3], range(K)))
            for g, r in zip(got, ref):
                np.testing.assert_allclose(g, r, rtol=1e-11)


@pytest.mark.parametrize("n", [100, 255, 257, 700, 2049, 3000])
def test_likelihood_only_solve_matches_full_solve(tg, n):
    """Without alpha (and n not a multiple of 256) y rides through the factorisation as an extra matrix row and the
    quadratic form is read off that row -- no triangular sweep.  Same numbers as the full solve and as the oracle."""
    from treegp_amd import _lib, ops
    rng = np.random.default_rng(n)
    X = rng.uniform(0, 1, (n, 2)); y = rng.standard_normal(n) * 3.0 + 0.5; e = rng.uniform(0.05, 0.15, n)
    spec = ops.KernelSpec(_lib.TGP_ARBF, amp=1.7, a=80.0, b=10.0, c=60.0)
    alpha, logdet, ydota, _ = ops.gp_solve(spec, X, y, e)
    _, logdet2, chi2, _ = ops.gp_solve(spec, X, y, e, want_alpha=False)
    np.testing.assert_allclose(chi2, ydota, rtol=1e-11)
    np.testing.assert_allclose(logdet2, logdet, rtol=1e-13)
    a_ref, ld_ref = O.gp_solve(O.kernel_matrix("gauss", X, amp=1.7, a=80.0, b=10.0, c=60.0), y, e)
    np.testing.assert_allclose(chi2, float(y @ a_ref), rtol=1e-10)
    np.testing.assert_allclose(logdet2, ld_ref, rtol=1e-11)


@pytest.mark.parametrize("n", [2049, 2303, 3000, 5000, 8000, 12289])
def test_alpha_from_the_augmented_row_matches_both_sweeps(tg, n):
    """n not a multiple of 256, alpha wanted, factor not kept: y rides through the factorisation as matrix row Np-1 and
    only the backward sweep runs (treegp/gp_interp.py:181-182 is cholesky + cho_solve = both sweeps).  Same alpha as the
    kept-factor path, which keeps a clean factor and does both sweeps, and as the oracle."""
    from treegp_amd import _lib, ops
    rng = np.random.default_rng(n)
    X = rng.uniform(0, 1, (n, 2)); y = rng.standard_normal(n) * 2.0 - 0.3; e = rng.uniform(0.05, 0.15, n)
    spec = ops.KernelSpec(_lib.TGP_ARBF, amp=1.4, a=90.0, b=12.0, c=70.0)
    alpha, logdet, ydota, _ = ops.gp_solve(spec, X, y, e)
    assert _lib.timings(_lib.get_ctx())[10] == 1.0                 # one sweep over L in the solve phase
    alpha2, logdet2, ydota2, fac = ops.gp_solve(spec, X, y, e, keep=True)
    assert _lib.timings(_lib.get_ctx())[10] == 2.0
    fac.free()
    np.testing.assert_allclose(alpha, alpha2, rtol=0, atol=1e-12 * np.abs(alpha2).max())
    np.testing.assert_allclose(logdet, logdet2, rtol=1e-13)
    np.testing.assert_allclose(ydota, ydota2, rtol=1e-11)
    if n <= 5000:
        a_ref, ld_ref = O.gp_solve(O.kernel_matrix("gauss", X, amp=1.4, a=90.0, b=12.0, c=70.0), y, e)
        np.testing.assert_allclose(alpha, a_ref, rtol=0, atol=1e-10 * np.abs(a_ref).max())
        np.testing.assert_allclose(logdet, ld_ref, rtol=1e-11)


def test_predict_larger_than_the_pinned_mirror(tg):
    """tgp_gp_predict copies through a pinned mirror of its staging arena (csrc/api.hip: h2d / d2h_sync), capped at 256 MB;
    a query set whose arena is larger takes the direct path for what lies beyond the mirror -- same values either way."""
    _lib, ops, ctx = tg
    rng = np.random.default_rng(5)
    n, m = 300, 15_000_000                     # arena: 2 n + n + 2 m + m doubles = 360 MB
    X, y, yerr = _field(rng, n)
    invL = _inv()
    spec = ops.KernelSpec(_lib.TGP_ARBF, amp=1.1, a=invL[0, 0], b=invL[0, 1], c=invL[1, 1])
    alpha, _, _, _ = ops.gp_solve(spec, X, y, yerr)
    Xs = rng.uniform(0, 1, (m, 2))
    yp = ops.gp_predict(spec, X, alpha, Xs)
    assert yp.shape == (m,) and np.isfinite(yp).all()
    for lo in (0, m // 2 - 500, m - 1000):     # pieces small enough to go through the mirror entirely
        ref = ops.gp_predict(spec, X, alpha, Xs[lo:lo + 1000])
        np.testing.assert_allclose(yp[lo:lo + 1000], ref, rtol=0, atol=1e-12 * np.abs(ref).max())   # (the split of the sum over the training points follows m)


def test_released_factor_memory_is_reused():
    """A kept factor that is dropped hands its packed matrix back to the context (tgp_factor_release): the next solve of
    the same size allocates nothing; Factor.free(keep_memory=False) returns the memory to the device."""
    from treegp_amd import _lib, ops
    import torch

    def free_bytes():                              # device-wide figure of the driver, whichever runtime copy asks
        return torch.cuda.mem_get_info(0)[0]

    ctx = _lib.new_ctx(0)
    try:
        rng = np.random.default_rng(11)
        n = 6000                                   # packed factor ~ 150 MB: far above the allocator's granularity
        X, y, yerr = _field(rng, n)
        invL = _inv()
        spec = ops.KernelSpec(_lib.TGP_ARBF, amp=1.0, a=invL[0, 0], b=invL[0, 1], c=invL[1, 1])
        a1, _, _, fac = ops.gp_solve(spec, X, y, yerr, keep=True, ctx=ctx)
        with_one = free_bytes()
        fac.free()                                 # -> the context's cache
        a2, _, _, fac2 = ops.gp_solve(spec, X, y, yerr, keep=True, ctx=ctx)
        assert abs(free_bytes() - with_one) < (8 << 20)          # the second factor lives where the first did
        np.testing.assert_array_equal(a1, a2)
        fac2.free(keep_memory=False)               # -> the device
        assert free_bytes() > with_one + (100 << 20)
    finally:
        _lib.load_library().tgp_destroy(ctx)


def test_a_solve_has_the_same_bits_with_and_without_panel_mid_kernel():
    """panel_mid_kernel runs only for a solve that is alone in its process (its workgroups wait for each other inside the
    launch); what a solve returns must not depend on that.  The launch-by-launch form of the same step therefore does the same
    arithmetic (16-row slices for rows 128..255, the same tile form for the rows below): alpha and the log-determinant are
    bit-identical with TGP_PANEL_MID=0 and 1, at sizes that cover the one-stream, look-ahead and chain-bound schedules."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = ("import sys, hashlib, numpy as np; sys.path.insert(0, %r)\n"
            "from treegp_amd import _lib, ops\n"
            "from treegp_amd.synthetic import star_field, headline_invlam\n"
            "iL = headline_invlam(); spec = ops.KernelSpec(_lib.TGP_ARBF, amp=1.0, a=iL[0,0], b=iL[0,1], c=iL[1,1])\n"
            "for n in (700, 1300, 3000, 5000, 8192):\n"
            "    X, y, ye, _ = star_field(n, 16, seed=n)\n"
            "    a, ld, yd, _ = ops.gp_solve(spec, X, y - y.mean(), ye)\n"
            "    print(n, hashlib.sha256(a.tobytes()).hexdigest(), repr(ld))\n" % root)
    outs = []
    for mid in ("0", "1"):
        r = subprocess.run([sys.executable, "-c", code], env=dict(os.environ, TGP_PANEL_MID=mid), capture_output=True, text=True, timeout=300)
        assert r.returncode == 0, r.stderr[-2000:]
        outs.append([ln for ln in r.stdout.splitlines() if ln and ln[0].isdigit()])
    assert len(outs[0]) == 5 and outs[0] == outs[1], (outs[0], outs[1])


def test_staged_latency_tile_has_the_bits_of_the_direct_one():
    """nt_slice_tile (operands through LDS with whole-row loads: what the panel chain's 16-row slices run on since round 5) must
    return, value for value, what nt_small_tile (operands straight from global memory in MFMA layout) returns: same matrix
    instructions in the same order.  The stand-alone probe runs both on cold operands for the three shapes the chain uses and
    counts the values that differ."""
    import os
    import re
    import shutil
    import subprocess
    import tempfile
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    with tempfile.TemporaryDirectory() as tmp:
        exe = os.path.join(tmp, "slice_probe")
        r = subprocess.run([hipcc, "-O3", "-std=c++17", "--offload-arch=gfx950", "-I", os.path.join(root, "treegp_amd", "csrc"), "-o", exe,
                            os.path.join(root, "tools", "probes", "slice_probe.hip")], capture_output=True, text=True, timeout=600)
        assert r.returncode == 0, r.stderr[-2000:]
        r = subprocess.run([exe], capture_output=True, text=True, timeout=120)
        assert r.returncode == 0, r.stderr[-2000:]
    counts = [int(m) for m in re.findall(r"(\d+) of \d+ values differ", r.stdout)]
    assert len(counts) == 3 and counts == [0, 0, 0], r.stdout

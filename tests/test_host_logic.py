"""CPU-only checks: C-ABI surface, host-side mirror of the reference interface (no compute)."""
import os
import re

import numpy as np
import pytest

import treegp_amd as tg
from treegp_amd import _lib
from treegp_amd.fits_io import read_bintable_row, write_bintable_row

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_cabi_exports_every_declared_symbol():
    hdr = open(os.path.join(ROOT, "include", "tgp.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    declared = set(re.findall(r"\b(tgp_[a-z0-9_]+)\s*\(", hdr))
    assert len(declared) >= 30
    lib = _lib.load_library()
    for name in declared:
        assert hasattr(lib, name), "libtgp.so does not export %s" % name
    assert declared == set(_lib.SIGNATURES), declared ^ set(_lib.SIGNATURES)
    assert b"gfx950" in lib.tgp_version()
    # layout helpers are pure host functions
    assert lib.tgp_padded_n(1) == 256 and lib.tgp_padded_n(256) == 256 and lib.tgp_padded_n(257) == 512
    assert lib.tgp_panel_elems(512) == 512 * 256 + 256 * 256
    assert lib.tgp_panel_off(1, 512) == 512 * 256


def test_no_gpu_fails_loudly():
    lib = _lib.load_library()
    if lib.tgp_device_count() > 0:
        pytest.skip("a GPU is present")
    with pytest.raises(RuntimeError, match="no CPU path"):
        _lib.get_ctx()
    k = tg.eval_kernel("RBF(1)")
    gp = tg.GPInterpolation(kernel="1.0**2 * AnisotropicRBF(scale_length=[2.0])", optimizer="none")
    X = np.random.rand(8, 1)
    gp.initialize(X, X[:, 0])
    with pytest.raises(RuntimeError):
        gp.predict(X)


def test_kernel_strings_and_theta(golden):
    g = golden("g4_kernels.npz")
    for tag in ("rbf", "arbf", "vk", "avk", "vk_noamp"):
        k = tg.eval_kernel(str(g[tag + "_str"]))
        np.testing.assert_allclose(k.theta, g[tag + "_theta"], rtol=1e-14)
        k2 = k.clone_with_theta(k.theta)
        np.testing.assert_allclose(k2.theta, k.theta, rtol=1e-13)
    k = tg.eval_kernel(str(g["arbf_str"]))
    s = tg.kernel_to_spec(k)
    iL = g["arbf_invLam"]
    assert (s.kind, s.amp) == (_lib.TGP_ARBF, 0.25)
    np.testing.assert_allclose([s.a, s.b, s.c], [iL[0, 0], iL[0, 1], iL[1, 1]], rtol=1e-15)
    # theta <-> invLam round trip (tests/test_kernels.py:60-66)
    theta = k.theta[1:]
    L1 = np.zeros((2, 2)); L1[np.diag_indices(2)] = np.exp(theta[:2]); L1[np.tril_indices(2, -1)] = theta[2:]
    np.testing.assert_allclose(L1 @ L1.T, iL, atol=1e-12)
    s = tg.kernel_to_spec(tg.eval_kernel("2.0**2 * VonKarman(length_scale=3.0)"))
    assert (s.kind, s.amp, s.ell) == (_lib.TGP_VK, 4.0, 3.0)
    s = tg.kernel_to_spec(tg.eval_kernel("RBF(0.5)"))
    assert s.kind == _lib.TGP_RBF and s.a == s.c == 4.0 and s.b == 0.0
    s = tg.kernel_to_spec(tg.eval_kernel("1.0**2 * AnisotropicRBF(scale_length=[2.0])"))
    assert (s.a, s.b, s.c) == (0.25, 0.0, 0.0)
    # clone resets bounds to the default (-5, 5) because get_params only returns invLam
    kb = tg.AnisotropicRBF(scale_length=[1.0, 2.0], bounds=(-3, 3))
    assert np.all(kb.bounds == [-3, 3]) and np.all(kb.clone_with_theta(kb.theta).bounds == [-5, 5])


def test_kernel_to_spec_only_describes_what_it_evaluates():
    """Matern derives from RBF in scikit-learn and must not be taken for one; sums, white noise, rational quadratic and
    products of two non-constant kernels go to the dense path (NotImplementedError here)."""
    for s in ("1.0**2 * Matern(length_scale=0.3, nu=1.5)", "Matern(0.3)", "1.0**2 * RBF(0.5) + WhiteKernel(1e-3)",
              "RationalQuadratic(0.2, 0.7)", "RBF(0.3) * VonKarman(0.2)"):
        with pytest.raises(NotImplementedError):
            tg.kernel_to_spec(tg.eval_kernel(s))
    assert tg.kernel_to_spec(tg.eval_kernel("2.0**2 * RBF(0.5)")).amp == 4.0


def test_error_behaviour():
    with pytest.raises(TypeError):
        tg.GPInterpolation(kernel=tg.eval_kernel("RBF(1)"))                    # gp_interp.py:87-89
    with pytest.raises(ValueError):
        tg.GPInterpolation(kernel="RBF(1)", optimizer="nope")                  # gp_interp.py:91-95
    with pytest.raises(RuntimeError):
        tg.eval_kernel("NotAKernel(1)")                                        # kernels.py:51-55
    with pytest.raises(TypeError):
        tg.AnisotropicRBF(invLam=np.eye(2), scale_length=[1, 1])               # kernels.py:97-100
    with pytest.raises(NotImplementedError):
        tg.kernel_to_spec(tg.eval_kernel("RBF(1) + WhiteKernel(1)"))
    with pytest.raises(ValueError):
        tg.two_pcf(np.zeros((4, 3)), np.zeros(4), np.zeros(4), 0.1, 1.0)       # two_pcf.py:243-247
    from treegp_amd.two_pcf import get_kernel_class, get_correlation_length_matrix
    with pytest.raises(ValueError):
        get_kernel_class(tg.eval_kernel("RBF(1)"))
    assert get_kernel_class(tg.eval_kernel("2.0 * AnisotropicVonKarman(scale_length=[1., 1.])")) is tg.AnisotropicVonKarman
    with pytest.raises(ValueError):
        get_correlation_length_matrix(1.0, 1.5, 0.0)


def test_initialize_semantics():
    rng = np.random.default_rng(0)
    X = rng.uniform(0, 1, (50, 2)); y = rng.standard_normal(50) + 3.0; e = 0.1 * np.ones(50)
    gp = tg.GPInterpolation(kernel="1.0**2 * AnisotropicRBF(scale_length=[0.2, 0.2])", optimizer="none", white_noise=0.05)
    gp.initialize(X, y, y_err=e)
    np.testing.assert_allclose(gp._y_err, np.sqrt(e ** 2 + 0.05 ** 2))          # gp_interp.py:217-219
    assert gp._mean == np.mean(y) and gp._alpha is None
    assert np.all(gp._spatial_average == 0) and np.all(gp._X0 == 0)             # gp_interp.py:212-215
    gp2 = tg.GPInterpolation(kernel="RBF(1)", optimizer="none", normalize=False)
    gp2.initialize(X, y)
    assert gp2._mean == 0.0 and np.all(gp2._y_err == 0)
    assert gp.kernel is not gp.kernel_template


def test_two_pcf_geometry_and_bootstrap_stream(golden):
    g = golden("g7_host_scalars.npz")
    X = np.random.default_rng(1).uniform(0, 1, (10, 2))
    for nb in (15, 20, 21):
        t = tg.two_pcf(X, np.zeros(10), np.zeros(10), 0.0, 0.3, nbins=nb, anisotropic=True)
        mask, dist = t._twod_geometry()
        assert np.array_equal(mask, g["mask_%d" % nb])
        bs = 0.6 / nb
        np.testing.assert_allclose(dist[:nb, 0], -0.3 + bs * (np.arange(nb) + 0.5), rtol=1e-13, atol=1e-15)   # dx varies fastest
        np.testing.assert_allclose(dist[::nb, 1], -0.3 + bs * (np.arange(nb) + 0.5), rtol=1e-13, atol=1e-15)
        np.testing.assert_allclose(dist, -dist[::-1], atol=1e-15)
    t = tg.two_pcf(X, np.arange(10.0), np.ones(10), 0.0, 0.3)
    rows = np.stack([t._bootstrap_index() for _ in range(4)])
    assert np.array_equal(rows, g["boot_n10"])
    t = tg.two_pcf(X, np.arange(10.0), np.ones(10), 0.0, 0.3)
    u, v, yy, ee = t.resample_bootstrap()
    assert np.array_equal(yy, np.arange(10.0)[g["boot_n10"][0]])
    # 1-D inputs get a zero second coordinate (two_pcf.py:250-251)
    t1 = tg.two_pcf(X[:, :1], np.zeros(10), np.zeros(10), 0.1, 1.0)
    assert t1.X.shape == (10, 2) and np.all(t1.X[:, 1] == 0)


def test_fits_reader_roundtrip(tmp_path, golden):
    g = golden("g6_meanify.npz")
    p = str(tmp_path / "mean.fits")
    write_bintable_row(p, {"COORDS0": g["X0"], "PARAMS0": g["y0"]})
    d = read_bintable_row(p)
    assert np.array_equal(d["COORDS0"], g["X0"]) and np.array_equal(d["PARAMS0"], g["y0"])
    assert os.path.getsize(p) % 2880 == 0
    gp = tg.GPInterpolation(kernel="RBF(1)", optimizer="none", average_fits=p)
    assert gp._X0.shape == (2500, 2) and gp._y0.shape == (2500,)


def test_device_bessel_function_on_host():
    """bessel_k56.h compiled with g++ against scipy.special.kv (the reference's dependency)."""
    import ctypes
    import subprocess
    import tempfile
    from scipy import special
    d = tempfile.mkdtemp()
    src = os.path.join(d, "t.cpp")
    open(src, "w").write('#include "%s"\nextern "C" void vk(const double* u, double* o, long n)'
                         '{ for (long i = 0; i < n; ++i) o[i] = vonkarman_unit(u[i]); }\n'
                         % os.path.join(ROOT, "treegp_amd", "csrc", "bessel_k56.h"))
    so = os.path.join(d, "libvk.so")
    subprocess.check_call(["g++", "-O2", "-shared", "-fPIC", "-o", so, src])
    lib = ctypes.CDLL(so)
    rng = np.random.default_rng(0)
    u = np.concatenate([[0.0], 10 ** rng.uniform(-9, 2.05, 20000), [1 / (2 * np.pi), 32 / (2 * np.pi), 111.0, 111.08, 200.0]])
    out = np.empty_like(u)
    lib.vk(u.ctypes.data_as(ctypes.c_void_p), out.ctypes.data_as(ctypes.c_void_p), ctypes.c_long(len(u)))
    lim0 = special.gamma(5 / 6) / (2 * np.pi ** (5 / 6))
    ref = np.ones_like(u)
    nz = u != 0
    ref[nz] = u[nz] ** (5 / 6) * special.kv(5 / 6, 2 * np.pi * u[nz]) / lim0
    assert out[0] == 1.0
    np.testing.assert_allclose(out, ref, rtol=2e-13, atol=1e-13)      # kernel tables are pinned at atol 1e-12
    assert np.all(out[2 * np.pi * u > 697.874] == 0.0) and np.all(ref[2 * np.pi * u > 697.874] == 0.0)


def test_potrf128_register_budget():
    """potrf128 must fit on a SIMD beside one trailing-update wave (<= 264 VGPRs), or the look-ahead stalls until a
    compute unit is empty; checked on the generated ISA (cross-compiles, no GPU needed)."""
    import shutil
    import subprocess
    if shutil.which("hipcc") is None and not os.path.exists("/opt/rocm/bin/hipcc"):
        pytest.skip("no hipcc")
    script = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools", "check_potrf_regs.sh")
    r = subprocess.run(["bash", script], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout + r.stderr


def test_two_pcf_adjugate_inverse_matches_lapack():
    from treegp_amd.two_pcf import _inv2x2
    rng = np.random.default_rng(1)
    for _ in range(20):
        m = rng.standard_normal((2, 2)) * 10.0 ** rng.integers(-3, 4)
        np.testing.assert_allclose(_inv2x2(m), np.linalg.inv(m), rtol=1e-12)
    with pytest.raises(np.linalg.LinAlgError):
        _inv2x2(np.array([[1.0, 2.0], [2.0, 4.0]]))


def test_bench_self_launches_its_ranks_and_fails_loudly_without_a_gpu():
    """`python bench.py --gpus 2` with no launcher around it starts its two ranks itself (before anything initialises
    HIP in the parent); without a HIP device every rank stops with the library's own error and the parent returns a
    non-zero exit code -- no silent CPU path."""
    import subprocess
    import sys
    if _lib.load_library().tgp_device_count() > 0:
        pytest.skip("a GPU is present (the rehearsal with real kernels is in tests/test_gpu_dist.py)")
    env = dict(os.environ, TGP_DIST_BACKEND="gloo", TGP_ONE_DEVICE="1")
    env.pop("WORLD_SIZE", None); env.pop("RANK", None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0",
                        "--ntrain", "512", "--cpu-sample", "0"], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode != 0
    assert "no CPU path" in r.stderr or "no HIP device" in r.stderr, r.stderr[-2000:]
    assert not any(line.startswith("{") for line in r.stdout.splitlines())


class _FakeMinuit2(object):
    """The slice of iminuit 2.x's API that two_pcf.py:157-168 of the reference (and robust_2dfit._run_migrad here) uses:
    Minuit(fcn, start), .migrad(), .parameters, .params[name].value, .accurate.  The minimiser behind it is SciPy's."""
    __version__ = "2.30.1"

    class Minuit(object):
        def __init__(self, fcn, start):
            self.fcn, self.start = fcn, np.asarray(start, float)
            self.parameters = tuple("x%d" % i for i in range(len(self.start)))

        def migrad(self):
            from scipy import optimize
            from types import SimpleNamespace
            res = optimize.minimize(self.fcn, self.start, method="Nelder-Mead", options=dict(xatol=1e-7, fatol=1e-10, maxiter=4000))
            self.params = {n: SimpleNamespace(value=v) for n, v in zip(self.parameters, res.x)}
            self.accurate = bool(res.success)


class _FakeMinuit1(object):
    """iminuit 1.x: Minuit.from_array_func(fcn, start, print_level=0), .values (ordered mapping), .migrad_ok()."""
    __version__ = "1.5.4"

    class Minuit(object):
        @classmethod
        def from_array_func(cls, fcn, start, print_level=0):
            m = cls()
            m.fcn, m.start = fcn, np.asarray(start, float)
            return m

        def migrad(self):
            from scipy import optimize
            res = optimize.minimize(self.fcn, self.start, method="Nelder-Mead", options=dict(xatol=1e-7, fatol=1e-10, maxiter=4000))
            self.values = {"x%d" % i: v for i, v in enumerate(res.x)}
            self._ok = bool(res.success)

        def migrad_ok(self):
            return self._ok


@pytest.mark.parametrize("fake", [_FakeMinuit2, _FakeMinuit1])
def test_robust_2dfit_minuit_plumbing(fake, monkeypatch):
    """robust_2dfit through the MIGRAD branch (both iminuit API generations the reference supports,
    treegp/two_pcf.py:157-168) on a noiseless model correlation function: (size, g1, g2), amplitude and constant come
    back.  The pixel model is a closed-form Gaussian so that no GPU is needed."""
    import sys
    T = sys.modules["treegp_amd.two_pcf"]            # the package attribute of that name is the class, as in the reference

    class HostGauss(object):
        def __init__(self, invLam=None):
            self.invLam = invLam

        def __rmul__(self, c):
            self.amp = c
            return self

        def __call__(self, X, Y=None):
            d = X - Y
            q = np.einsum("ni,ij,nj->n", d, self.invLam, d)
            return (self.amp * np.exp(-0.5 * q))[:, None]

    monkeypatch.setattr(T, "iminuit", fake)
    monkeypatch.setattr(T, "get_kernel_class", lambda k: HostGauss)
    nb, mx = 15, 1.0
    c = (np.arange(nb) + 0.5) * 2 * mx / nb - mx
    xx, yy = np.meshgrid(c, c)
    x, y = xx.ravel(), yy.ravel()
    truth = (0.35, 0.15, -0.1)
    model = HostGauss(invLam=np.linalg.inv(T.get_correlation_length_matrix(*truth)))
    model.amp = 1.7 ** 2
    data = model(np.column_stack([x, y]), Y=np.zeros((1, 2)))[:, 0] + 0.02
    fit = T.robust_2dfit(None, data, x, y, np.eye(len(x)))
    fit.minimize_minuit(p0=[0.3, 0.0, 0.0])
    assert fit._fit_ok and hasattr(fit, "m")
    np.testing.assert_allclose(fit.result[1:4], truth, atol=2e-3)
    np.testing.assert_allclose(fit.result[0], 1.7, atol=2e-3)
    np.testing.assert_allclose(fit.result[4], 0.02, atol=1e-4)


def test_native_host_logic_under_sanitizers(tmp_path):
    """tools/sanitize_host.sh: all translation units rebuilt with AddressSanitizer + UBSan on the host side and the pure host
    logic (Morton keys, counting sorts, layout helpers, tile maps, the no-device error path) run on the CPU."""
    import shutil
    import subprocess
    if not (os.path.exists("/opt/rocm/bin/hipcc") or shutil.which("hipcc")):
        pytest.skip("hipcc not available")
    r = subprocess.run(["bash", os.path.join(ROOT, "tools", "sanitize_host.sh"), str(tmp_path)], capture_output=True, text=True,
                       timeout=1200)
    assert r.returncode == 0, (r.stdout[-1500:], r.stderr[-3000:])
    assert "host logic ok under sanitizers" in r.stdout
    assert "runtime error" not in r.stderr and "AddressSanitizer" not in r.stderr


def test_two_pcf_host_pieces_against_reference(golden, monkeypatch):
    """g12: the pure host pieces of treegp/two_pcf.py as the reference itself runs them -- correlation-length matrix, the
    seeded bootstrap resampling stream (two_pcf.py:264-281), the 1-D coordinate padding (:248-251), and robust_2dfit's chi2
    algebra (:115-148) fed with the reference's own model values (the model itself needs the GPU: tests/test_gpu_api.py)."""
    import sys
    T = sys.modules["treegp_amd.two_pcf"]
    g = golden("g12_two_pcf_host.npz")
    for q, ref in zip(g["clm_params"], g["clm"]):
        np.testing.assert_allclose(T.get_correlation_length_matrix(*q), ref, rtol=1e-15, atol=0)
    obj = tg.two_pcf(g["boot_X"], g["boot_y"], g["boot_yerr"], 0.0, 0.3, nbins=7, anisotropic=True)
    for r in range(3):
        u, v, yr, er = obj.resample_bootstrap()
        np.testing.assert_array_equal(u, g["boot_u"][r]); np.testing.assert_array_equal(v, g["boot_v"][r])
        np.testing.assert_array_equal(yr, g["boot_yr"][r]); np.testing.assert_array_equal(er, g["boot_er"][r])
    obj1 = tg.two_pcf(g["pad_X1"], g["boot_y"][:9], g["boot_yerr"][:9], 0.1, 1.0, nbins=5)
    np.testing.assert_array_equal(obj1.X, g["pad_X"])
    # chi2 / linear amplitudes with the reference's model values standing in for the kernel evaluation
    monkeypatch.setattr(T, "get_kernel_class", lambda k: None)
    for tag in ("arbf", "avk"):
        fit = T.robust_2dfit(None, g[tag + "_data"], g["fit_x"], g["fit_y"], g["fit_W"], mask=g["fit_mask"])
        for q, chi2, alpha, model in zip(g["fit_trial"], g[tag + "_chi2"], g[tag + "_alpha"], g[tag + "_model"]):
            fit._model_skl = lambda s, c, g1, g2, m=model: (None if (max(abs(g1), abs(g2)) > 1) else m)
            got = fit.chi2(q)
            if np.isfinite(chi2):
                np.testing.assert_allclose(got, chi2, rtol=1e-9)
                np.testing.assert_allclose(np.ravel(fit.alpha), alpha, rtol=1e-9)
            else:
                assert got == np.inf


def test_ml_gradient_mode_selection(monkeypatch):
    """log_likelihood.gradient: "auto" keeps the reference's finite differences where their evaluations run side by side and
    switches to the exact gradient above 16 384 points for kernels that have one; "analytic" insists; "fd" never."""
    import sys
    mod = sys.modules["treegp_amd.log_likelihood"]
    small, big = np.zeros((100, 2)), np.zeros((mod._PARALLEL_MAX_N + 1, 2))
    gauss = tg.eval_kernel("1.0**2 * AnisotropicRBF(scale_length=[0.3, 0.2])")
    vk = tg.eval_kernel("1.0**2 * VonKarman(length_scale=0.3)")
    monkeypatch.delenv("TGP_ML_GRADIENT", raising=False)
    assert mod.log_likelihood(small, None, None).gradient == "auto"
    assert not mod.log_likelihood(small, None, None)._use_exact_gradient(gauss)
    assert mod.log_likelihood(big, None, None)._use_exact_gradient(gauss)
    assert not mod.log_likelihood(big, None, None)._use_exact_gradient(vk)
    # beyond the device gradient's own limit (row tiles of a matrix of at most 65 535 rows, csrc/cov.hip) the fit keeps
    # finite differences instead of failing into -inf at the start point: n = 65 281 pads to 65 536
    assert mod.log_likelihood(np.zeros((65280, 2)), None, None)._use_exact_gradient(gauss)
    assert not mod.log_likelihood(np.zeros((65281, 2)), None, None)._use_exact_gradient(gauss)
    assert not mod.log_likelihood(np.zeros((65536, 2)), None, None)._use_exact_gradient(gauss)
    monkeypatch.setenv("TGP_ML_GRADIENT", "analytic")
    assert mod.log_likelihood(small, None, None)._use_exact_gradient(gauss)
    with pytest.raises(NotImplementedError, match="no analytic derivative"):
        mod.log_likelihood(small, None, None)._use_exact_gradient(vk)
    monkeypatch.setenv("TGP_ML_GRADIENT", "fd")
    assert not mod.log_likelihood(big, None, None)._use_exact_gradient(gauss)


def test_multi_gpu_route_selection(monkeypatch):
    """Which solves take the multi-GPU route (treegp_amd.dist.enable / scope / TGP_DIST, GPInterpolation(backend=...)): host
    logic only, with a stand-in for the engine (the real one needs a GPU)."""
    from treegp_amd import dist, ops

    class FakeEngine(object):
        def __init__(self, comm=None, device=None, min_n=None, profile=False):
            self.min_n = dist.DEFAULT_MIN_N if min_n is None else int(min_n)
            self.comm = type("Comm", (), {"size": 2, "rank": 0})()

    monkeypatch.setattr(dist, "DistEngine", FakeEngine)
    monkeypatch.setattr(ops, "set_pair_comm", lambda comm: None)
    monkeypatch.delenv("TGP_DIST", raising=False)
    assert dist.engine_for(10 ** 6) is None                              # nothing enabled: one GPU
    eng = dist.enable(min_n=1000)
    try:
        assert dist.engine_for(999) is None and dist.engine_for(1000) is eng      # never implicit below the threshold
        assert ops._dist_engine(5000, None) is eng
        assert ops._dist_engine(5000, object()) is None                  # a caller that names its context stays on its GPU
        with dist.scope("single"):
            assert dist.engine_for(10 ** 6) is None
        with dist.scope("dist"):
            assert dist.engine_for(5) is eng                             # the explicit kwarg: any size
            with dist.scope(None):                                       # a nested scope that says nothing changes nothing
                assert dist.engine_for(5) is eng
        assert dist.engine_for(5) is None
    finally:
        dist.disable()
    assert dist.engine_for(10 ** 6) is None
    # backend="dist" with no engine and no process group: an ordinary single-process script gets the worker pool
    # (treegp_amd/dist_pool.py; created here, its processes only start with the first solve) ...
    for v in ("RANK", "LOCAL_RANK", "TORCHELASTIC_RUN_ID", "TGP_DIST_POOL"):
        monkeypatch.delenv(v, raising=False)
    with dist.scope("dist"):
        pool_eng = dist.engine_for(5)
    assert type(pool_eng).__name__ == "PoolEngine" and pool_eng.pool is None
    assert dist.engine_for(5) is None and dist.engine_for(10 ** 9) is pool_eng     # from then on it is this process's engine
    dist.disable()
    # ... unless pools are switched off, or this process is itself a rank of a torchrun job: then it is loud
    for var, val in (("TGP_DIST_POOL", "0"), ("RANK", "0")):
        monkeypatch.setenv(var, val)
        with pytest.raises(RuntimeError, match="backend"):
            with dist.scope("dist"):
                dist.engine_for(5)
        monkeypatch.delenv(var)
    assert dist.engine_for(5) is None                                    # (the scope was left properly)
    monkeypatch.setenv("TGP_DIST", "1")
    monkeypatch.setattr(dist, "_warned_no_group", False)
    assert dist.engine_for(100) is None                                  # below the threshold: nothing is demanded, nothing said
    with pytest.warns(RuntimeWarning, match="TGP_DIST=1"):
        assert dist.engine_for(10 ** 6) is None                          # asked for by environment, torch.distributed not up:
    import warnings                                                      # one warning, then the single-GPU path, quietly
    with warnings.catch_warnings():
        warnings.simplefilter("error")
        assert dist.engine_for(10 ** 6) is None
    monkeypatch.delenv("TGP_DIST")
    with pytest.raises(ValueError):
        dist.scope("gpu")
    with pytest.raises(ValueError, match="backend"):
        tg.GPInterpolation(kernel="RBF(1)", optimizer="none", backend="cluster")
    gp = tg.GPInterpolation(kernel="RBF(1)", optimizer="none", backend="single")
    assert gp.backend == "single" and tg.GPInterpolation(kernel="RBF(1)", optimizer="none").backend is None


def test_reflected_block_cyclic_owner_map_partitions_and_balances():
    """The multi-GPU owner map (treegp_amd.dist.owner and friends = csrc/tgp_internal.h dist_*): every block has one owner,
    local indices are b // G, the gathered-panel slots are distinct and within cmax, the C side agrees on the sizes, and the
    lower triangle's work is even over the ranks (what the plain deal b % G was not: 4.1 % above the mean on the last rank)."""
    from treegp_amd import _lib, dist
    lib = _lib.load_library()
    for G in (1, 2, 3, 4, 5, 8):
        for nB in (1, 2, G, G + 1, 2 * G, 2 * G + 1, 3 * G - 1, 37, 256):
            own = [dist.owner(b, G) for b in range(nB)]
            for r in range(G):
                mine = [b for b in range(nB) if own[b] == r]
                assert mine == [dist.block_of(q, r, G) for q in range(len(mine))]
                assert all(b // G == q for q, b in enumerate(mine))
                for p in range(nB + 1):
                    ge = [b for b in mine if b >= p]
                    assert dist.panel_blocks(p, nB, r, G) == len(ge)
                    if ge:
                        assert dist.first_ge(p, r, G) == ge[0] and dist.first_round(p, r, G) == ge[0] // G
                    assert lib.tgp_dist_panel_rows(p, nB * 256, G, r) == 256 * len(ge)
                assert lib.tgp_dist_local_elems(nB * 256, G, r) == sum(256 * 256 * len([b for b in mine if b >= p]) for p in range(nB))
            for first in range(0, nB, max(1, nB // 7)):
                cmax = dist.panel_cmax(first, nB, G)
                slots = [dist.gathered_index(b, first, G) for b in range(first, nB)]
                assert len(set(slots)) == len(slots) and all(0 <= i < cmax for _, i in slots)
                assert cmax == max(sum(1 for r_, _ in slots if r_ == r) for r in range(G))
    for G, nB, tol in ((8, 256, 1.004), (8, 512, 1.002), (4, 256, 1.002), (2, 256, 1.001)):
        work = [0.0] * G
        for b in range(nB):
            work[dist.owner(b, G)] += b + 0.5          # block row b: b full blocks and half a diagonal one
        assert max(work) / (sum(work) / G) < tol, (G, nB, max(work) / (sum(work) / G))
        plain = [sum(b + 0.5 for b in range(r, nB, G)) for r in range(G)]
        assert max(plain) / (sum(plain) / G) > max(work) / (sum(work) / G)


def test_device_errors_inside_one_likelihood_evaluation(monkeypatch):
    """treegp/log_likelihood.py:38-39 turns any failure of one evaluation into -inf.  Here: run-time device errors (rc -2)
    likewise, with a RuntimeWarning; argument errors (rc -1) raise; TGP_ML_STRICT=1 raises both."""
    import warnings
    from treegp_amd import _lib, ops
    import sys
    import treegp_amd  # noqa: F401
    from treegp_amd.kernels import eval_kernel
    mod = sys.modules["treegp_amd.log_likelihood"]

    def failing(rc):
        def f(*a, **k):
            err = _lib.TgpError("tgp_gp_solve failed (%d): hipMalloc: out of memory" % rc)
            err.rc = rc
            raise err
        return f

    monkeypatch.setattr(ops, "_dist_engine", lambda n, ctx: None)
    ll = mod.log_likelihood(np.zeros((10, 2)), np.zeros(10), np.ones(10))
    k = eval_kernel("1.0**2 * AnisotropicRBF(invLam=array([[30., 4.], [4., 20.]]))")
    monkeypatch.delenv("TGP_ML_STRICT", raising=False)
    monkeypatch.setattr(ops, "gp_solve", failing(-2))
    with pytest.warns(RuntimeWarning, match="counts as -inf"):
        assert ll.log_likelihood(k) == -np.inf
    with pytest.warns(RuntimeWarning):
        val, grad = ll.log_likelihood_gradient(k)
    assert val == -np.inf and np.all(grad == 0)
    monkeypatch.setattr(ops, "gp_solve", failing(-1))
    with pytest.raises(_lib.TgpError):
        ll.log_likelihood(k)
    monkeypatch.setattr(ops, "gp_solve", failing(-2))
    monkeypatch.setenv("TGP_ML_STRICT", "1")
    with pytest.raises(_lib.TgpError):
        ll.log_likelihood(k)

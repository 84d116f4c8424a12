"""The multi-GPU route BEHIND the drop-in API (SURVEY 8e through treegp/gp_interp.py:143-194, 196-227, 229-243):
``GPInterpolation(backend="dist")`` / ``treegp_amd.dist.enable()`` on configs[4]'s recipe -- mean-function table,
non-uniform y_err, white noise, normalize -- with 2, 4 and 8 virtual ranks on one GPU (threads, own tgp_ctx each, the
real kernels), and the same script as real processes over gloo.  Compared with the single-GPU API result and the oracle."""
import os
import socket
import subprocess
import sys
import threading

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)


def _recipe(n, m, tmp_path, seed=5):
    from treegp_amd.fits_io import write_bintable_row
    from treegp_amd.synthetic import star_field_with_mean, headline_kernel_string
    X, y, y_err, Xs, X0, y0 = star_field_with_mean(n, m, seed=seed)
    fits = os.path.join(str(tmp_path), "mean.fits")
    write_bintable_row(fits, {"COORDS0": X0, "PARAMS0": y0})
    kw = dict(kernel=headline_kernel_string(), optimizer="none", normalize=True, white_noise=0.004, average_fits=fits)
    return X, y, y_err, Xs, X0, y0, kw


def _oracle_predict(X, y, y_err, Xs, X0, y0, white_noise):
    """configs[4]'s recipe restated with the oracle: KNN-4 mean function, mean of the residual, noise in quadrature"""
    from oracle import gp_oracle as O
    from treegp_amd.synthetic import headline_invlam
    iL = headline_invlam()
    kw = dict(amp=1.0, a=iL[0, 0], b=iL[0, 1], c=iL[1, 1])
    avg = O.knn_mean(X0, y0, X, 4)
    mean = np.mean(y - avg)
    sigma = np.sqrt(y_err ** 2 + white_noise ** 2)
    alpha, logdet = O.gp_solve(O.kernel_matrix("gauss", X, **kw), y - mean - avg, sigma)
    return O.gp_predict(O.kernel_matrix("gauss", Xs, X, **kw), alpha) + mean + O.knn_mean(X0, y0, Xs, 4)


def _virtual_ranks(G, fn):
    """run fn(rank) on G threads, each with its own context and a thread-local engine over an in-process communicator"""
    import torch
    from treegp_amd import _lib, dist
    from _dist_helpers import ThreadComm
    shared = ThreadComm.Shared(G)
    results, errors = [None] * G, []
    dev = torch.device("cuda", 0)

    def run(rank):
        try:
            _lib.set_thread_ctx(_lib.new_ctx(0))
            dist.enable(comm=ThreadComm(shared, rank), device=dev, min_n=10 ** 9, thread_local=True)
            try:
                results[rank] = fn(rank)
            finally:
                dist.disable()
                _lib.set_thread_ctx(None)
        except BaseException as e:            # noqa: BLE001 - surface any failure of a virtual rank
            import sys
            import traceback
            sys.stderr.write("virtual rank %d:\n%s\n" % (rank, traceback.format_exc()))
            errors.append(e)
            try:
                shared.barrier.abort()
            except Exception:
                pass

    threads = [threading.Thread(target=run, args=(r,)) for r in range(G)]
    for t in threads:
        t.start()
    for t in threads:
        t.join(timeout=600)
    torch.cuda.empty_cache()                  # the ranks' shares (tens of GB at the BASELINE sizes) go back to the device
    assert not errors, errors
    return results


@pytest.mark.parametrize("G,n,group", [(2, 12000, 2), (4, 12000, 4), (8, 12000, 2), (8, 5000, 4)])
def test_config5_recipe_through_the_api_virtual_ranks(G, n, group, tmp_path, monkeypatch):
    import treegp_amd
    monkeypatch.setenv("TGP_DIST_GROUP", str(group))
    m = 3000
    X, y, y_err, Xs, X0, y0, kw = _recipe(n, m, tmp_path)
    one = treegp_amd.GPInterpolation(**kw)
    one.initialize(X, y, y_err)
    ref = one.predict(Xs)
    ll_ref = one.return_log_likelihood()
    _, cov_ref = one.predict(Xs[:150], return_cov=True)
    ora = _oracle_predict(X, y, y_err, Xs, X0, y0, kw["white_noise"]) if n <= 6000 else None

    def rank_fn(rank):
        gp = treegp_amd.GPInterpolation(backend="dist", **kw)
        gp.initialize(X, y, y_err)
        yp = gp.predict(Xs)
        assert gp._alpha is not None
        yp_again = gp.predict(Xs)                              # the cached alpha: no second factorisation
        _, cov = gp.predict(Xs[:150], return_cov=True)
        return yp, yp_again, cov, gp.return_log_likelihood()

    scale = np.abs(ref).max()
    for yp, yp_again, cov, ll in _virtual_ranks(G, rank_fn):
        np.testing.assert_allclose(yp, ref, rtol=0, atol=1e-10 * scale)
        np.testing.assert_array_equal(yp, yp_again)
        np.testing.assert_allclose(cov, cov_ref, rtol=0, atol=1e-9 * np.abs(cov_ref).max())
        np.testing.assert_allclose(ll, ll_ref, rtol=1e-11)
        if ora is not None:
            np.testing.assert_allclose(yp, ora, rtol=0, atol=1e-10 * np.abs(ora).max())


def test_default_schedule_at_a_size_that_takes_groups_of_four(tmp_path):
    """Nothing forced: at N = 30 000 the driver picks groups of four panels by itself (Np >= 28 672), the schedule the
    BASELINE sizes run; four virtual ranks through the API against the single-GPU API and the device residual."""
    import treegp_amd
    from treegp_amd import ops
    from treegp_amd.kernels import kernel_to_spec
    n, m, G = 30000, 4000, 4
    X, y, y_err, Xs, X0, y0, kw = _recipe(n, m, tmp_path, seed=13)
    one = treegp_amd.GPInterpolation(**kw)
    one.initialize(X, y, y_err)
    ref = one.predict(Xs)

    def rank_fn(rank):
        gp = treegp_amd.GPInterpolation(backend="dist", **kw)
        gp.initialize(X, y, y_err)
        return gp.predict(Xs), gp._alpha.copy()

    spec = kernel_to_spec(one.kernel)
    resid_y = one._y - one._mean - one._spatial_average
    for yp, alpha in _virtual_ranks(G, rank_fn):
        np.testing.assert_allclose(yp, ref, rtol=0, atol=1e-10 * np.abs(ref).max())
        Ka = ops.gp_predict(spec, X, alpha, X)                 # K alpha without K
        rel = np.linalg.norm(Ka + one._y_err ** 2 * alpha - resid_y) / np.linalg.norm(resid_y)
        assert rel < 1e-10, rel


def test_config4_problem_with_eight_virtual_ranks_through_the_api(tmp_path):
    """configs[3] of BASELINE.json in its eight-rank form, through the drop-in API: N = 65 536, groups of four panels, replicated
    factor, 8 virtual ranks sharing this one GPU (the collectives' payload goes through an in-process communicator, not xGMI).
    Predictions equal the single-GPU API's at 1e-10 of the field's scale; the solve's residual is at rounding level."""
    import treegp_amd
    from treegp_amd import ops
    from treegp_amd.kernels import kernel_to_spec
    from treegp_amd.synthetic import star_field, headline_kernel_string
    n, m, G = 65536, 4096, 8
    X, y, y_err, Xs = star_field(n, m)
    kw = dict(kernel=headline_kernel_string(), optimizer="none", normalize=True)
    one = treegp_amd.GPInterpolation(**kw)
    one.initialize(X, y, y_err)
    ref = one.predict(Xs)

    def rank_fn(rank):
        gp = treegp_amd.GPInterpolation(backend="dist", **kw)
        gp.initialize(X, y, y_err)
        yp = gp.predict(Xs)
        return (yp, gp._alpha.copy()) if rank == 0 else (yp, None)

    res = _virtual_ranks(G, rank_fn)
    for yp, _ in res:
        np.testing.assert_allclose(yp, ref, rtol=0, atol=1e-10 * np.abs(ref).max())
    alpha = res[0][1]
    spec = kernel_to_spec(one.kernel)
    r = one._y - one._mean
    Ka = ops.gp_predict(spec, X, alpha, X)
    rel = np.linalg.norm(Ka + y_err ** 2 * alpha - r) / np.linalg.norm(r)
    assert rel < 1e-10, rel


def test_config5_full_size_with_eight_virtual_ranks_through_the_api(tmp_path, monkeypatch):
    """configs[4] of BASELINE.json at full size in its eight-rank form, through the drop-in API: N = 131 072 with the mean-function
    table (KNN-4) and non-uniform y_err; 8 virtual ranks on this one GPU, so the factor cannot be replicated eight times and the
    sweeps are the distributed ones (TGP_DIST_REPLICATE=0 -- on a real node every rank has room for its copy).  Against the
    single-GPU API on the same data: predictions at 1e-10 of the field's scale, and the device residual of the solve."""
    import treegp_amd
    from treegp_amd import ops
    from treegp_amd.kernels import kernel_to_spec
    monkeypatch.setenv("TGP_DIST_REPLICATE", "0")
    n, m, G = 131072, 2048, 8
    X, y, y_err, Xs, X0, y0, kw = _recipe(n, m, tmp_path, seed=20240613)
    one = treegp_amd.GPInterpolation(**kw)
    one.initialize(X, y, y_err)
    ref = one.predict(Xs)
    assert np.abs(one._spatial_average).max() > 0.02            # the mean function is really there

    def rank_fn(rank):
        gp = treegp_amd.GPInterpolation(backend="dist", **kw)
        gp.initialize(X, y, y_err)
        yp = gp.predict(Xs)
        return (yp, gp._alpha.copy()) if rank == 0 else (yp, None)

    res = _virtual_ranks(G, rank_fn)
    for yp, _ in res:
        np.testing.assert_allclose(yp, ref, rtol=0, atol=1e-10 * np.abs(ref).max())
    alpha = res[0][1]
    r = one._y - one._mean - one._spatial_average
    Ka = ops.gp_predict(kernel_to_spec(one.kernel), X, alpha, X)
    rel = np.linalg.norm(Ka + one._y_err ** 2 * alpha - r) / np.linalg.norm(r)
    assert rel < 1e-10, rel


def test_without_the_replicated_factor(tmp_path, monkeypatch):
    """TGP_DIST_REPLICATE=0 (or a factor that does not fit beside a rank's share): the solves run as distributed sweeps and
    give the same predictions; calls that need a kept factor say so on every rank instead of computing something else."""
    import treegp_amd
    monkeypatch.setenv("TGP_DIST_REPLICATE", "0")
    n, m = 2600, 500
    X, y, y_err, Xs, X0, y0, kw = _recipe(n, m, tmp_path, seed=21)
    one = treegp_amd.GPInterpolation(**kw)
    one.initialize(X, y, y_err)
    ref, ll_ref = one.predict(Xs), one.return_log_likelihood()

    def rank_fn(rank):
        gp = treegp_amd.GPInterpolation(backend="dist", **kw)
        gp.initialize(X, y, y_err)
        yp = gp.predict(Xs)
        with pytest.raises(NotImplementedError, match="replicated factor"):
            gp.predict(Xs[:20], return_cov=True)
        with pytest.raises(NotImplementedError, match="replicated factor"):
            gp.predict_fields(np.stack([y, y]), Xs)
        return yp, gp.return_log_likelihood()

    for yp, ll in _virtual_ranks(3, rank_fn):
        np.testing.assert_allclose(yp, ref, rtol=0, atol=1e-10 * np.abs(ref).max())
        np.testing.assert_allclose(ll, ll_ref, rtol=1e-11)


def test_threshold_route_fields_and_errors_virtual_ranks(tmp_path):
    """An enabled engine takes objects that name no backend once they reach its size threshold and leaves smaller ones
    alone; ``predict_fields`` solves every field against the replicated factor; a matrix that is not positive definite
    raises LinAlgError on every rank; ``backend="single"`` never leaves the rank's own GPU."""
    import treegp_amd
    from treegp_amd import dist
    n, m, G = 3000, 800, 3
    X, y, y_err, Xs, X0, y0, kw = _recipe(n, m, tmp_path, seed=9)
    rng = np.random.default_rng(2)
    Y = np.stack([y, y[::-1].copy(), y + 0.1 * rng.standard_normal(n)])
    one = treegp_amd.GPInterpolation(**kw)
    one.initialize(X, y, y_err)
    ref, ref_fields = one.predict(Xs), one.predict_fields(Y, Xs)

    def rank_fn(rank):
        eng = dist.engine_for(10 ** 9)                        # the thread's engine (min_n = 1e9 so far)
        assert dist.engine_for(n) is None
        eng.min_n = 2000
        assert dist.engine_for(n) is eng and dist.engine_for(1999) is None
        gp = treegp_amd.GPInterpolation(**kw)                 # no backend named: the threshold decides
        gp.initialize(X, y, y_err)
        solves = []
        orig = eng.gp_solve
        eng.gp_solve = lambda *a, **k: (solves.append(1), orig(*a, **k))[1]
        yp = gp.predict(Xs)
        fields = gp.predict_fields(Y, Xs)
        assert len(solves) == 2
        single = treegp_amd.GPInterpolation(backend="single", **kw)
        single.initialize(X, y, y_err)
        ys = single.predict(Xs)
        assert len(solves) == 2                               # that one stayed on the single-GPU path
        bad = treegp_amd.GPInterpolation(kernel=kw["kernel"], optimizer="none", backend="dist")
        Xd = X.copy()
        Xd[1] = Xd[0]
        bad.initialize(Xd, y, np.zeros(n))
        with pytest.raises(np.linalg.LinAlgError):
            bad.predict(Xs[:10])
        # the same with the coincident pair among the LAST points: the failing pivot lies in the replicated finish, which every
        # rank factors redundantly -- still one LinAlgError on every rank
        bad2 = treegp_amd.GPInterpolation(kernel=kw["kernel"], optimizer="none", backend="dist")
        Xe = X.copy()
        Xe[n - 1] = Xe[n - 2]
        bad2.initialize(Xe, y, np.zeros(n))
        with pytest.raises(np.linalg.LinAlgError):
            bad2.predict(Xs[:10])
        return yp, fields, ys

    scale = np.abs(ref).max()
    for yp, fields, ys in _virtual_ranks(G, rank_fn):
        np.testing.assert_allclose(yp, ref, rtol=0, atol=1e-10 * scale)
        np.testing.assert_allclose(ys, ref, rtol=0, atol=1e-12 * scale)
        np.testing.assert_allclose(fields, ref_fields, rtol=0, atol=1e-10 * np.abs(ref_fields).max())


def test_ml_fit_and_two_pcf_through_the_api_virtual_ranks():
    """``solve()`` with the likelihood optimiser and with the 2-pcf optimiser on the multi-GPU route: every likelihood
    evaluation is one distributed factorisation, the pair binning is sharded over the same ranks (ops.set_pair_comm by
    the same switch).  Data: Gaussian random fields drawn from the kernels, as the reference's tests/test_hyp_search.py
    draws them; the fits end where the single-GPU fits end and within the reference's tolerance of the truth."""
    import treegp_amd
    from treegp_amd.synthetic import correlation_length_matrix
    invL = np.linalg.inv(correlation_length_matrix(0.5, 0.2, 0.2))
    kern = "%f**2*%s" % (2.0, "AnisotropicRBF") + "(invLam={0!r})".format(invL)
    truth = treegp_amd.eval_kernel(kern)
    rng = np.random.default_rng(42)
    n = 600
    X = rng.uniform(-10, 10, (n, 2))
    y = rng.multivariate_normal(np.zeros(n), truth(X)) + 0.01 * rng.standard_normal(n)
    y_err = np.full(n, 0.01)
    iso = "1.000000**2 * RBF(0.500000)"
    n1 = 2000
    X1 = rng.uniform(-10, 10, (n1, 1))
    y1 = rng.multivariate_normal(np.zeros(n1), treegp_amd.eval_kernel(iso)(X1)) + 0.01 * rng.standard_normal(n1)
    e1 = np.full(n1, 0.01)
    n2 = 2000
    X2 = rng.uniform(-10, 10, (n2, 2))
    y2 = rng.multivariate_normal(np.zeros(n2), truth(X2)) + 0.01 * rng.standard_normal(n2)
    e2 = np.full(n2, 0.01)
    cases = {"log-likelihood": (kern, X, y, y_err, dict()),
             "two-pcf": (iso, X1, y1, e1, dict(nbins=15, min_sep=0.1, max_sep=1.75)),
             # the TwoD correlation function with its bootstrap covariance (resamples dealt to the ranks) and the robust 2-D fit
             "anisotropic": (kern, X2, y2, e2, dict(nbins=21, min_sep=0.0, max_sep=1.0, p0=[0.3, 0.0, 0.0]))}

    def fit(opt, backend):
        k, Xc, yc, ec, extra = cases[opt]
        gp = treegp_amd.GPInterpolation(kernel=k, optimizer=opt, normalize=True, backend=backend, **extra)
        gp.initialize(Xc, yc, ec)
        gp.solve()
        return gp.kernel.theta.copy(), gp.return_log_likelihood(), gp.predict(Xc[:50])

    single = {opt: fit(opt, None) for opt in cases}

    def rank_fn(rank):
        return {opt: fit(opt, "dist") for opt in cases}

    for out in _virtual_ranks(2, rank_fn):
        # the 2-pcf fit sees the same binned sums to 1e-12 and ends at the same kernel
        np.testing.assert_allclose(out["two-pcf"][0], single["two-pcf"][0], rtol=1e-6)
        np.testing.assert_allclose(out["two-pcf"][0], treegp_amd.eval_kernel(iso).theta, atol=7e-1)       # test_hyp_search.py:43
        # the likelihood fit: finite-difference gradients of solves that differ in their last bits take another path to the
        # same maximum -- same likelihood there, theta within the reference's own tolerance of the truth (:141)
        np.testing.assert_allclose(out["log-likelihood"][1], single["log-likelihood"][1], rtol=1e-6)
        np.testing.assert_allclose(out["log-likelihood"][0], truth.theta, atol=5e-1)
        np.testing.assert_allclose(out["log-likelihood"][0], single["log-likelihood"][0], atol=1e-1)
        # the anisotropic fit: same pixel sums and bootstrap rows to 1e-12, same chi2 minimiser: the same kernel
        np.testing.assert_allclose(out["anisotropic"][0], single["anisotropic"][0], rtol=1e-5, atol=1e-6)
        # (against the truth this draw is a 0.76 fluctuation in one component; tests/test_gpu_api.py holds the reference's own draw)


@pytest.mark.parametrize("G", [2, 4])
def test_api_script_as_real_processes_gloo(G):
    """tests/_api_dist_script.py under torch.distributed.run: G processes sharing this box's one GPU over gloo -- the
    launch a multi-GPU node makes with nccl = RCCL and one rank per device."""
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ, TGP_DIST_BACKEND="gloo", TGP_ONE_DEVICE="1", OMP_NUM_THREADS="4", HSA_ENABLE_IPC_MODE_LEGACY="0")
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT", "TGP_DIST"):
        env.pop(k, None)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(G), "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(HERE, "_api_dist_script.py"), "3000", "1500"]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0 and "api dist ok: world %d" % G in r.stdout, r.stdout[-2000:] + r.stderr[-4000:]


def test_api_script_on_rccl_world_of_one():
    """the same script on the real nccl (RCCL) backend with one rank: every torch.distributed call of the API route"""
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK="0", WORLD_SIZE="1", LOCAL_RANK="0")
    env.pop("TGP_DIST_BACKEND", None)
    r = subprocess.run([sys.executable, os.path.join(HERE, "_api_dist_script.py"), "2500", "700"], env=env,
                       capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "api dist ok: world 1" in r.stdout, r.stdout[-2000:] + r.stderr[-4000:]


@pytest.mark.parametrize("G,how", [(2, "explicit"), (4, "explicit"), (2, "env")])
def test_unmodified_single_process_script_reaches_the_multi_gpu_route(G, how):
    """VERDICT r4 item 4 / SURVEY 8(b): an ordinary ``python script.py`` -- no torchrun, no torch.distributed in the script --
    gets the multi-GPU route from ``GPInterpolation(backend="dist")`` (or from TGP_DIST=1 alone): the parent starts G worker
    processes once (treegp_amd/dist_pool.py; here all on this box's one GPU over gloo), ships the arrays through shared
    memory and returns rank 0's results -- equal to the single-GPU API at 1e-10, LinAlgError included, workers reused."""
    env = dict(os.environ, TGP_DIST_BACKEND="gloo", TGP_ONE_DEVICE="1", TGP_DIST_POOL_WORLD=str(G), TGP_DIST_WATCHDOG_S="120")
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "TORCHELASTIC_RUN_ID", "TGP_DIST"):
        env.pop(k, None)
    if how == "env":
        env["TGP_DIST"] = "1"
        env["TGP_DIST_MIN_N"] = "1000"
    r = subprocess.run([sys.executable, os.path.join(HERE, "_api_pool_script.py"), "3000", "1500", how], env=env,
                       capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    assert "POOL_OK world=%d" % G in r.stdout, r.stdout[-2000:]

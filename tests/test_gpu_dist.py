"""GPU tests of the multi-GPU kernels and driver on ONE GPU: world of one, and G virtual ranks
(threads, each with its own tgp_ctx) exchanging data through an in-process communicator."""
import os
import sys
import threading

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))


def _problem(n, m, seed=0):
    from treegp_amd.synthetic import star_field, headline_invlam
    from treegp_amd import _lib, ops
    X, y, y_err, Xs = star_field(n, m, seed=seed)
    iL = headline_invlam()
    spec = ops.KernelSpec(_lib.TGP_ARBF, amp=1.0, a=iL[0, 0], b=iL[0, 1], c=iL[1, 1])
    return spec, X, y - y.mean(), y_err, Xs


def test_world_of_one_matches_single_gpu_path():
    import torch
    from treegp_amd import _lib, ops
    from treegp_amd.dist import DistributedGP, SelfComm
    spec, X, y, y_err, Xs = _problem(1900, 1000)
    alpha_ref, logdet_ref, _, _ = ops.gp_solve(spec, X, y, y_err)
    yp_ref = ops.gp_predict(spec, X, alpha_ref, Xs)
    ctx = _lib.new_ctx(0)
    gp = DistributedGP(ctx, spec, X, y, y_err, Xs, comm=SelfComm(), device=torch.device("cuda", 0))
    alpha, dys = gp.step()
    torch.cuda.synchronize()
    a = alpha.cpu().numpy()[:len(y)]
    np.testing.assert_allclose(a, alpha_ref, rtol=0, atol=1e-11 * np.abs(alpha_ref).max())
    np.testing.assert_allclose(float(gp.logdet[0]), logdet_ref, rtol=1e-13)
    np.testing.assert_allclose(dys.cpu().numpy(), yp_ref, rtol=0, atol=1e-11 * np.abs(yp_ref).max())
    _lib.load_library().tgp_reset_stream(ctx)


@pytest.mark.parametrize("G,n,group", [(2, 1500, 2), (3, 2300, 2), (4, 1100, 2), (8, 4500, 2), (5, 7000, 2), (8, 12000, 2),
                                       (6, 257, 2), (4, 513, 2), (3, 3000, 4), (8, 6000, 4), (2, 2700, 3), (4, 1900, 1)])
def test_virtual_ranks(G, n, group, monkeypatch):
    monkeypatch.setenv("TGP_DIST_GROUP", str(group))
    _run_virtual_ranks(G, n)


@pytest.mark.parametrize("G,n,group,units", [(2, 9000, 4, "1"), (4, 6000, 2, "2"), (1, 5000, 4, "3"), (3, 7000, 3, "-1")])
def test_virtual_ranks_queued_bulk(G, n, group, units, monkeypatch):
    """The bulk update as the persistent grid that keeps compute units clear for the panel chain (syrk_distn_queue_kernel),
    the chain's diagonal blocks on units of their own: forced with 1 .. 3 units per shader engine, and the per-step
    decision (-1) with a chain estimate that makes every step chain-bound."""
    monkeypatch.setenv("TGP_DIST_GROUP", str(group))
    monkeypatch.setenv("TGP_DIST_QUEUE", units)
    monkeypatch.setenv("TGP_DIST_CHAIN_US", "100000")
    _run_virtual_ranks(G, n)


@pytest.mark.parametrize("G,n,group,finish,replicate", [(4, 6000, 2, 7, None), (8, 9000, 4, 12, None), (3, 5000, 3, 0, None),
                                                        (2, 4000, 4, 64, None), (1, 5000, 4, 8, None), (4, 7000, 4, 9, "0"),
                                                        (1, 4000, 2, 6, "0"), (1, 4000, 2, 6, "1")])
def test_virtual_ranks_replicated_finish(G, n, group, finish, replicate, monkeypatch):
    """TGP_DIST_FINISH: the last `finish` blocks leave the distributed chain -- one all-gather of the trailing matrix, every rank
    factors it with the single-GPU schedule (tgp_dd_tail_assemble + tgp_d_potrf + tgp_dd_tail_scatter): tails that are not
    multiples of the group size, none at all, the whole matrix, a world of one, and the route without the replicated factor
    (the tail is factored in a buffer and only scattered back into the shares)."""
    monkeypatch.setenv("TGP_DIST_GROUP", str(group))
    monkeypatch.setenv("TGP_DIST_FINISH", str(finish))
    if replicate is not None:
        monkeypatch.setenv("TGP_DIST_REPLICATE", replicate)
    _run_virtual_ranks(G, n)


def test_world_of_one_forced_to_the_fused_launch(monkeypatch):
    """a world of one takes the two-launch form by default (whole-chip launches: LAB_NOTES A.0); TGP_DIST_FUSED=1 forces the fused one"""
    monkeypatch.setenv("TGP_DIST_GROUP", "4")
    monkeypatch.setenv("TGP_DIST_FUSED", "1")
    monkeypatch.setenv("TGP_DIST_FINISH", "0")
    _run_virtual_ranks(1, 6000)


@pytest.mark.parametrize("G,n,group,units", [(2, 5000, 4, "0"), (8, 6000, 4, "0"), (3, 4000, 2, "2")])
def test_virtual_ranks_split_update_with_events(G, n, group, units, monkeypatch):
    """The two-launch form of a group's update (head columns, event, rest) that the driver falls back to where stream
    wait-value hand-offs are not available (csrc/handoff.hip's trial, TGP_SYNC_EVENTS=1); the default everywhere else in this
    file is the fused launch that releases the panel chain from inside the kernel."""
    monkeypatch.setenv("TGP_DIST_GROUP", str(group))
    monkeypatch.setenv("TGP_DIST_QUEUE", units)
    monkeypatch.setenv("TGP_DIST_FUSED", "0")
    _run_virtual_ranks(G, n)


def test_hand_offs_by_stream_wait_value_pass_their_trial_here():
    """On a plain GPU box the first-use trial of csrc/handoff.hip (consumer parked first, producer second, host timeout)
    must choose flags + hipStreamWaitValue32 -- a silent fall-back to events would cost 5 - 9 % at N <= 8192 and the fused
    multi-GPU launch -- and TGP_SYNC_EVENTS=1 (the profiler setting) must choose events."""
    import subprocess
    from treegp_amd import _lib
    lib = _lib.load_library()
    assert lib.tgp_handoff_mode(_lib.new_ctx(0)) == 1
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = "import sys; sys.path.insert(0, %r); from treegp_amd import _lib; print('mode', _lib.load_library().tgp_handoff_mode(_lib.new_ctx(0)))" % root
    r = subprocess.run([sys.executable, "-c", code], env=dict(os.environ, TGP_SYNC_EVENTS="1"), capture_output=True, text=True, timeout=300)
    assert "mode 2" in r.stdout, r.stdout + r.stderr


def _run_virtual_ranks(G, n):
    import torch
    from treegp_amd import _lib, ops
    from treegp_amd.dist import DistributedGP
    from _dist_helpers import ThreadComm
    from oracle import gp_oracle as O
    m = 777
    spec, X, y, y_err, Xs = _problem(n, m, seed=G)
    alpha_ref, logdet_ref, _, _ = ops.gp_solve(spec, X, y, y_err)
    yp_ref = ops.gp_predict(spec, X, alpha_ref, Xs)
    shared = ThreadComm.Shared(G)
    results, errors = [None] * G, []
    dev = torch.device("cuda", 0)

    def run(rank):
        try:
            ctx = _lib.new_ctx(0)
            gp = DistributedGP(ctx, spec, X, y, y_err, Xs, comm=ThreadComm(shared, rank), device=dev)
            alpha, _ = gp.step()
            full = gp.gather_predictions()
            torch.cuda.synchronize()
            results[rank] = (alpha.cpu().numpy()[:n], float(gp.logdet[0]), full.cpu().numpy())
        except BaseException as e:            # noqa: BLE001 - surface any failure of a virtual rank
            errors.append(e)
            try:
                shared.barrier.abort()
            except Exception:
                pass

    threads = [threading.Thread(target=run, args=(r,)) for r in range(G)]
    for t in threads:
        t.start()
    for t in threads:
        t.join(timeout=300)
    assert not errors, errors
    for r in range(G):
        a, ld, yp = results[r]
        np.testing.assert_allclose(a, alpha_ref, rtol=0, atol=1e-10 * np.abs(alpha_ref).max())
        np.testing.assert_allclose(ld, logdet_ref, rtol=1e-12)
        np.testing.assert_allclose(yp, yp_ref, rtol=0, atol=1e-10 * np.abs(yp_ref).max())
    # and against the oracle (north-star tolerance on predictions)
    kw = dict(amp=spec.amp, a=spec.a, b=spec.b, c=spec.c)
    a_o, _ = O.gp_solve(O.kernel_matrix("gauss", X, **kw), y, y_err)
    yp_o = O.gp_predict(O.kernel_matrix("gauss", Xs, X, **kw), a_o)
    np.testing.assert_allclose(results[0][2], yp_o, rtol=0, atol=1e-10 * np.abs(yp_o).max())
    return results


@pytest.mark.parametrize("G,n,group", [(1, 5000, 4), (2, 6000, 4), (3, 7000, 3), (4, 6000, 2), (8, 9000, 4), (5, 2300, 4)])
def test_panel_exchange_off_the_chain_virtual_ranks(G, n, group, monkeypatch):
    """TGP_DIST_CHAIN_BCAST=1 with the real kernels: the strips of the panel chain read their column operand
    from the diagonal block's broadcast (tgp_dd_strip_left) and never wait for an all-gathered panel.  Same results as the
    single-GPU path and the oracle, and bit for bit the alpha and log-determinant of the default form (same products, same
    order, one pass of depth 256 j instead of j passes of depth 256)."""
    monkeypatch.setenv("TGP_DIST_GROUP", str(group))
    monkeypatch.setenv("TGP_DIST_FINISH", "4")
    monkeypatch.setenv("TGP_DIST_CHAIN_BCAST", "0")
    ref = _run_virtual_ranks(G, n)
    monkeypatch.setenv("TGP_DIST_CHAIN_BCAST", "1")
    got = _run_virtual_ranks(G, n)
    for r in range(G):
        assert np.array_equal(ref[r][0], got[r][0]), "alpha differs on rank %d" % r
        assert ref[r][1] == got[r][1]


@pytest.mark.parametrize("G", [2, 3])
def test_sharded_pair_binning_virtual_ranks(G):
    """SURVEY 8e pair histogram: i-tiles (kk_twod / kk_log) and whole resamples (bootstrap) dealt to the
    ranks, one sum-reduction; equal to the single-GPU call and to the oracle."""
    from treegp_amd import _lib, ops
    from _dist_helpers import ThreadComm
    from oracle import gp_oracle as O
    rng = np.random.default_rng(G)
    n, nbins = 3000, 15
    x, y = rng.uniform(0, 1, n), rng.uniform(0, 1, n)
    k = rng.standard_normal(n)
    w = rng.uniform(0.5, 2.0, n)
    e = rng.uniform(0.1, 0.2, n)
    idx = O.bootstrap_indices(n, 7)
    one = (ops.kk_twod(x, y, k, w, 0.0, 0.2, nbins), ops.kk_log(x, y, k, None, 0.01, 0.5, 12),
           ops.kk_twod_bootstrap(x, y, k, e, idx, 0.0, 0.2, nbins))
    ora = O.kk_twod(x, y, k, w, 0.0, 0.2, nbins)
    shared = ThreadComm.Shared(G)
    results, errors = [None] * G, []

    def run(rank):
        try:
            ctx = _lib.new_ctx(0)
            ops.set_pair_comm(ThreadComm(shared, rank))
            results[rank] = (ops.kk_twod(x, y, k, w, 0.0, 0.2, nbins, ctx=ctx),
                             ops.kk_log(x, y, k, None, 0.01, 0.5, 12, ctx=ctx),
                             ops.kk_twod_bootstrap(x, y, k, e, idx, 0.0, 0.2, nbins, ctx=ctx))
            ops.set_pair_comm(None)
        except BaseException as ex:           # noqa: BLE001
            errors.append(ex)
            try:
                shared.barrier.abort()
            except Exception:
                pass

    threads = [threading.Thread(target=run, args=(r,)) for r in range(G)]
    for t in threads:
        t.start()
    for t in threads:
        t.join(timeout=300)
    assert not errors, errors
    for r in range(G):
        twod, lg, boot = results[r]
        np.testing.assert_array_equal(twod[2], one[0][2])                 # pair counts: exact
        np.testing.assert_array_equal(twod[2], ora[2])
        np.testing.assert_allclose(twod[0], ora[0], rtol=0, atol=1e-12 * np.abs(ora[0]).max())
        np.testing.assert_allclose(twod[1], one[0][1], rtol=1e-12)
        np.testing.assert_array_equal(lg[4], one[1][4])
        for a, b in zip(lg[:4], one[1][:4]):
            np.testing.assert_allclose(a, b, rtol=0, atol=1e-12 * np.abs(b).max())
        np.testing.assert_allclose(boot, one[2], rtol=0, atol=1e-12 * np.abs(one[2]).max())


def test_update_entry_points_agree():
    """tgp_dd_update / tgp_dd_update2 are the one- and two-panel forms of tgp_dd_update_group: same bytes out."""
    import torch
    from treegp_amd import _lib
    from treegp_amd.dist import HipLocalOps, BLK
    spec, X, y, y_err, Xs = _problem(2300, 10)
    dev = torch.device("cuda", 0)
    ops_ = HipLocalOps(_lib.new_ctx(0), spec, len(y), 1, 0, dev)
    from treegp_amd._lib import as_xy
    dX, de = ops_.to_device(as_xy(X)), ops_.to_device(y_err)
    ops_.kbuild(dX, de)
    torch.cuda.synchronize()
    base = ops_.A.clone()
    rng = np.random.default_rng(0)
    nB = ops_.nB
    g0 = ops_.to_device(1e-3 * rng.standard_normal((nB - 1) * BLK * BLK))       # stand-ins for gathered panels 0 and 1
    g1 = ops_.to_device(1e-3 * rng.standard_normal((nB - 2) * BLK * BLK))
    outs = []
    for variant in range(2):
        ops_.A.copy_(base)
        if variant == 0:
            ops_.update(0, g0, nB - 1)
            ops_.update2(0, g0, nB - 1, g1, nB - 2, 0, -1)
        else:
            ops_.update_group(0, [g0], [nB - 1])
            ops_.update_group(0, [g0, g1], [nB - 1, nB - 2], 0, -1)
        torch.cuda.synchronize()
        outs.append(ops_.A.clone())
    assert torch.equal(outs[0], outs[1])
    assert not torch.equal(outs[0], base)
    _lib.load_library().tgp_reset_stream(ops_.ctx)


@pytest.mark.parametrize("G,g,n,ns", [(1, 0, 2300, 2), (3, 1, 5200, 4), (8, 7, 9000, 4), (8, 0, 9000, 3), (4, 2, 3000, 1)])
def test_fused_update_equals_the_two_launches(G, g, n, ns):
    """tgp_dd_update_group_fused (head columns first, the rest behind them in the same launch, XCD-balanced tail, flag
    published from inside) writes the same bytes as the head launch + the rest launch, plain and as the persistent grid,
    on one rank's share of a G-rank layout; and it releases a stream parked on tgp_dd_wait_head."""
    import torch
    from treegp_amd import _lib
    from treegp_amd.dist import HipLocalOps, BLK, panel_cmax
    from treegp_amd._lib import as_xy
    spec, X, y, y_err, Xs = _problem(n, 10)
    dev = torch.device("cuda", 0)
    o = HipLocalOps(_lib.new_ctx(0), spec, len(y), G, g, dev, replicate=False)
    if _lib.load_library().tgp_handoff_mode(o.ctx) != 1:
        pytest.skip("hand-offs by events on this box")
    dX, de = o.to_device(as_xy(X)), o.to_device(y_err)
    o.kbuild(dX, de)
    torch.cuda.synchronize()
    base = o.A.clone()
    rng = np.random.default_rng(G * 10 + g)
    cm = [panel_cmax(s + 1, o.nB, G) for s in range(ns)]
    bufs = [o.to_device(1e-3 * rng.standard_normal(max(G * c, 1) * BLK * BLK)) for c in cm]
    outs = []
    for variant in range(4):
        o.A.copy_(base)
        if variant == 0:
            o.update_group(0, bufs, cm, 0, 2 * ns)
            o.update_group(0, bufs, cm, 2 * ns, -1)
        elif variant == 1:
            o.update_group(0, bufs, cm, 0, -1)
        else:
            o.queue_reset()
            o.update_group_fused(0, bufs, cm, 2 * ns, queue_nres=0 if variant == 2 else 2)
            o.side_wait_head()
            with o.on_side():
                marker = torch.ones(1, device=dev)          # runs only once the head is done
        torch.cuda.synchronize()
        outs.append(o.A.clone())
    assert torch.equal(outs[0], outs[1]) and torch.equal(outs[0], outs[2]) and torch.equal(outs[0], outs[3])
    assert not torch.equal(outs[0], base) or o.nloc == 0
    assert float(marker[0]) == 1.0
    _lib.load_library().tgp_reset_stream(o.ctx)


def test_driver_on_rccl_backend_world_of_one():
    """tests/_nccl_world1.py: the torch.distributed calls of the G > 1 path on the real nccl (RCCL) backend"""
    import socket
    import subprocess
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK="0", WORLD_SIZE="1", LOCAL_RANK="0")
    script = os.path.join(os.path.dirname(os.path.abspath(__file__)), "_nccl_world1.py")
    r = subprocess.run([sys.executable, script], env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and "nccl world-of-one ok" in r.stdout, r.stdout[-2000:] + r.stderr[-4000:]


@pytest.mark.parametrize("G", [2, 4])
def test_bench_self_launch_rehearsal(G):
    """`python bench.py --gpus G` as the driver types it, with no launcher around it: the parent starts G ranks
    (torch.distributed.run child, rendezvous on 127.0.0.1) before touching the GPU, rank 0 prints ONE JSON line and the
    exit code is 0.  Rehearsed here with gloo and all ranks on this box's one GPU; on a multi-GPU node the same
    command runs over nccl = RCCL, one rank per device."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, TGP_DIST_BACKEND="gloo", TGP_ONE_DEVICE="1")
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", str(G), "--steps", "2", "--warmup", "1",
                        "--ntrain", "6144", "--cpu-sample", "0"], env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    out = json.loads(lines[0])
    assert out["n_gpus"] == G and out["ranks_seen"] == G and out["steps"] == 2
    assert out["value"] > 0 and out["unit"] == "points/s" and out["scaling"] == "strong"
    # the N > 1 step goes through the drop-in API on the multi-GPU route, and the line explains itself per rank
    assert out["config"]["through_api"] is True and "api_ms_per_step" in out
    pr = out["per_rank_ms_per_step"]
    for key in ("chol_ms", "bulk_ms", "chain_ms", "gather_wait_ms", "trsv_ms", "predict_ms", "bytes_received"):
        assert len(pr[key]) == G and all(v >= 0 for v in pr[key]), key
    assert all(v > 0 for v in pr["chain_ms"]) and all(v > 0 for v in pr["bulk_ms"])
    # counted from the tensors handed to the collectives: the ideal volume 4 N^2 (G-1)/G give or take the diagonal blocks
    # (not exchanged), the padding of the gathered panels and the broadcasts (on top)
    assert all(0.9 * out["bytes_received_expected"] <= v <= 1.5 * out["bytes_received_expected"] for v in pr["bytes_received"])
    gp = out["gather_probe"]                       # dist.enable() timed both panel exchanges and rank 0 chose
    assert gp["chosen"] in ("allgather", "p2p") and gp["allgather_GBps"] > 0 and gp["p2p_GBps"] > 0
    assert out["collective_backend"]["backend"] == "gloo" and out["cpu_baseline"] == "see n_gpus=1 line"
    assert "reflected" in out["owner_map"]
    assert out["panel_chain"]["form"] == "gather"                # TGP_DIST_CHAIN_BCAST unset: the default chain
    # every rank receives the panels it does not own: ~ 4 N^2 (G-1)/G bytes (+ the diagonal-block broadcasts, + padding)
    exp = out["bytes_received_expected"]
    assert all(0.8 * exp < v < 1.6 * exp for v in pr["bytes_received"]), (pr["bytes_received"], exp)


def test_bench_meanify_recipe_through_api_rehearsal():
    """`bench.py --gpus 2 --meanify`: configs[4]'s recipe (mean-function table + y_err) through GPInterpolation on the
    multi-GPU route, rehearsed with gloo on one GPU."""
    import json
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, TGP_DIST_BACKEND="gloo", TGP_ONE_DEVICE="1")
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "1",
                        "--ntrain", "5000", "--meanify", "--cpu-sample", "0"], env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    out = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][0])
    assert out["config"]["meanify"] is True and out["config"]["through_api"] is True and out["n_gpus"] == 2


def test_bench_line_schema_single_gpu():
    """The one JSON line of `python bench.py` (contract of the round driver): keys, units and the roofline / cpu_baseline
    objects, on a small instance of the workload (the CPU leg on a 1024-point sample)."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ)
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--steps", "2", "--warmup", "1", "--ntrain", "24576",
                        "--cpu-sample", "1024"], env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1
    out = json.loads(lines[0])
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
                "dtype", "data", "config", "roofline", "cpu_baseline", "roofline_kbuild", "roofline_trsv", "roofline_predict"):
        assert key in out, key
    assert out["unit"] == "points/s" and out["dtype"] == "f64" and out["data"] == "synthetic" and out["vs_baseline"] is None
    assert out["n_gpus"] == 1 and out["steps"] == 2 and out["warmup"] == 1 and out["higher_is_better"] is True
    assert "workload" in out["config"] and "model" not in out["config"]
    rf = out["roofline"]
    assert rf["bound"] == "mfma" and rf["unit"] == "TFLOP/s" and abs(rf["frac"] - rf["achieved"] / rf["peak"]) < 1e-12
    assert 0.3 < rf["frac"] < 1.0 and "traffic" in rf and "traffic_source" in rf
    assert "pipe" in rf and (rf["pipe"] is None or 0.5 < rf["pipe"]["mfma_busy_frac"] <= 1.0)      # PMC-derived, None on other sources
    for sub in ("roofline_kbuild", "roofline_trsv"):
        assert out[sub]["bound"] == "hbm" and out[sub]["unit"] == "GB/s" and 0 < out[sub]["frac"] < 1
    cb = out["cpu_baseline"]
    for key in ("value", "unit", "cores", "blas_threads", "kind", "sample", "passes", "phases_s"):
        assert key in cb, key
    assert cb["kind"] == "port" and 1 <= cb["cores"] <= cb["blas_threads"] and cb["passes"] >= 3
    assert out["api_route_ms_per_step_one_gpu"]["total"] > 0.9 * out["ms_per_step"]       # the same work through GPInterpolation
    # configs[1] and configs[2] ride on the same line
    c0, c1, c2 = out["configs_measured"]
    assert "N=512" in c0["config"] and c0["api_route_ms"]["total"] > 0 and c0["api_route_ms"]["passes"] >= 5
    assert c1["api_route_ms"]["passes"] >= 5 and c1["api_route_ms"]["host_tax_ms"] < c1["api_route_ms"]["total"]
    assert c1["sweeps_per_solve"] == 2 and out["roofline_trsv"]["sweeps_per_solve"] == 2
    hl = out["api_route_ms"]["headline"]
    assert hl["passes"] >= 5 and hl["total_min"] <= hl["total"] <= hl["total_max"]
    cfgs = out["cpu_baseline"]["configs"]
    assert cfgs["configs[0]"]["n_train"] == 512 and cfgs["configs[0]"]["passes"] >= 5 and cfgs["configs[0]"]["value"] > 0
    c2b = cfgs["configs[2]"]
    assert c2b["n_sample"] >= 2048 and c2b["kbuild_ns_per_element"] > 0 and "EXTRAPOLATED" in c2b["extrapolated_n32768"]["label"]
    assert "kk_log_over_lds_atomic_ceiling" not in c2 and c2["kk_log_pairs_per_sec"] > 0
    for key in ("ms", "gp_solves_per_sec", "likelihood_evaluations_per_sec", "cholesky_tflops_fp64", "cholesky_frac_mfma_peak",
                "trsv_frac_hbm", "predict_pairs_per_sec", "phases_ms"):
        assert key in c1, key
    assert "N=8192" in c1["config"] and 0 < c1["cholesky_frac_mfma_peak"] < 1 and 0 < c1["trsv_frac_hbm"] < 1
    for key in ("kbuild_elements_per_sec", "kbuild_frac_of_vk_ceiling", "two_pcf_fit_ms", "kk_log_pairs_per_sec",
                "bootstrap_444_resamples_21x21_ms", "solve_plus_predict_32768_ms"):
        assert key in c2, key
    assert "N=32768" in c2["config"] and c2["two_pcf_fit_ms"] > 0
    # SURVEY 8(d) for the pair kernels and the von Karman predict (round 5): a roofline fraction each, none above 1
    for which in ("log", "twod"):
        rk = c2["roofline_kk"][which]
        assert rk["unit"] == "binned pairs/s" and rk["bound"] in ("valu", "lds_atomics") and 0 < rk["frac"] <= 1.0, rk
        assert abs(rk["frac"] - rk["achieved"] / rk["peak"]) < 1e-12 and rk["binned_pairs"] > 0
    pv = c2["roofline_predict_vk"]
    # (its peak is a probe's measured figure, not a data-sheet number: 3 % of room for the clocks of another box)
    assert pv["bound"] == "valu" and pv["unit"] == "pairs/s" and 0 < pv["frac"] <= 1.03, pv
    assert "TreeCorr absent" in c2b["pair_binning"] and "restatement" in c2b["pair_binning"]
    assert any(k.startswith("extrapolated_n") for k in cb)
    np.testing.assert_allclose(out["value"], (24576 + 4 * 24576) / (out["ms_per_step"] * 1e-3), rtol=1e-9)


def test_bench_multi_gpu_path_on_rccl_world_of_one():
    """The N > 1 leg of bench.py -- process group with the high-priority option, engine, GPInterpolation(backend="dist"),
    per-rank diagnostics -- on the REAL backend (nccl = RCCL) with one rank (TGP_BENCH_FORCE_DIST=1): everything the driver's
    multi-GPU run executes except the payload of the collectives."""
    import json
    import socket
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK="0", WORLD_SIZE="1", LOCAL_RANK="0",
               TGP_BENCH_FORCE_DIST="1")
    env.pop("TGP_DIST_BACKEND", None)
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "1", "--steps", "1", "--warmup", "1", "--ntrain",
                        "30000", "--meanify"], env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    out = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][0])
    assert out["config"]["through_api"] and out["config"]["meanify"] and out["config"]["parallelism"] == "rowcyclic1"
    assert out["roofline"]["kernel"].startswith("syrk_distn") and 0.3 < out["roofline"]["frac"] < 1.0
    assert len(out["per_rank_ms_per_step"]["chain_ms"]) == 1 and out["per_rank_ms_per_step"]["bytes_received"] == [0.0]
    assert out["gather_probe"]["skipped"] == "world of one"            # nothing to time with one rank: skipped cleanly
    cb = out["collective_backend"]
    assert cb["backend"] == "nccl" and cb["rccl_version"][0].isdigit(), cb

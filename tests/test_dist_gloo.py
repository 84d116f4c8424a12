"""World-size-2 (and 3) gloo test of the multi-GPU orchestration on CPUs: the communication
logic of treegp_amd.dist.DistributedCholesky with a NumPy stand-in for the per-rank kernels."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, n, out_dir, group=None, gather=None, finish=None):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    if finish is not None:
        os.environ["TGP_DIST_FINISH"] = str(finish)
    if group is not None:
        os.environ["TGP_DIST_GROUP"] = str(group)
    if gather is not None:
        os.environ["TGP_DIST_GATHER"] = gather
    torch.set_num_threads(1)
    lean = group is not None                 # the many-rank cases: one pass with the replicated factor, no side checks
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import sys
        sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
        from _dist_helpers import NumpyLocalOps
        from treegp_amd.dist import DistributedCholesky, TorchComm
        rng = np.random.default_rng(0)
        X = rng.uniform(0, 1, (n, 2))
        d2 = ((X[:, None, :] - X[None, :, :]) ** 2).sum(-1)
        K = np.exp(-0.5 * d2 / 0.1 ** 2) + np.diag(0.05 + 0.01 * rng.uniform(size=n))
        y = rng.standard_normal(n)
        comm = TorchComm()
        assert gather is None or comm.gather_mode == gather
        ref = np.linalg.solve(K, y)
        for replicated in ((True,) if lean else (False, True)):          # distributed sweeps / sweeps on the replicated factor
            ops = NumpyLocalOps(K, n, world, rank, replicated=replicated)
            ch = DistributedCholesky(ops, comm)
            assert group is None or ch.group == group
            assert ch.factorize() == 0
            assert world == 1 or ch.bytes_received > 0
            ypad = torch.zeros(ops.Np, dtype=torch.float64)
            ypad[:n] = torch.from_numpy(y)
            alpha = ch.solve(ypad).numpy()[:n]
            logdet = float(ch.logdet()[0])
            np.testing.assert_allclose(alpha, ref, rtol=0, atol=1e-9 * np.abs(ref).max())
            np.testing.assert_allclose(logdet, np.linalg.slogdet(K)[1], rtol=1e-11)
        if lean:
            open(os.path.join(out_dir, "ok%d" % rank), "w").write("ok")
            return
        # the replicate decision is a collective one: rank 0's environment setting wins over the others',
        # and one rank without room for the copy makes every rank fall back to the distributed sweeps
        from treegp_amd.dist import agree_replicate
        assert agree_replicate(comm, lambda env: True, env_rank0="1" if rank == 0 else "0") is True
        assert agree_replicate(comm, lambda env: env != "0", env_rank0="0" if rank == 0 else "1") is False
        assert agree_replicate(comm, lambda env: rank != world - 1) is False
        assert agree_replicate(comm, lambda env: True) is True
        # ranks that disagree anyway are stopped before the first solve can pair a collective with nothing
        with pytest.raises(RuntimeError, match="disagree"):
            DistributedCholesky(NumpyLocalOps(K, n, world, rank, replicated=(rank == 0)), comm)
        # a matrix that is not positive definite is reported on every rank
        Kbad = K.copy()
        Kbad[300, 300] = -1.0
        ch2 = DistributedCholesky(NumpyLocalOps(Kbad, n, world, rank), comm)
        assert ch2.factorize() > 0
        # ... also when the failing pivot lies in the replicated finish (the last rows, factored by every rank)
        Kbad = K.copy()
        Kbad[n - 2, n - 2] = -1.0
        ch3 = DistributedCholesky(NumpyLocalOps(Kbad, n, world, rank), comm)
        assert ch3.factorize() > 0
        open(os.path.join(out_dir, "ok%d" % rank), "w").write("ok")
    finally:
        dist.destroy_process_group()


@pytest.fixture(autouse=True)
def _one_blas_thread_per_rank(monkeypatch):
    for var in ("OMP_NUM_THREADS", "OPENBLAS_NUM_THREADS", "MKL_NUM_THREADS"):
        monkeypatch.setenv(var, "1")         # the ranks share this box's few CPUs: one BLAS thread each


@pytest.mark.parametrize("world,n", [(2, 1100), (3, 1300)])
def test_distributed_cholesky_gloo(tmp_path, world, n):
    mp.spawn(_worker, args=(world, _free_port(), n, str(tmp_path)), nprocs=world, join=True)
    assert all(os.path.exists(os.path.join(str(tmp_path), "ok%d" % r)) for r in range(world))


@pytest.mark.parametrize("world,n,group", [(4, 1900, 4), (8, 2100, 4)])
def test_distributed_cholesky_gloo_groups_of_four(tmp_path, world, n, group):
    """the schedule the headline size runs (groups of four panels, look-ahead, replicated factor) with 4 and 8 REAL
    processes: more ranks than panels per group, ranks that own no block of a group, a short last group"""
    mp.spawn(_worker, args=(world, _free_port(), n, str(tmp_path), group), nprocs=world, join=True)
    assert all(os.path.exists(os.path.join(str(tmp_path), "ok%d" % r)) for r in range(world))


@pytest.mark.parametrize("world,n,group,finish", [(2, 1100, None, 0), (3, 1300, None, 3), (4, 1900, 4, 5), (3, 1500, 2, 64)])
def test_distributed_cholesky_gloo_replicated_finish(tmp_path, world, n, group, finish):
    """TGP_DIST_FINISH: off (every panel through the distributed chain), a tail that is not a multiple of the group size, a
    tail longer than a group of four, and a tail that covers the whole matrix (every rank factors all of it); the default
    (a quarter of the blocks) is what the other tests in this file run."""
    mp.spawn(_worker, args=(world, _free_port(), n, str(tmp_path), group, None, finish), nprocs=world, join=True)
    assert all(os.path.exists(os.path.join(str(tmp_path), "ok%d" % r)) for r in range(world))


def test_distributed_cholesky_gloo_point_to_point_panel_exchange(tmp_path):
    """TGP_DIST_GATHER=p2p: every panel exchanged by one send + one receive per peer (all xGMI links in parallel on a real
    node) instead of all_gather_into_tensor; same factor, same solution -- four processes, groups of four panels"""
    world, n = 4, 1900
    mp.spawn(_worker, args=(world, _free_port(), n, str(tmp_path), 4, "p2p"), nprocs=world, join=True)
    assert all(os.path.exists(os.path.join(str(tmp_path), "ok%d" % r)) for r in range(world))


def _probe_worker(rank, world, port, out_dir):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ.pop("TGP_DIST_GATHER", None)
    torch.set_num_threads(1)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import json
        from treegp_amd.dist import TorchComm
        comm = TorchComm()
        rec = comm.probe_gather(elems=1 << 15, reps=2)
        assert rec["chosen"] in ("allgather", "p2p") and comm.gather_mode == rec["chosen"]
        assert rec["allgather_GBps"] > 0 and rec["p2p_GBps"] > 0 and comm.bytes_in == 0      # the probe's traffic is not the solve's
        # the chosen exchange delivers, and its bytes are counted from the tensors handed over
        inp = torch.full((1000,), float(rank), dtype=torch.float64)
        out = torch.empty(1000 * world, dtype=torch.float64)
        comm.all_gather_start(out, inp).wait()
        assert all(float(out[r * 1000]) == r for r in range(world)) and comm.bytes_in == 8 * 1000 * (world - 1)
        t = torch.full((10,), float(rank))
        comm.broadcast(t, 1)
        assert comm.bytes_in == 8 * 1000 * (world - 1) + (0 if rank == 1 else 40)
        # an override in rank 0's environment wins on every rank and nothing is timed
        if rank == 0:
            os.environ["TGP_DIST_GATHER"] = "p2p"
        rec2 = TorchComm().probe_gather()
        assert rec2 == {"skipped": "TGP_DIST_GATHER", "chosen": "p2p"}
        json.dump(rec, open(os.path.join(out_dir, "probe%d.json" % rank), "w"))
    finally:
        dist.destroy_process_group()


def test_panel_exchange_probe_gloo(tmp_path):
    """TorchComm.probe_gather (what dist.enable() runs on worlds of more than one rank): both panel exchanges timed on every
    rank, rank 0's clock decides, every rank ends with the same choice and the same record; four processes."""
    import json
    world = 4
    mp.spawn(_probe_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    recs = [json.load(open(os.path.join(str(tmp_path), "probe%d.json" % r))) for r in range(world)]
    assert all(r["chosen"] == recs[0]["chosen"] for r in recs)
    assert all(r["allgather_GBps"] == recs[0]["allgather_GBps"] and r["p2p_GBps"] == recs[0]["p2p_GBps"] for r in recs)


def _pair_worker(rank, world, port, out_dir):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import sys
        sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
        from _dist_helpers import numpy_kk_partial
        from treegp_amd import ops
        from treegp_amd.dist import TorchComm
        from oracle import gp_oracle as O
        rng = np.random.default_rng(5)
        n, nbins = 1500, 11
        x, y = rng.uniform(0, 1, n), rng.uniform(0, 1, n)
        k = rng.standard_normal(n)
        w = rng.uniform(0.5, 2.0, n)
        # the per-rank kernels are replaced by NumPy; what is under test is the sharding + reduction
        ops.kk_partial = numpy_kk_partial

        def boot_local(ctx, lib, x, y, yv, e, idx, min_sep, max_sep, nbins):
            rows = []
            for r in idx:
                kr = yv[r] - np.mean(yv[r])
                wr = None if (e is None or np.sum(e[r]) == 0) else 1.0 / e[r] ** 2
                rows.append(O.kk_twod(x[r], y[r], kr, wr, min_sep, max_sep, nbins)[0])
            return np.array(rows)
        ops._kk_twod_bootstrap_local = boot_local
        ops._lib.get_ctx = lambda device=None: None
        ops._lib.load_library = lambda: None
        ops.set_pair_comm(TorchComm())
        xi, wt, npairs = ops.kk_twod(x, y, k, w, 0.0, 0.2, nbins)
        rxi, rwt, rn = O.kk_twod(x, y, k, w, 0.0, 0.2, nbins)
        np.testing.assert_array_equal(npairs, rn)
        np.testing.assert_allclose(wt, rwt, rtol=1e-13)
        np.testing.assert_allclose(xi, rxi, rtol=0, atol=1e-13 * np.abs(rxi).max())
        out = ops.kk_log(x, y, k, None, 0.01, 0.5, 9)
        ref = O.kk_log(x, y, k, None, 0.01, 0.5, 9)
        np.testing.assert_array_equal(out[4], ref[4])
        for a, b in zip(out[:4], ref[:4]):
            np.testing.assert_allclose(a, b, rtol=0, atol=1e-13 * np.abs(b).max())
        idx = O.bootstrap_indices(300, 5)
        e = rng.uniform(0.1, 0.2, 300)
        xb = ops.kk_twod_bootstrap(x[:300], y[:300], k[:300], e, idx, 0.0, 0.3, 7)
        ref = boot_local(None, None, x[:300], y[:300], k[:300], e, idx, 0.0, 0.3, 7)
        np.testing.assert_array_equal(xb, ref)
        ops.set_pair_comm(None)
        open(os.path.join(out_dir, "ok%d" % rank), "w").write("ok")
    finally:
        dist.destroy_process_group()


def test_sharded_pair_binning_gloo(tmp_path):
    world = 2
    mp.spawn(_pair_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    assert all(os.path.exists(os.path.join(str(tmp_path), "ok%d" % r)) for r in range(world))


def _stall_worker(rank, world, port, out_dir):
    """rank 1 goes silent for a while in the middle of the factorisation; the watchdog limit is 1 s"""
    import time
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ["TGP_DIST_WATCHDOG_S"] = "1"
    os.environ["TGP_DIST_FINISH"] = "0"
    torch.set_num_threads(1)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import sys
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    from _dist_helpers import NumpyLocalOps
    from treegp_amd.dist import DistributedCholesky, DistStall, TorchComm
    n = 1300
    rng = np.random.default_rng(0)
    X = rng.uniform(0, 1, (n, 2))
    d2 = ((X[:, None, :] - X[None, :, :]) ** 2).sum(-1)
    K = np.exp(-0.5 * d2 / 0.1 ** 2) + np.diag(0.05 + 0.01 * rng.uniform(size=n))
    comm = TorchComm()
    if rank == 1:                                   # the third broadcast this rank takes part in comes 4 s late
        calls, orig = [0], comm.broadcast

        def late(t, src):
            calls[0] += 1
            if calls[0] == 3:
                time.sleep(4.0)
            return orig(t, src)
        comm.broadcast = late
    ch = DistributedCholesky(NumpyLocalOps(K, n, world, rank), comm)
    t0 = time.monotonic()
    try:
        ch.factorize()
        verdict = "finished"
    except DistStall as e:
        verdict = "stall %.2f %s" % (time.monotonic() - t0, str(e)[:80])
    open(os.path.join(out_dir, "verdict%d" % rank), "w").write(verdict)
    os._exit(0)                                     # the group is unusable after a time-out: no orderly shutdown to wait for


def test_watchdog_turns_a_stalled_rank_into_an_error_on_every_rank(tmp_path):
    """a rank that stops taking part must not leave the others waiting for ever.  With
    TGP_DIST_WATCHDOG_S=1 the ranks that wait for the silent one raise DistStall after about a second (long before it
    wakes up), and the late rank, whose partners are gone, gets the same error when it comes back -- every rank returns an
    error, none hangs, none re-executes anything."""
    world = 3
    mp.spawn(_stall_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    verdicts = [open(os.path.join(str(tmp_path), "verdict%d" % r)).read() for r in range(world)]
    for r in (0, 2):
        assert verdicts[r].startswith("stall"), verdicts
        assert float(verdicts[r].split()[1]) < 3.5, verdicts          # gave up after ~1 s, not after the 4 s nap
    assert verdicts[1].startswith("stall"), verdicts


def _chain_bcast_worker(rank, world, port, n, out_dir, group):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ["TGP_DIST_GROUP"] = str(group)
    torch.set_num_threads(1)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import sys
        sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
        from _dist_helpers import NumpyLocalOps
        from treegp_amd.dist import DistributedCholesky, TorchComm
        rng = np.random.default_rng(1)
        X = rng.uniform(0, 1, (n, 2))
        d2 = ((X[:, None, :] - X[None, :, :]) ** 2).sum(-1)
        K = np.exp(-0.5 * d2 / 0.1 ** 2) + np.diag(0.05 + 0.01 * rng.uniform(size=n))
        y = rng.standard_normal(n)
        comm = TorchComm()
        got = {}
        for form in ("0", "1"):
            os.environ["TGP_DIST_CHAIN_BCAST"] = form
            ops = NumpyLocalOps(K, n, world, rank, replicated=True)
            ch = DistributedCholesky(ops, comm)
            assert ch.factorize() == 0
            assert ch.chain_form == ("bcast" if form == "1" else "gather")
            ypad = torch.zeros(ops.Np, dtype=torch.float64)
            ypad[:n] = torch.from_numpy(y)
            got[form] = (dict((b, r.copy()) for b, r in ops.rows.items()), ch.solve(ypad).numpy()[:n].copy(), ops.Lfull.copy())
        for b in got["0"][0]:
            assert np.array_equal(np.tril(got["0"][0][b][:, :(b + 1) * 256]), np.tril(got["1"][0][b][:, :(b + 1) * 256])), b
        assert np.array_equal(got["0"][2], got["1"][2])           # the replicated factor, bit for bit
        assert np.array_equal(got["0"][1], got["1"][1])
        np.testing.assert_allclose(got["1"][1], np.linalg.solve(K, y), rtol=0, atol=1e-9 * np.abs(got["1"][1]).max())
        open(os.path.join(out_dir, "ok%d" % rank), "w").write("ok")
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,n,group", [(2, 1300, 2), (3, 1900, 4), (4, 2100, 4), (8, 2300, 3)])
def test_panel_exchange_off_the_chain_gives_the_same_factor_bit_for_bit(tmp_path, world, n, group):
    """VERDICT r4 item 3 (TGP_DIST_CHAIN_BCAST=1): the owner of a panel's diagonal block appends its rows of the group's
    earlier panels to the broadcast, every rank updates the panel's columns in one left-looking strip from that, and the
    all-gathers feed the bulk update only.  Same products in the same order as the right-looking strips: the local shares,
    the replicated factor and the solution are bit-identical to the default form; worlds of 2, 3, 4 and 8 processes,
    groups of 2, 3 and 4 panels (ranks that own no block of a group, a short last group)."""
    mp.spawn(_chain_bcast_worker, args=(world, _free_port(), n, str(tmp_path), group), nprocs=world, join=True)
    assert all(os.path.exists(os.path.join(str(tmp_path), "ok%d" % r)) for r in range(world))

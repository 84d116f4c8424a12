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


def _worker(rank, world, port, n, out_dir):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
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
        ops = NumpyLocalOps(K, n, world, rank)
        ch = DistributedCholesky(ops, comm)
        assert ch.factorize() == 0
        ypad = torch.zeros(ops.Np, dtype=torch.float64)
        ypad[:n] = torch.from_numpy(y)
        alpha = ch.solve(ypad).numpy()[:n]
        logdet = float(ch.logdet()[0])
        ref = np.linalg.solve(K, y)
        np.testing.assert_allclose(alpha, ref, rtol=0, atol=1e-9 * np.abs(ref).max())
        np.testing.assert_allclose(logdet, np.linalg.slogdet(K)[1], rtol=1e-11)
        # a matrix that is not positive definite is reported on every rank
        Kbad = K.copy()
        Kbad[300, 300] = -1.0
        ch2 = DistributedCholesky(NumpyLocalOps(Kbad, n, world, rank), comm)
        assert ch2.factorize() > 0
        open(os.path.join(out_dir, "ok%d" % rank), "w").write("ok")
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,n", [(2, 1100), (3, 1300)])
def test_distributed_cholesky_gloo(tmp_path, world, n):
    mp.spawn(_worker, args=(world, _free_port(), n, str(tmp_path)), nprocs=world, join=True)
    assert all(os.path.exists(os.path.join(str(tmp_path), "ok%d" % r)) for r in range(world))

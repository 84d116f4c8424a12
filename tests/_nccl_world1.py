"""Run by test_gpu_dist.py in a child process: the multi-GPU driver on the REAL backend (nccl = RCCL) with a world of
one rank -- the collectives are trivial, but every torch.distributed call the G > 1 runs make (async
all_gather_into_tensor on the look-ahead stream, broadcast, all_reduce, the high-priority process-group option of
bench.py) goes through RCCL on the GPU."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    import torch
    import torch.distributed as dist
    from treegp_amd import _lib, ops
    from treegp_amd.dist import DistributedGP, TorchComm
    from treegp_amd.synthetic import star_field, headline_invlam
    torch.cuda.set_device(0)
    opts = dist.ProcessGroupNCCL.Options(is_high_priority_stream=True)
    dist.init_process_group("nccl", device_id=torch.device("cuda", 0), pg_options=opts)
    n, m = 2900, 1500
    X, y, y_err, Xs = star_field(n, m, seed=3)
    y = y - y.mean()
    iL = headline_invlam()
    spec = ops.KernelSpec(_lib.TGP_ARBF, amp=1.0, a=iL[0, 0], b=iL[0, 1], c=iL[1, 1])
    alpha_ref, logdet_ref, _, _ = ops.gp_solve(spec, X, y, y_err)
    yp_ref = ops.gp_predict(spec, X, alpha_ref, Xs)
    for replicate in ("1", "0"):
        os.environ["TGP_DIST_REPLICATE"] = replicate
        ctx = _lib.new_ctx(0)
        comm = TorchComm()
        assert comm.native_gather and comm.size == 1
        gp = DistributedGP(ctx, spec, X, y, y_err, Xs, comm=comm, device=torch.device("cuda", 0))
        assert gp.ops.replicated == (replicate == "1")
        for _ in range(2):
            alpha, dys = gp.step()
        torch.cuda.synchronize()
        a = alpha.cpu().numpy()[:n]
        np.testing.assert_allclose(a, alpha_ref, rtol=0, atol=1e-11 * np.abs(alpha_ref).max())
        np.testing.assert_allclose(float(gp.logdet[0]), logdet_ref, rtol=1e-13)
        np.testing.assert_allclose(gp.gather_predictions().cpu().numpy(), yp_ref, rtol=0, atol=1e-11 * np.abs(yp_ref).max())
        _lib.load_library().tgp_reset_stream(ctx)
    # pair binning with the reduction on RCCL
    want = ops.kk_twod(X[:, 0], X[:, 1], y, None, 0.0, 0.2, 9)
    ops.set_pair_comm(TorchComm())
    got = ops.kk_twod(X[:, 0], X[:, 1], y, None, 0.0, 0.2, 9)
    ops.set_pair_comm(None)
    for a, b in zip(got, want):
        np.testing.assert_allclose(a, b, rtol=1e-12, atol=1e-300)
    dist.barrier()
    dist.destroy_process_group()
    print("nccl world-of-one ok")


if __name__ == "__main__":
    main()

"""Run under ``python -m torch.distributed.run`` by tests/test_gpu_api_dist.py (and usable by hand on a multi-GPU node):
configs[4]'s recipe -- mean-function table, non-uniform y_err, white noise, normalize -- through the drop-in API on the
multi-GPU route (``GPInterpolation(backend="dist")``), every rank comparing with the single-GPU API on its own device.

TGP_DIST_BACKEND=gloo TGP_ONE_DEVICE=1: rehearsal with all ranks on one GPU; default nccl = RCCL, one rank per GPU."""
import os
import sys
import tempfile

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 3000
    m = int(sys.argv[2]) if len(sys.argv) > 2 else 1500
    import torch
    import torch.distributed as dist
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    local = 0 if os.environ.get("TGP_ONE_DEVICE") == "1" else int(os.environ.get("LOCAL_RANK", "0"))
    os.environ["TGP_DEVICE"] = str(local)
    torch.cuda.set_device(local)
    backend = os.environ.get("TGP_DIST_BACKEND", "nccl")
    if backend == "nccl":
        dist.init_process_group("nccl", device_id=torch.device("cuda", local))
    else:
        dist.init_process_group(backend)
    import treegp_amd
    from treegp_amd import dist as tdist
    from treegp_amd.fits_io import write_bintable_row
    from treegp_amd.synthetic import star_field_with_mean, headline_kernel_string

    X, y, y_err, Xs, X0, y0 = star_field_with_mean(n, m, seed=11)
    fits = os.path.join(tempfile.mkdtemp(), "mean_%d.fits" % rank)
    write_bintable_row(fits, {"COORDS0": X0, "PARAMS0": y0})
    kw = dict(kernel=headline_kernel_string(), optimizer="none", normalize=True, white_noise=0.005, average_fits=fits)

    one = treegp_amd.GPInterpolation(backend="single", **kw)
    one.initialize(X, y, y_err)
    yp1, cov1 = one.predict(Xs[:200], return_cov=True)
    yp1_all = one.predict(Xs)
    ll1 = one.return_log_likelihood()

    eng = tdist.enable(min_n=10 ** 9)                 # the threshold says "never": only the explicit kwarg routes here
    assert eng.comm.size == world
    gp = treegp_amd.GPInterpolation(backend="dist", **kw)
    gp.initialize(X, y, y_err)
    yp_all = gp.predict(Xs)
    yp, cov = gp.predict(Xs[:200], return_cov=True)
    ll = gp.return_log_likelihood()
    scale = np.abs(yp1_all).max()
    np.testing.assert_allclose(yp_all, yp1_all, rtol=0, atol=1e-10 * scale)
    np.testing.assert_allclose(yp, yp1, rtol=0, atol=1e-10 * scale)
    np.testing.assert_allclose(cov, cov1, rtol=0, atol=1e-9 * np.abs(cov1).max())
    np.testing.assert_allclose(ll, ll1, rtol=1e-11)
    # the threshold route: an engine with min_n <= n takes objects that say nothing about the backend
    tdist.disable()
    tdist.enable(min_n=n)
    gp2 = treegp_amd.GPInterpolation(**kw)
    gp2.initialize(X, y, y_err)
    np.testing.assert_allclose(gp2.predict(Xs), yp1_all, rtol=0, atol=1e-10 * scale)
    # not positive definite: the same LinAlgError on every rank
    bad = treegp_amd.GPInterpolation(kernel=headline_kernel_string(), optimizer="none", backend="dist")
    Xd = X.copy()
    Xd[1] = Xd[0]
    bad.initialize(Xd, y, np.zeros(n))
    try:
        bad.predict(Xs[:10])
        raise AssertionError("duplicate points without noise must not factorise")
    except np.linalg.LinAlgError:
        pass
    tdist.disable()
    dist.barrier()
    dist.destroy_process_group()
    if rank == 0:
        print("api dist ok: world %d n %d" % (world, n))


if __name__ == "__main__":
    main()

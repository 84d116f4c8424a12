"""An ORDINARY single-process script (no torchrun, no torch.distributed in sight) that uses the drop-in API on the multi-GPU
route -- run by tests/test_gpu_api_dist.py with TGP_DIST_BACKEND=gloo TGP_ONE_DEVICE=1 TGP_DIST_POOL_WORLD=G on a one-GPU
box; on a multi-GPU node, with nothing set, the pool takes one worker per visible GPU over RCCL.
usage: python _api_pool_script.py [N] [M] [explicit|env]"""
import os
import sys
import tempfile

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 3000
    m = int(sys.argv[2]) if len(sys.argv) > 2 else 1500
    how = sys.argv[3] if len(sys.argv) > 3 else "explicit"
    import treegp_amd
    from treegp_amd.fits_io import write_bintable_row
    from treegp_amd.synthetic import star_field_with_mean, headline_kernel_string

    X, y, y_err, Xs, X0, y0 = star_field_with_mean(n, m, seed=11)
    fits = os.path.join(tempfile.mkdtemp(), "mean.fits")
    write_bintable_row(fits, {"COORDS0": X0, "PARAMS0": y0})
    kw = dict(kernel=headline_kernel_string(), optimizer="none", normalize=True, white_noise=0.005, average_fits=fits)

    one = treegp_amd.GPInterpolation(backend="single", **kw)
    one.initialize(X, y, y_err)
    ref = one.predict(Xs)
    ll_ref = one.return_log_likelihood()

    if how == "explicit":
        gp = treegp_amd.GPInterpolation(backend="dist", **kw)        # the only line that differs from a single-GPU script
    else:
        assert os.environ.get("TGP_DIST") == "1"                      # ... or none at all: the environment routes it
        gp = treegp_amd.GPInterpolation(**kw)
    gp.initialize(X, y, y_err)
    got = gp.predict(Xs)
    got2 = gp.predict(Xs[:100])                                       # second call: the workers (and alpha) are still there
    ll = gp.return_log_likelihood()
    from treegp_amd import dist
    eng = dist._process_engine
    assert eng is not None and type(eng).__name__ == "PoolEngine", eng
    world = eng.world
    scale = np.abs(ref).max()
    np.testing.assert_allclose(got, ref, rtol=0, atol=1e-10 * scale)
    np.testing.assert_allclose(got2, ref[:100], rtol=0, atol=1e-10 * scale)
    np.testing.assert_allclose(ll, ll_ref, rtol=1e-11)
    # a matrix that is not positive definite is a LinAlgError here too, and the pool survives it
    bad = treegp_amd.GPInterpolation(backend="dist", kernel=headline_kernel_string(), optimizer="none", normalize=False)
    Xd = X.copy()
    Xd[1] = Xd[0]
    try:
        bad.initialize(Xd, y, np.zeros(n))
        bad.predict(Xs[:10])
        raise AssertionError("expected LinAlgError")
    except np.linalg.LinAlgError:
        pass
    again = gp.predict(Xs[:50])
    np.testing.assert_allclose(again, ref[:50], rtol=0, atol=1e-10 * scale)
    dist.disable()
    print("POOL_OK world=%d n=%d max|diff|/scale=%.2e" % (world, n, np.abs(got - ref).max() / scale))


if __name__ == "__main__":
    main()

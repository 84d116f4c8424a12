#!/usr/bin/env python3
"""Wall-clock of the BASELINE.json configurations through the public Python API (GPInterpolation), as a user
would call it.  Development aid / README numbers."""
import sys
import time

import numpy as np

sys.path.insert(0, __file__.rsplit("/tools/", 1)[0])
import treegp_amd as treegp  # noqa: E402
from treegp_amd.synthetic import star_field, headline_invlam  # noqa: E402


def timed(label, fn):
    t0 = time.perf_counter()
    out = fn()
    print("%-70s %9.1f ms" % (label, (time.perf_counter() - t0) * 1e3), flush=True)
    return out


def main():
    iL = headline_invlam()
    kern = "1.0**2 * AnisotropicRBF(invLam=array(%s))" % np.array2string(iL, separator=",", precision=17)
    # C2: N=8192, predict 32768
    X, y, ye, Xs = star_field(8192, 32768)
    gp = treegp.GPInterpolation(kernel=kern, optimizer="none", normalize=True)
    gp.initialize(X, y, y_err=ye)
    gp.predict(Xs[:10])
    gp.initialize(X, y, y_err=ye)
    timed("C2  N=8192 AnisotropicRBF: initialize + predict(32768)", lambda: (gp.initialize(X, y, y_err=ye), gp.predict(Xs)))
    timed("C2  predict(32768) again (alpha cached)", lambda: gp.predict(Xs))
    timed("C2  predict(4096, return_cov=True)", lambda: gp.predict(Xs[:4096], return_cov=True))
    # C3: VonKarman N=32768 with two-pcf fit
    X, y, ye, Xs = star_field(32768, 32768)
    gp = treegp.GPInterpolation(kernel="1.0**2 * VonKarman(length_scale=0.1)", optimizer="two-pcf", nbins=20, normalize=True)
    timed("C3  N=32768 VonKarman: initialize", lambda: gp.initialize(X, y, y_err=ye))
    timed("C3  solve() (isotropic two-pcf fit)", gp.solve)
    timed("C3  predict(32768)", lambda: gp.predict(Xs))
    # anisotropic variant with the 444-resample bootstrap
    gp = treegp.GPInterpolation(kernel="1.0**2 * AnisotropicVonKarman(invLam=array(%s))" % np.array2string(iL, separator=",", precision=17),
                                optimizer="anisotropic", nbins=21, min_sep=0.0, max_sep=0.15, p0=[0.05, 0.0, 0.0], normalize=True)
    gp.initialize(X, y, y_err=ye)
    timed("C3b N=32768 AnisotropicVonKarman: solve() (TwoD pcf + 444 bootstrap + robust fit)", gp.solve)
    timed("C3b predict(32768)", lambda: gp.predict(Xs))
    # C4 through the API on one GPU
    X, y, ye, Xs = star_field(65536, 262144)
    gp = treegp.GPInterpolation(kernel=kern, optimizer="none", normalize=True)
    gp.initialize(X, y, y_err=ye)
    timed("C4  N=65536 AnisotropicRBF: predict(262144) incl. factorisation", lambda: gp.predict(Xs))


if __name__ == "__main__":
    main()

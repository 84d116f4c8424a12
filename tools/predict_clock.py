#!/usr/bin/env python3
"""Shader clock (rocm-smi) while the fused predict kernel runs back to back: what its 22-instructions-per-pair loop is priced against."""
import subprocess
import sys
import threading
import time

import numpy as np

sys.path.insert(0, __file__.rsplit("/tools/", 1)[0])
from treegp_amd import _lib, ops  # noqa: E402
from treegp_amd.synthetic import star_field, headline_invlam  # noqa: E402

samples, stop = [], False


def poll():
    while not stop:
        try:
            out = subprocess.run(["rocm-smi", "-d", "0", "--showclocks", "--showpower", "--csv"], capture_output=True, text=True, timeout=5).stdout
            f = out.strip().splitlines()[1].split(",")
            samples.append((time.perf_counter(), f[5], f[9]))
        except Exception:  # noqa: BLE001
            pass
        time.sleep(0.1)


def main():
    global stop
    n, m = 65536, 262144
    X, y, ye, Xs = star_field(n, m)
    iL = headline_invlam()
    spec = ops.KernelSpec(_lib.TGP_ARBF, amp=1.0, a=iL[0, 0], b=iL[0, 1], c=iL[1, 1])
    alpha = np.random.default_rng(0).standard_normal(n)
    ops.gp_predict(spec, X, alpha, Xs)
    th = threading.Thread(target=poll)
    th.start()
    t0 = time.perf_counter()
    reps = 150
    dev = []
    for _ in range(reps):
        ops.gp_predict(spec, X, alpha, Xs)
        dev.append(_lib.timings(_lib.get_ctx())[3])
    wall = time.perf_counter() - t0
    stop = True
    th.join()
    ms = float(np.median(dev))
    print("predict N=%d M=%d: %.3f ms device (median of %d), %.3e pairs/s; %.2f s wall" % (n, m, ms, reps, n * m / ms * 1e3, wall))
    print(" ".join("%.1fs:%s/%sW" % (t - t0, c.strip("()"), p) for t, c, p in samples))


if __name__ == "__main__":
    main()

#!/usr/bin/env python3
"""Development aid: sample GPU clock / power (rocm-smi) while (a) the vendor fp64 GEMM and (b) this
library's Cholesky run, to see whether the trailing update is clock(power)-limited."""
import ctypes as C
import subprocess
import sys
import threading
import time

import numpy as np

sys.path.insert(0, __file__.rsplit("/tools/", 1)[0])

samples = []
stop = False


def poll():
    while not stop:
        try:
            out = subprocess.run(["rocm-smi", "-d", "0", "--showclocks", "--showpower", "--csv"], capture_output=True, text=True, timeout=5).stdout
            samples.append((time.perf_counter(), out.strip().splitlines()))
        except Exception as e:  # noqa: BLE001
            samples.append((time.perf_counter(), ["ERR %r" % (e,)]))
        time.sleep(0.05)


def main():
    global stop
    import torch
    from treegp_amd import _lib, ops
    from treegp_amd.synthetic import star_field, headline_invlam
    th = threading.Thread(target=poll)
    th.start()
    dev = torch.device("cuda", 0)
    n, k = 32768, 512
    a = torch.randn(n, k, dtype=torch.float64, device=dev)
    b = torch.randn(n, k, dtype=torch.float64, device=dev)
    c = torch.zeros(n, n, dtype=torch.float64, device=dev)
    torch.cuda.synchronize()
    marks = [("vendor_start", time.perf_counter())]
    for _ in range(200):
        torch.addmm(c, a, b.t(), beta=1.0, alpha=-1.0, out=c)
    torch.cuda.synchronize()
    marks.append(("vendor_end", time.perf_counter()))
    print("vendor: %.1f TF" % (200 * 2.0 * n * n * k / (marks[-1][1] - marks[-2][1]) / 1e12))
    del a, b, c
    time.sleep(1.0)
    lib = _lib.load_library()
    ctx = _lib.get_ctx()
    N = 65536
    X, y, ye, Xs = star_field(N, 16)
    iL = headline_invlam()
    spec = ops.KernelSpec(_lib.TGP_ARBF, amp=1.0, a=iL[0, 0], b=iL[0, 1], c=iL[1, 1])
    dX = ops.DeviceBuffer.from_array(ctx, X); dy = ops.DeviceBuffer.from_array(ctx, y - y.mean()); de = ops.DeviceBuffer.from_array(ctx, ye)
    da = ops.DeviceBuffer(ctx, N * 8)
    ld, yd = C.c_double(), C.c_double()
    marks.append(("chol_start", time.perf_counter()))
    for _ in range(2):
        lib.tgp_d_gp_solve(ctx, C.byref(spec.to_c()), dX.ptr, N, dy.ptr, de.ptr, da.ptr, C.byref(ld), C.byref(yd), None)
    marks.append(("chol_end", time.perf_counter()))
    print("chol: %.1f TF" % (2 * N ** 3 / 3 / (marks[-1][1] - marks[-2][1]) / 1e12))
    stop = True
    th.join()
    print("header:", samples[0][1][0] if samples else None)
    for name, t in marks:
        print("MARK %s %.3f" % (name, t - marks[0][1]))
    for t, lines in samples:
        print("%.3f %s" % (t - marks[0][1], " | ".join(lines[1:])))


if __name__ == "__main__":
    main()

#!/usr/bin/env python3
"""Calibration only: what the vendor library (rocBLAS / hipBLASLt through torch.mm) reaches for an fp64
NT GEMM and a Cholesky on this box, to put the hand-written trailing update in context.  Nothing in the
product calls these."""
import time

import torch


def main():
    dev = torch.device("cuda", 0)
    for n, k in ((16384, 512), (32768, 512), (16384, 16384)):
        a = torch.randn(n, k, dtype=torch.float64, device=dev)
        b = torch.randn(n, k, dtype=torch.float64, device=dev)
        c = torch.zeros(n, n, dtype=torch.float64, device=dev)
        for _ in range(2):
            torch.addmm(c, a, b.t(), beta=1.0, alpha=-1.0, out=c)
        torch.cuda.synchronize()
        reps = 10 if k <= 512 else 3
        t0 = time.perf_counter()
        for _ in range(reps):
            torch.addmm(c, a, b.t(), beta=1.0, alpha=-1.0, out=c)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / reps
        print("vendor dgemm NT C(%d,%d) -= A(%d,%d) B^T: %.2f ms  %.1f TFLOP/s" % (n, n, n, k, dt * 1e3, 2.0 * n * n * k / dt / 1e12), flush=True)
        del a, b, c
    # sustained: does the rate hold over seconds (power / clock management)?
    n, k = 32768, 512
    a = torch.randn(n, k, dtype=torch.float64, device=dev)
    b = torch.randn(n, k, dtype=torch.float64, device=dev)
    c = torch.zeros(n, n, dtype=torch.float64, device=dev)
    for chunk in range(8):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(20):
            torch.addmm(c, a, b.t(), beta=1.0, alpha=-1.0, out=c)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / 20
        print("sustained chunk %d: %.2f ms  %.1f TFLOP/s" % (chunk, dt * 1e3, 2.0 * n * n * k / dt / 1e12), flush=True)
    del a, b, c
    for n in (8192, 16384, 32768):
        x = torch.randn(n, 64, dtype=torch.float64, device=dev)
        a = x @ x.t() + n * torch.eye(n, dtype=torch.float64, device=dev)
        torch.cuda.synchronize()
        torch.linalg.cholesky(a)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        torch.linalg.cholesky(a)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        print("vendor potrf (torch.linalg.cholesky) N=%d: %.1f ms  %.1f TFLOP/s" % (n, dt * 1e3, n ** 3 / 3 / dt / 1e12), flush=True)
        del a, x


if __name__ == "__main__":
    main()

#!/usr/bin/env python3
"""Soak of everything that hands over without a kernel boundary or an event: the factorisation's cross-stream hand-offs (flags +
stream wait-value, round 3) and the multi-GPU driver's fused bulk launches that release the panel chain from inside the
kernel (round 4).  Many solves
per size over every schedule regime, each compared BIT FOR BIT with the first solve of its size -- a lost or early hand-off
shows up as a different bit (or a hang: run under `timeout -k 10 ...`).  Then the same from two host threads with a context
each, then the multi-GPU driver with a world of one and with four virtual ranks.
usage: soak_handoffs.py [seconds per size, default 12]"""
import sys
import threading
import time

import numpy as np

sys.path.insert(0, __file__.rsplit("/tools/", 1)[0])
from treegp_amd import _lib, ops  # noqa: E402
from treegp_amd.synthetic import star_field, headline_invlam  # noqa: E402

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 12.0
iL = headline_invlam()
spec = ops.KernelSpec(_lib.TGP_ARBF, amp=1.0, a=iL[0, 0], b=iL[0, 1], c=iL[1, 1])


def soak(n, seconds, tag=""):
    X, y, ye, _ = star_field(n, 16, seed=n)
    y = y - y.mean()
    ref = None
    t0, count = time.time(), 0
    while time.time() - t0 < seconds or count < 3:
        a, ld, yd, _ = ops.gp_solve(spec, X, y, ye)
        got = (a.tobytes(), float(ld), float(yd))
        if ref is None:
            ref = got
        elif got != ref:
            raise SystemExit("%sN=%d: solve %d differs from the first one (max |d alpha| %.3e, logdet %r vs %r)" % (
                tag, n, count, np.abs(np.frombuffer(got[0]) - np.frombuffer(ref[0])).max(), got[1], ref[1]))
        count += 1
    print("%sN=%6d: %4d identical solves in %.1f s" % (tag, n, count, time.time() - t0), flush=True)


sizes = (1536, 2000, 2560, 3072, 4096, 5000, 6144, 8192, 10240, 12288, 16384, 20480, 22528, 24576, 32768)
for n in sizes:
    soak(n, budget)


def worker(i):
    _lib.set_thread_ctx(_lib.new_ctx(0))
    for n in (2560, 4096, 8192):
        soak(n, budget / 2, tag="thread %d " % i)


ths = [threading.Thread(target=worker, args=(i,)) for i in range(2)]
for t in ths:
    t.start()
for t in ths:
    t.join()


def soak_dist(n, G, seconds):
    """the multi-GPU driver (fused launches, replicated finish) with G virtual ranks on this GPU: alpha bit for bit over repetitions"""
    import os
    import torch
    sys.path.insert(0, __file__.rsplit("/tools/", 1)[0] + "/tests")
    from _dist_helpers import ThreadComm
    from treegp_amd.dist import DistributedGP, SelfComm
    X, y, ye, Xs = star_field(n, 64, seed=n)
    y = y - y.mean()
    dev = torch.device("cuda", 0)
    shared = ThreadComm.Shared(G)
    out, errs = [None] * G, []

    def rank(r):
        try:
            gp = DistributedGP(_lib.new_ctx(0), spec, X, y, ye, Xs, comm=(ThreadComm(shared, r) if G > 1 else SelfComm()), device=dev)
            ref, t0, count = None, time.time(), 0
            while True:
                alpha, _ = gp.step()
                torch.cuda.synchronize()
                got = alpha.cpu().numpy().tobytes()
                if ref is None:
                    ref = got
                elif got != ref:
                    raise RuntimeError("rank %d of %d, N=%d: step %d differs from the first" % (r, G, n, count))
                count += 1
                # every rank takes the same number of steps: the decision to stop is rank 0's, published through the barrier
                if r == 0:
                    shared.stop = time.time() - t0 >= seconds and count >= 3
                if G > 1:
                    shared.barrier.wait()
                if getattr(shared, "stop", False):
                    break
            out[r] = count
        except BaseException as ex:      # noqa: BLE001
            errs.append(ex)
            try:
                shared.barrier.abort()
            except Exception:
                pass

    ths = [threading.Thread(target=rank, args=(r,)) for r in range(G)]
    for t in ths:
        t.start()
    for t in ths:
        t.join()
    if errs:
        raise SystemExit("dist soak failed: %r" % (errs[0],))
    print("multi-GPU driver, %d virtual rank(s), N=%6d: %4d identical steps" % (G, n, out[0]), flush=True)


for n, G in ((8192, 1), (24576, 1), (40000, 1), (12000, 4), (30000, 4)):
    soak_dist(n, G, budget / 2)
print("soak ok")

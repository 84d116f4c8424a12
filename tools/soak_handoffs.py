#!/usr/bin/env python3
"""Soak of the factorisation's cross-stream hand-offs (flags + stream wait-value, round 3): many solves per size over every
schedule regime, each compared BIT FOR BIT with the first solve of its size -- a lost or early hand-off shows up as a different
bit (or a hang: run under `timeout -k 10 ...`).  Then the same from two host threads with a context each.
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
print("soak ok")

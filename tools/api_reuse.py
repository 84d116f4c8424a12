#!/usr/bin/env python3
"""Where do the 20 ms go that bench.py's API passes spend outside their device phases?  Variants of the same calls in
a fresh process, each printing wall against device phases per pass:  A  one GPInterpolation reused;  B  the same after two
device-resident steps with resident buffers kept alive (bench.py's state);  C  as B with the resident buffers freed first.
usage: api_reuse.py [N=65536]     (TGP_HOST_PHASES=1 adds the library's own printout)"""
import ctypes as C
import sys
import time

import numpy as np

sys.path.insert(0, __file__.rsplit("/tools/", 1)[0])
import treegp_amd  # noqa: E402
from treegp_amd import _lib, ops  # noqa: E402
from treegp_amd.synthetic import star_field, headline_invlam, headline_kernel_string  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
X, y, ye, Xs = star_field(n, 4 * n)
lib, ctx = _lib.load_library(), _lib.get_ctx()


def passes(tag, k=4):
    gp = treegp_amd.GPInterpolation(kernel=headline_kernel_string(), optimizer="none", normalize=True, white_noise=0.0, backend="single")
    for it in range(k):
        t0 = time.perf_counter()
        gp.initialize(X, y, ye)
        gp.predict(Xs)
        wall = (time.perf_counter() - t0) * 1e3
        tm = _lib.timings(ctx)
        dev = tm[0] + tm[1] + tm[2] + tm[3]
        print("%s pass %d: wall %.1f ms, device phases %.1f ms, outside %.1f ms" % (tag, it, wall, dev, wall - dev), flush=True)


passes("A (reused object, fresh process)")
iL = headline_invlam()
spec = ops.KernelSpec(_lib.TGP_ARBF, amp=1.0, a=iL[0, 0], b=iL[0, 1], c=iL[1, 1])
kc = spec.to_c()
bufs = [ops.DeviceBuffer.from_array(ctx, a) for a in (X, y - y.mean(), ye, Xs)]
da, dys = ops.DeviceBuffer(ctx, n * 8), ops.DeviceBuffer(ctx, 4 * n * 8)
ld, yd = C.c_double(), C.c_double()
for _ in range(2):
    lib.tgp_d_gp_solve(ctx, C.byref(kc), bufs[0].ptr, n, bufs[1].ptr, bufs[2].ptr, da.ptr, C.byref(ld), C.byref(yd), None)
    lib.tgp_d_gp_predict(ctx, C.byref(kc), bufs[0].ptr, n, da.ptr, bufs[3].ptr, 4 * n, dys.ptr)
passes("B (after device-resident steps, resident buffers alive)")
for b in bufs + [da, dys]:
    b.free()
passes("C (resident buffers freed)")
# D: bench.py's exact state: per-launch profiling ON during device-resident steps, OFF for the API passes
bufs = [ops.DeviceBuffer.from_array(ctx, a) for a in (X, y - y.mean(), ye, Xs)]
da, dys = ops.DeviceBuffer(ctx, n * 8), ops.DeviceBuffer(ctx, 4 * n * 8)
lib.tgp_set_profiling(ctx, 1)
for _ in range(2):
    lib.tgp_d_gp_solve(ctx, C.byref(kc), bufs[0].ptr, n, bufs[1].ptr, bufs[2].ptr, da.ptr, C.byref(ld), C.byref(yd), None)
    lib.tgp_d_gp_predict(ctx, C.byref(kc), bufs[0].ptr, n, da.ptr, bufs[3].ptr, 4 * n, dys.ptr)
lib.tgp_set_profiling(ctx, 0)
passes("D (profiled device-resident steps first, profiling off now)")
lib.tgp_set_profiling(ctx, 1)
passes("E (profiling on)", 3)

#!/usr/bin/env python3
"""Fused predict alone (random alpha, no solve) at the three BASELINE shapes, best of 6: pairs/s.  A/B two builds with TGP_LIB_PATH=...; --vk: von Karman (generic kernel)."""
import ctypes as C
import sys

import numpy as np

sys.path.insert(0, __file__.rsplit("/tools/", 1)[0])
from treegp_amd import _lib, ops  # noqa: E402
from treegp_amd.synthetic import star_field, headline_invlam  # noqa: E402

lib, ctx = _lib.load_library(), _lib.get_ctx()
lib.tgp_set_profiling(ctx, 1)
iL = headline_invlam()
spec = ops.KernelSpec(_lib.TGP_ARBF, amp=1.0, a=iL[0, 0], b=iL[0, 1], c=iL[1, 1])
vk = "--vk" in sys.argv
if vk:
    sys.argv.remove("--vk")
    spec = ops.KernelSpec(_lib.TGP_VK, amp=1.0, ell=0.1)
for n in [int(v) for v in sys.argv[1:]] or (8192, 32768, 65536):
    X, y, ye, Xs = star_field(n, 4 * n)
    alpha = np.random.default_rng(1).standard_normal(n)
    dX, da, dXs = (ops.DeviceBuffer.from_array(ctx, a) for a in (X, alpha, Xs))
    dys = ops.DeviceBuffer(ctx, 4 * n * 8)
    best = 1e9
    for _ in range(6):
        rc = lib.tgp_d_gp_predict(ctx, C.byref(spec.to_c()), dX.ptr, n, da.ptr, dXs.ptr, 4 * n, dys.ptr)
        assert rc == 0
        best = min(best, _lib.timings(ctx)[3])
    print("N %6d M %7d  predict %.3f ms  %.3fe12 pairs/s" % (n, 4 * n, best, 4.0 * n * n / best / 1e9), flush=True)
    for b in (dX, da, dXs, dys):
        b.free()

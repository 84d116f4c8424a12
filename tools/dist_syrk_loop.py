#!/usr/bin/env python3
"""The multi-GPU bulk update (syrk_distn_kernel<4>, a world of one) alone, back to back on random data, beside the single-GPU
depth-1024 kernel on the same amount of work (tools/syrk_loop.py with TGP_DEBUG_SEGS=1): is the difference seen inside the
factorisation (65.7 vs 69.0 TF) the kernel's own or its surroundings'?  usage: dist_syrk_loop.py [Np=32768] [reps=20]"""
import ctypes as C
import os
import sys

import numpy as np
import torch

sys.path.insert(0, __file__.rsplit("/tools/", 1)[0])
from treegp_amd import _lib, ops  # noqa: E402
from treegp_amd.dist import HipLocalOps, BLK  # noqa: E402

Np = int(sys.argv[1]) if len(sys.argv) > 1 else 32768
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 20
dev = torch.device("cuda", 0)
spec = ops.KernelSpec(_lib.TGP_ARBF, amp=1.0, a=1.0, b=0.0, c=1.0)
o = HipLocalOps(_lib.new_ctx(0), spec, Np, 1, 0, dev)
o.A.copy_(1e-3 * torch.randn(o.A.numel(), dtype=torch.float64, device=dev))
nB = o.nB
GS = 4
# the gathered panels of group 0 in a world of one ARE the rank's own rows below each panel's diagonal block
bufs = [o.panel_send_view(k, nB - k - 1) for k in range(GS)]
cm = [nB - k - 1 for k in range(GS)]
for queue in (0, 1):
    if queue:
        o.queue_reset()
    o.update_group(0, bufs, cm, 2 * GS, -1, queue_nres=queue)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        o.update_group(0, bufs, cm, 2 * GS, -1, queue_nres=queue)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / reps
    m = (nB - GS) * BLK - 2 * GS * 128 / 2                   # rows of the trailing matrix right of the next group's columns (approx.)
    T = 2 * (nB - GS)
    tiles = T * (T + 1) / 2 - (2 * GS) * (2 * GS + 1) / 2 - (T - 2 * GS) * 0    # lower-triangle tiles with column >= 2 GS
    tiles = sum(max(0, ti - 2 * GS + 1) for ti in range(T))
    flops = tiles * 2.0 * 128 * 128 * 256 * GS
    print("syrk_distn_kernel<4>%s: Np=%d  %.3f ms/launch  %.2f TFLOP/s (%d tiles)" % (" queued(1)" if queue else "", Np, ms, flops / ms / 1e9, tiles), flush=True)
    if reps > 40:
        break
ctx = _lib.get_ctx()
lib = _lib.load_library()
os.environ["TGP_DEBUG_SEGS"] = "1"
buf = ops.DeviceBuffer(ctx, lib.tgp_panel_elems(Np) * 8)
host = (1e-3 * np.random.default_rng(0).standard_normal(min(lib.tgp_panel_elems(Np), 1 << 26)))
off, ne = 0, lib.tgp_panel_elems(Np)
while off < ne:
    cnt = min(len(host), ne - off)
    _lib.check(ctx, lib.tgp_h2d(ctx, C.c_void_p(buf.ptr.value + off * 8), host.ctypes.data_as(C.c_void_p), cnt * 8), "h2d")
    off += cnt
ms, fl = C.c_double(), C.c_double()
_lib.check(ctx, lib.tgp_debug_syrk_loop(ctx, buf.ptr, Np, reps, C.byref(ms), C.byref(fl)), "syrk_loop")
print("syrk_segs_kernel<4> (single-GPU): Np=%d  %.3f ms/launch  %.2f TFLOP/s" % (Np, ms.value, fl.value / ms.value / 1e9))

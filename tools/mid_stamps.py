#!/usr/bin/env python3
"""In-kernel stamps of ONE panel_mid_kernel launch (stamps build): when its first 25 workgroups -- eight slices L10 = A10 W0^T,
sixteen half slices A11 -= L10 L10^T, the second diagonal block -- started, computed, published and were released, and the diagonal
block before it: where the ~19 us between a panel's two diagonal blocks go (launch, two hand-offs, loads).
usage: TGP_LIB_PATH=.../libtgp_stamps.so python tools/mid_stamps.py [N=8192] [panel=20 ...]"""
import ctypes as C, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from treegp_amd import _lib, ops  # noqa: E402
from treegp_amd.synthetic import star_field, headline_invlam  # noqa: E402
n = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
panels = [int(v) for v in sys.argv[2:]] or [20]
_lib.load_library(); ctx = _lib.get_ctx()
iL = headline_invlam(); spec = ops.KernelSpec(_lib.TGP_ARBF, amp=1.0, a=iL[0, 0], b=iL[0, 1], c=iL[1, 1])
X, y, ye, _ = star_field(n, 16)
raw = C.CDLL(_lib.LIB_PATH)
for k in panels:
    raw.tgp_debug_mid_stamps(None, 256 * k)
    for it in range(3):
        ops.gp_solve(spec, X, y - y.mean(), ye, want_alpha=False)
    m = (C.c_ulonglong * (25 * 8))(); raw.tgp_debug_mid_stamps(m, -1)
    p = (C.c_ulonglong * (1024 * 20))(); raw.tgp_debug_potrf_stamps(p)
    ms = np.array(m[:], dtype=np.float64).reshape(25, 8); ps = np.array(p[:], dtype=np.float64).reshape(1024, 20)
    t0 = ps[2 * k, 16]                      # end of the panel's first diagonal block
    f = lambda v: (v - t0) / 100.0
    print("panel %d of N = %d: times in us after the end of its first diagonal block (potrf128 of block %d; the second one starts its body at %.1f, ends %.1f)"
          % (k, n, 2 * k, f(ps[2 * k + 1, 0]), f(ps[2 * k + 1, 16])))
    for b in range(8):
        print("  L10 slice %d:   entry %6.1f  computed+stored %6.1f  published %6.1f" % (b, f(ms[b, 0]), f(ms[b, 1]), f(ms[b, 2])))
    for b in range(8, 24):
        print("  A11 slice %d.%d: entry %6.1f  released %6.1f  computed+stored %6.1f  published %6.1f" % ((b - 8) >> 1, b & 1, f(ms[b, 0]), f(ms[b, 1]), f(ms[b, 2]), f(ms[b, 3])))
    print("  diagonal block: entry %6.1f  released %6.1f  done %6.1f" % (f(ms[24, 0]), f(ms[24, 1]), f(ms[24, 2])))

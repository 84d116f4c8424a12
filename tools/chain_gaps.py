#!/usr/bin/env python3
"""The panel chain's own timeline of ONE factorisation, without a profiler: in-kernel stamps of the diagonal blocks (library
built with tools/build_stamps.sh).  Per pair of panels: when its first diagonal block started, how long the chain took to the
end of its fourth, and the gap to the next pair's first block -- the rows' last step, the hand-overs and U2a, plus whatever
the chain waited for the bulk's stream.  usage: TGP_LIB_PATH=.../libtgp_stamps.so python tools/chain_gaps.py [N=8192]"""
import ctypes as C, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from treegp_amd import _lib, ops  # noqa: E402
from treegp_amd.synthetic import star_field, headline_invlam  # noqa: E402
n = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
_lib.load_library(); ctx = _lib.get_ctx()
iL = headline_invlam(); spec = ops.KernelSpec(_lib.TGP_ARBF, amp=1.0, a=iL[0, 0], b=iL[0, 1], c=iL[1, 1])
X, y, ye, _ = star_field(n, 16)
raw = C.CDLL(_lib.LIB_PATH)
buf = (C.c_ulonglong * (1024 * 20))()
rows = []
for it in range(5):
    ops.gp_solve(spec, X, y - y.mean(), ye, want_alpha=False)
    chol = _lib.timings(ctx)[1]
    raw.tgp_debug_potrf_stamps(buf)
    s = np.array(buf[:], dtype=np.float64).reshape(1024, 20)
    nblk = (n + 127) // 128
    st, en = (s[:nblk, 0] - s[0, 0]) / 100.0, (s[:nblk, 16] - s[0, 0]) / 100.0
    rows.append((chol, st, en))
chol, st, en = sorted(rows, key=lambda r: r[0])[0]
print("N = %d, Cholesky %.3f ms (best of 5); pair: first block start | chain to the end of the 4th block | gap to the next pair | cycle" % (n, chol))
tot_gap = 0.0
for p in range(0, nblk // 4):
    b = 4 * p
    nxt = st[b + 4] if b + 4 < nblk else float("nan")
    # (T3: tile rows of the bulk update that runs beside this pair's chain -- the one after the pair before it)
    print("  pair %2d (T3 %3d)  %8.1f  %7.1f  %7.1f  %7.1f" % (p, max(n // 128 - 4 * p - 4, 0) if p else 0, st[b], en[b + 3] - st[b], nxt - en[b + 3], nxt - st[b]))

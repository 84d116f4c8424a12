"""In-kernel time stamps of potrf128 (library built with -DTGP_POTRF_STAMPS, see tools/build_stamps.sh): per phase, for
every diagonal block of ONE factorisation of order N -- i.e. in situ, next to the bulk update.  s_memrealtime ticks
(100 MHz, one clock for the whole chip; s_memtime is per XCD / not comparable across compute units).
usage: TGP_LIB_PATH=.../libtgp_stamps.so python tools/potrf_stamps.py [N=128] [first block] [blocks]"""
import ctypes as C
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from treegp_amd import _lib, ops  # noqa: E402
from treegp_amd.synthetic import star_field, headline_invlam  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 128
first = int(sys.argv[2]) if len(sys.argv) > 2 else 0
count = int(sys.argv[3]) if len(sys.argv) > 3 else 16
lib = _lib.load_library()
ctx = _lib.get_ctx()
iL = headline_invlam()
spec = ops.KernelSpec(_lib.TGP_ARBF, amp=1.0, a=iL[0, 0], b=iL[0, 1], c=iL[1, 1])
X, y, ye, Xs = star_field(n, 16)
for it in range(3):
    ops.gp_solve(spec, X, y - y.mean(), ye, want_alpha=False)
buf = (C.c_ulonglong * (1024 * 20))()
raw = C.CDLL(_lib.LIB_PATH)
print("rc", raw.tgp_debug_potrf_stamps(buf))
s = np.array(buf[:], dtype=np.float64).reshape(1024, 20)
names = ["load"] + sum([["diag%d" % j, "strip%d" % j, "schur%d" % j] for j in range(4)], []) + ["storeL", "phase2", "storeW"]
print("block   start(us) total(us) | " + " ".join("%6s" % nm for nm in names) + "   (phases in 10 ns ticks)")
t0 = s[0, 0]
nblk = (n + 127) // 128
for b in range(first, min(first + count, nblk)):
    d = np.diff(s[b, :17])
    where = int(s[b, 17])
    print("%5d  %9.1f %8.1f  | " % (b, (s[b, 0] - t0) / 100.0, (s[b, 16] - s[b, 0]) / 100.0) + " ".join("%6.0f" % v for v in d) +
          "   xcc %d se %d cu %d" % (where >> 8, (where >> 5) & 7, where & 15))

if hasattr(raw, "tgp_debug_potrf_fine"):
    fine = (C.c_ulonglong * (1024 * 8))()
    raw.tgp_debug_potrf_fine(fine)
    f = np.array(fine[:], dtype=np.float64).reshape(1024, 8)
    print("inside the diagonal step of 32-column block 1, shader-clock ticks: sweep 1 | L21 = A21 D11^T + stores | A22 -= L21 L21^T | sweep 2 | S = L21 D11 | D21 = -D22 S")
    for b in range(first, min(first + 4, nblk)):
        print("%5d   " % b + " ".join("%6.0f" % v for v in np.diff(f[b, :7])))

# one panel GEMM of the chain (gemm_col_kernel<0> with TGP_STAMP_GRID workgroups): when and where each workgroup ran
grid = int(os.environ.get("TGP_STAMP_GRID", "0"))
if grid:
    raw.tgp_debug_gemm_stamps(None, grid)
    ops.gp_solve(spec, X, y - y.mean(), ye, want_alpha=False)
    raw.tgp_debug_potrf_stamps(buf)
    s = np.array(buf[:], dtype=np.float64).reshape(1024, 20)
    g = (C.c_ulonglong * (1024 * 4))()
    raw.tgp_debug_gemm_stamps(g, 0)
    gs = np.array(g[:], dtype=np.uint64).reshape(1024, 4)[:grid]
    # the diagonal block that precedes this GEMM: the one whose end is the latest before the first workgroup start
    ends = s[:nblk, 16]
    first = float(gs[:, 0].min())
    prev = int(np.argmax(np.where(ends <= first, ends, -np.inf)))
    print("panel GEMM with %d workgroups; preceding diagonal block %d ended at tick 0" % (grid, prev))
    for i in range(grid):
        print("  wg %3d  start %8.1f  end %8.1f  (us after that block)  xcc %d  se %d cu %d" %
              (i, (float(gs[i, 0]) - ends[prev]) / 100.0, (float(gs[i, 1]) - ends[prev]) / 100.0, int(gs[i, 2]) >> 8, (int(gs[i, 2]) >> 5) & 7, int(gs[i, 2]) & 15))

# the queued bulk update with T == TGP_STAMP_T tile rows: which workgroups left (reserved CUs), where the others sat
T = int(os.environ.get("TGP_STAMP_T", "0"))
if T:
    raw.tgp_debug_queue_stamps(None, T)
    ops.gp_solve(spec, X, y - y.mean(), ye, want_alpha=False)
    q = (C.c_ulonglong * (1024 * 4))()
    raw.tgp_debug_queue_stamps(q, 0)
    qs = np.array(q[:], dtype=np.uint64).reshape(1024, 4)
    used = qs[:, 0] != 0
    qs = qs[used]
    t0 = float(qs[:, 0].min())
    print("queued bulk update, T = %d: %d workgroups reported" % (T, len(qs)))
    for xcc in range(8):
        sel = qs[((qs[:, 2] >> np.uint64(8)) & np.uint64(15)) == xcc]
        left = sel[(sel[:, 2] >> np.uint64(32)) != 0]
        stay = sel[(sel[:, 2] >> np.uint64(32)) == 0]
        cus = {}
        for r in stay:
            cus.setdefault(int(r[2]) & 0xff, []).append(int(r[3]))
        lcus = {}
        for r in left:
            k = int(r[2]) & 0xff
            lcus[k] = lcus.get(k, 0) + 1
        per = {}
        for k, v in cus.items():
            per[len(v)] = per.get(len(v), 0) + 1
        if len(stay):
            st0, en = (stay[:, 0].astype(np.float64) - t0) / 100.0, (stay[:, 1].astype(np.float64) - t0) / 100.0
            tl = stay[:, 3].astype(np.float64)
            print("  xcc %d: staying workgroups start %.1f .. %.1f us, end %.1f .. %.1f us (mean %.1f), %d tiles, us per tile %.1f"
                  % (xcc, st0.min(), st0.max(), en.min(), en.max(), en.mean(), int(tl.sum()), float(((en - st0).sum()) / max(tl.sum(), 1.0))))
        print("  xcc %d: %3d workgroups, %2d left from CUs %s (arrival %.1f .. %.1f us); staying: CUs x workgroups %s, tiles per workgroup %s"
              % (xcc, len(sel), len(left), {("se%d cu%d" % ((k >> 5) & 7, k & 15)): c for k, c in lcus.items()},
                 (float(left[:, 0].min()) - t0) / 100.0 if len(left) else 0.0, (float(left[:, 0].max()) - t0) / 100.0 if len(left) else 0.0,
                 per, sorted(set(int(t) for t in stay[:, 3]))))

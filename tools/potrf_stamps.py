import sys, ctypes as C, numpy as np
sys.path.insert(0, "/root/repo" if len(sys.argv) < 2 else sys.argv[1])
from treegp_amd import _lib, ops
from treegp_amd.synthetic import star_field, headline_invlam
lib = _lib.load_library(); ctx = _lib.get_ctx()
iL = headline_invlam(); spec = ops.KernelSpec(_lib.TGP_ARBF, amp=1.0, a=iL[0,0], b=iL[0,1], c=iL[1,1])
X, y, ye, Xs = star_field(128, 16)
for it in range(3): ops.gp_solve(spec, X, y - y.mean(), ye)
buf = (C.c_ulonglong * 32)()
raw = C.CDLL(_lib.LIB_PATH)
print("rc", raw.tgp_debug_potrf_stamps(buf))
s = np.array(buf[:17], dtype=np.float64)
names = ["load"] + sum([["diag32[%d]" % j, "Xstrip[%d]" % j, "schur[%d]" % j] for j in range(4)], []) + ["storeL", "phase2", "storeW"]
d = np.diff(s)
for n, v in zip(names, d): print("%-10s %8.0f ticks" % (n, v))
print("total %d ticks (s_memtime: 100 MHz? -> check)" % (s[16]-s[0]))

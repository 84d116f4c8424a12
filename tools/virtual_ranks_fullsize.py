import sys, os, threading, time, numpy as np, torch
root = sys.argv[1]
sys.path.insert(0, root); sys.path.insert(0, os.path.join(root, "tests"))
from treegp_amd import _lib, ops
from treegp_amd.dist import DistributedGP
from _dist_helpers import ThreadComm
from treegp_amd.synthetic import star_field, headline_invlam
G, n, m = int(sys.argv[2]), int(sys.argv[3]), 4096
iL = headline_invlam(); spec = ops.KernelSpec(_lib.TGP_ARBF, amp=1.0, a=iL[0,0], b=iL[0,1], c=iL[1,1])
X, y, ye, Xs = star_field(n, m); y = y - y.mean()
t0 = time.perf_counter(); a_ref, ld_ref, _, _ = ops.gp_solve(spec, X, y, ye); yp_ref = ops.gp_predict(spec, X, a_ref, Xs)
print("single GPU: %.2f s" % (time.perf_counter() - t0), flush=True)
shared = ThreadComm.Shared(G); res, errs = [None] * G, []
dev = torch.device("cuda", 0)
def run(rank):
    try:
        gp = DistributedGP(_lib.new_ctx(0), spec, X, y, ye, Xs, comm=ThreadComm(shared, rank), device=dev)
        alpha, _ = gp.step(); full = gp.gather_predictions(); torch.cuda.synchronize()
        res[rank] = (alpha.cpu().numpy()[:n], float(gp.logdet[0]), full.cpu().numpy())
    except BaseException as e:
        errs.append(e)
        try: shared.barrier.abort()
        except Exception: pass
t0 = time.perf_counter()
th = [threading.Thread(target=run, args=(r,)) for r in range(G)]
[t.start() for t in th]; [t.join(timeout=900) for t in th]
print("%d virtual ranks: %.2f s, errors: %s" % (G, time.perf_counter() - t0, errs), flush=True)
for r in range(G):
    a, ld, yp = res[r]
    print("rank %d: max|alpha-ref|/max|ref| = %.2e, logdet rel %.2e, max|pred-ref|/max = %.2e" % (r, np.abs(a - a_ref).max() / np.abs(a_ref).max(), abs(ld - ld_ref) / abs(ld_ref), np.abs(yp - yp_ref).max() / np.abs(yp_ref).max()), flush=True)

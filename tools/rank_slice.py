#!/usr/bin/env python3
"""ONE rank's share of a G-rank factorisation, alone on this GPU: the driver of treegp_amd/dist.py with a communicator that
pretends to have G ranks and moves nothing (gathered panels hold whatever the buffers held: the arithmetic is meaningless,
its cost is not -- fp64 MFMA time does not depend on the data).  Gives the compute side of the multi-GPU forecast: the rank's
bulk-update time, its panel-chain time without any communication, and the stall of the main stream behind the chain.
usage: rank_slice.py [N=65536] [G=8] [rank=0]"""
import sys
import time

import numpy as np
import torch

sys.path.insert(0, __file__.rsplit("/tools/", 1)[0])
from treegp_amd import _lib, ops  # noqa: E402
from treegp_amd.dist import DistributedCholesky, HipLocalOps, _Done  # noqa: E402
from treegp_amd.synthetic import star_field, headline_invlam  # noqa: E402


class SliceComm(object):
    """G ranks on paper, one in fact: every collective returns at once and leaves the buffers as they are."""

    def __init__(self, size, rank):
        self.size, self.rank = size, rank

    def broadcast(self, t, src):
        pass

    def all_reduce_sum(self, t):
        pass

    def all_reduce_max(self, t):
        pass

    def all_gather(self, out, inp):
        pass

    def all_gather_start(self, out, inp):
        return _Done(out)


n = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
G = int(sys.argv[2]) if len(sys.argv) > 2 else 8
g = int(sys.argv[3]) if len(sys.argv) > 3 else 0
dev = torch.device("cuda", 0)
iL = headline_invlam()
spec = ops.KernelSpec(_lib.TGP_ARBF, amp=1.0, a=iL[0, 0], b=iL[0, 1], c=iL[1, 1])
X, y, ye, _ = star_field(n, 16)
o = HipLocalOps(_lib.new_ctx(0), spec, n, G, g, dev, replicate=False)
comm = SliceComm(G, g)
timer = lambda: torch.cuda.Event(enable_timing=True)      # noqa: E731
chol = DistributedCholesky(o, comm, timer=timer)
for buf in chol.gathered:
    buf.normal_(0.0, 1e-4)                                 # finite numbers in the panels "received" from the other ranks
dX, de = o.to_device(_lib.as_xy(X)), o.to_device(ye)
res = []
for it in range(3):
    o.kbuild(dX, de)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    chol.factorize()                                       # (reports a failed pivot: the panels are not the real ones)
    torch.cuda.synchronize()
    res.append(((time.perf_counter() - t0) * 1e3, chol.update_ms, chol.chain_ms, chol.wait_ms, chol.update_flops))
wall, bulk, chain, wait, fl = min(res[1:])
print("N=%d, rank %d of %d alone on one MI355X (no communication): factorisation %.1f ms wall; bulk update %.1f ms (%.1f TF on "
      "the rank's share); panel chain %.1f ms on the side stream; main stream stalled behind the chain %.1f ms; group size %d"
      % (n, g, G, wall, bulk, fl / bulk / 1e9, chain, wait, chol.group), flush=True)

if len(sys.argv) > 4 and sys.argv[4] == "launches":
    # per-launch view of the bulk updates (main stream): tiles, time, rate
    rec = []
    orig = o.update_group

    def wrapped(k, bufs, cmaxs, col_lo=0, col_hi=-1, side=False, queue_nres=0):
        if side:
            return orig(k, bufs, cmaxs, col_lo, col_hi, side=side, queue_nres=queue_nres)
        e0, e1 = timer(), timer()
        e0.record()
        orig(k, bufs, cmaxs, col_lo, col_hi, side=side, queue_nres=queue_nres)
        e1.record()
        rec.append((k, col_lo, col_hi, e0, e1))

    o.update_group = wrapped
    orig_fused = o.update_group_fused

    def wrapped_fused(k, bufs, cmaxs, head_cols, queue_nres=0):
        e0, e1 = timer(), timer()
        e0.record()
        orig_fused(k, bufs, cmaxs, head_cols, queue_nres=queue_nres)
        e1.record()
        rec.append((k, 0, -1, e0, e1))

    o.update_group_fused = wrapped_fused
    o.kbuild(dX, de)
    chol.factorize()
    torch.cuda.synchronize()
    GS, nB = chol.group, o.nB
    from treegp_amd.dist import block_of, first_round
    print("   k  cols        tiles   ms      TF    rounds(512 slots)")
    for k, lo, hi, e0, e1 in rec:
        s0 = k + GS
        tiles = 0
        q = first_round(s0, g, G)
        while block_of(q, g, G) < nB:
            b = block_of(q, g, G)
            for half in (0, 1):
                gti = 2 * (b - s0) + half                       # last valid global tile column of this local tile row
                top = gti if hi < 0 else min(gti, hi - 1)
                tiles += max(0, top - lo + 1)
            q += 1
        ms = e0.elapsed_time(e1)
        if tiles:
            print("%4d  %3d..%-4s %7d  %7.3f  %5.1f  %6.1f" % (k, lo, "end" if hi < 0 else hi, tiles, ms, tiles * 2.0 * 128 * 128 * 256 * GS / ms / 1e9, tiles / 512.0))

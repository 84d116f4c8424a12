#!/usr/bin/env python3
"""What tools/rank_slice.py leaves out: ONE rank's share of a G-rank factorisation alone on this GPU, as there, but with a
communicator that issues the REAL torch.distributed calls on RCCL -- a process group of one rank -- at the message sizes and
counts of rank g of G: per panel a 768 KB broadcast and an all_gather_into_tensor of the rank's cmax-block send view (what a
rank contributes; a world of one receives nothing), plus the probe-free all-reduces.  Nothing crosses a link, so this is the
per-collective FLOOR of the host path (Python + ProcessGroupNCCL + the stream hand-over to RCCL's stream and back) on the
panel chain, next to the same slice with a communicator that returns at once.  Recorded per collective: host time to enqueue,
device time between the events around it on the side stream, and whether the side stream was already idle when the host got
there (the host, not the device, was the bottleneck at that point).
usage: rank_slice_rccl.py [N=65536] [G=8] [rank=0]"""
import os
import sys
import time

import numpy as np
import torch
import torch.distributed as dist

sys.path.insert(0, __file__.rsplit("/tools/", 1)[0])
from treegp_amd import _lib, ops  # noqa: E402
from treegp_amd.dist import DistributedCholesky, HipLocalOps, _Done  # noqa: E402
from treegp_amd.synthetic import star_field, headline_invlam  # noqa: E402


class SliceComm(object):
    """G ranks on paper, one in fact: every collective returns at once (tools/rank_slice.py)."""

    def __init__(self, size, rank):
        self.size, self.rank = size, rank

    def broadcast(self, t, src):
        pass

    def all_reduce_sum(self, t):
        pass

    def all_reduce_max(self, t):
        pass

    def all_gather_start(self, out, inp):
        return _Done(out)


class _TimedWork(object):
    def __init__(self, work, tensor, rec):
        self.work, self.tensor, self.rec, self.first = work, tensor, rec, True

    def wait(self):
        self.work.wait()                                   # orders the current stream behind RCCL's
        if self.first:                                     # the chain's own wait comes first (dist.py: _side_group)
            self.first = False
            self.rec[2].record(torch.cuda.current_stream())


class RcclSliceComm(SliceComm):
    """The same G ranks on paper, every call a real one on a one-rank RCCL group."""

    def __init__(self, size, rank):
        SliceComm.__init__(self, size, rank)
        self.calls = {"broadcast": [], "all_gather": [], "all_reduce": []}

    def _rec(self, kind):
        st = torch.cuda.current_stream()
        idle = st.query()                                  # nothing queued ahead of this collective: the device waits for the host
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        rec = [0.0, e0, e1, idle]
        self.calls[kind].append(rec)
        e0.record(st)
        return rec, st

    def broadcast(self, t, src):
        rec, st = self._rec("broadcast")
        t0 = time.perf_counter()
        dist.broadcast(t, src=0)
        rec[0] = (time.perf_counter() - t0) * 1e6
        rec[2].record(st)

    def all_reduce_sum(self, t):
        rec, st = self._rec("all_reduce")
        t0 = time.perf_counter()
        dist.all_reduce(t)
        rec[0] = (time.perf_counter() - t0) * 1e6
        rec[2].record(st)

    def all_reduce_max(self, t):
        rec, st = self._rec("all_reduce")
        t0 = time.perf_counter()
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        rec[0] = (time.perf_counter() - t0) * 1e6
        rec[2].record(st)

    def all_gather_start(self, out, inp):
        rec, st = self._rec("all_gather")
        n = inp.numel()
        t0 = time.perf_counter()
        w = dist.all_gather_into_tensor(out[self.rank * n:(self.rank + 1) * n], inp, async_op=True)
        rec[0] = (time.perf_counter() - t0) * 1e6
        return _TimedWork(w, out, rec)


def run(chol, o, dX, de, reps=3):
    res = []
    for it in range(reps):
        o.kbuild(dX, de)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        chol.factorize()
        torch.cuda.synchronize()
        res.append(((time.perf_counter() - t0) * 1e3, chol.update_ms, chol.chain_ms, chol.wait_ms))
    return min(res[1:])


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
    G = int(sys.argv[2]) if len(sys.argv) > 2 else 8
    g = int(sys.argv[3]) if len(sys.argv) > 3 else 0
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29541")
    dist.init_process_group("nccl", rank=0, world_size=1)
    dev = torch.device("cuda", 0)
    iL = headline_invlam()
    spec = ops.KernelSpec(_lib.TGP_ARBF, amp=1.0, a=iL[0, 0], b=iL[0, 1], c=iL[1, 1])
    X, y, ye, _ = star_field(n, 16)
    o = HipLocalOps(_lib.new_ctx(0), spec, n, G, g, dev, replicate=False)
    timer = lambda: torch.cuda.Event(enable_timing=True)      # noqa: E731
    dX, de = o.to_device(_lib.as_xy(X)), o.to_device(ye)
    out = {}
    for name, comm in (("no communication", SliceComm(G, g)), ("RCCL, one rank", RcclSliceComm(G, g))):
        chol = DistributedCholesky(o, comm, timer=timer)
        for buf in chol.gathered:
            buf.normal_(0.0, 1e-4)
        if isinstance(comm, RcclSliceComm):
            for v in comm.calls.values():
                del v[:]
        wall, bulk, chain, wait = run(chol, o, dX, de)
        out[name] = wall
        print("N=%d rank %d of %d, %s: factorisation %.1f ms wall; bulk %.1f ms; panel chain %.1f ms on the side stream; main stream "
              "stalled behind the chain %.1f ms" % (n, g, G, name, wall, bulk, chain, wait), flush=True)
        if isinstance(comm, RcclSliceComm):
            torch.cuda.synchronize()
            for kind, recs in comm.calls.items():
                recs = recs[len(recs) * 2 // 3:]                       # the last of the three factorisations
                if not recs:
                    continue
                host = np.array([r[0] for r in recs])
                devt = np.array([1e3 * r[1].elapsed_time(r[2]) for r in recs])
                idle = np.mean([1.0 if r[3] else 0.0 for r in recs])
                print("   %-10s x %3d per factorisation: host enqueue median %.1f us (p90 %.1f, sum %.2f ms); device span median %.1f us "
                      "(p90 %.1f, sum %.2f ms); side stream already idle at %.0f %% of the calls"
                      % (kind, len(recs), np.median(host), np.percentile(host, 90), host.sum() / 1e3, np.median(devt),
                         np.percentile(devt, 90), devt.sum() / 1e3, 100.0 * idle), flush=True)
    a, b = out["no communication"], out["RCCL, one rank"]
    print("   the real calls cost %.1f ms of %.1f (%.1f %%) before a byte crosses a link" % (b - a, a, 100.0 * (b - a) / a))
    dist.destroy_process_group()


if __name__ == "__main__":
    main()

"""Multi-GPU GP solve: row-block-cyclic distributed Cholesky + triangular solves + sharded predict.

One process per GPU (``torch.distributed``, backend ``nccl`` = RCCL over xGMI).  The reference has
no distributed code; this is the N-scaling axis of SURVEY.md 8(e) for the same seams as
``tgp_gp_solve`` / ``tgp_gp_predict`` (treegp/gp_interp.py:177-183).

Layout: the N x N kernel matrix is cut into 256-row blocks, dealt to the G ranks block-cyclically with
the deal REFLECTED every G blocks (``owner``: ranks 0..G-1, then G-1..0, ...): block row b of the lower
triangle carries b + 1 block columns, and the reflection evens out what the plain deal b % G piles on
the last rank (max / mean work 1.041 -> 1.002 at 256 blocks on 8 ranks).  A rank stores the lower-
triangular part of its block rows as packed 256-wide panels like the single-GPU factor.  Per panel k:

    owner(k)        factor the 256x256 diagonal block            tgp_dd_factor_diag
    broadcast       [L_kk | W0 | W1] = 768 KB                    dist.broadcast
    every rank      solve its rows of panel k (GEMMs with W)     tgp_dd_trsm
    all-gather      the panel, ((N - 256 k)/G) x 256 per rank    dist.all_gather_into_tensor
    every rank      update its own block rows on fp64 MFMA       tgp_dd_update

K build needs no communication (``tgp_dd_kbuild``).  The triangular solves move one 2 KB
broadcast (forward) / one 2 KB all-reduce (backward) per block.  Prediction points are sharded,
``alpha`` and the coordinates are replicated.

The orchestration below is independent of where the local arithmetic runs: ``HipLocalOps`` is the
product implementation (device pointers into libtgp.so); tests substitute a NumPy stand-in to
exercise the communication logic under ``gloo`` on CPUs.
"""
import ctypes as C
import os
import threading

import numpy as np

BLK = 256                      # block-row height = panel width
BCAST_ELEMS = BLK * BLK + 2 * 128 * 128


def owner(b, G):
    """rank that owns block row b: round q = b // G deals ranks 0..G-1 in even rounds, G-1..0 in odd ones
    (csrc/tgp_internal.h: dist_owner)"""
    q, p = divmod(b, G)
    return G - 1 - p if q & 1 else p


def block_of(q, r, G):
    """rank r's block of round q (= its q-th local block)"""
    return q * G + (G - 1 - r if q & 1 else r)


def first_round(s, r, G):
    """round (= local index) of the smallest block >= s owned by rank r"""
    q = s // G
    return q if block_of(q, r, G) >= s else q + 1


def first_ge(s, r, G):
    """smallest block index >= s owned by rank r"""
    return block_of(first_round(s, r, G), r, G)


def panel_blocks(p, nB, g, G):
    """number of blocks p <= b < nB owned by rank g"""
    if nB <= 0:
        return 0
    ql = (nB - 1) // G
    if block_of(ql, g, G) >= nB:
        ql -= 1
    return max(0, ql - first_round(p, g, G) + 1)


FINISH_BLOCKS_MAX = 64          # upper limit of TGP_DIST_FINISH (blocks of 256 rows factored redundantly at the end)


def panel_cmax(p, nB, G):
    """most blocks p <= b < nB any rank holds (the per-rank slot count of an all-gathered panel).  With the reflected deal
    a window shorter than G can hold two blocks of one rank (G-1 and G are both rank G-1's), so this is not ceil((nB-p)/G)."""
    return max(panel_blocks(p, nB, r, G) for r in range(G)) if nB > p else 0


def gathered_index(b, first, G):
    """(rank, index) of block b inside an all-gathered panel [rank][cmax][256][256] that holds the blocks >= first"""
    r = owner(b, G)
    return r, b // G - first_round(first, r, G)


class DistStall(RuntimeError):
    """A rank saw no progress for TGP_DIST_WATCHDOG_S seconds: a collective whose partner never arrived, or device streams
    that stopped advancing.  Raised on every rank that observes it (all do: every panel's collectives involve every rank);
    the process group is unusable afterwards -- tear the processes down and start fresh ones."""


def _is_stall(e):
    """a backend error that means "my partner is late or gone" (time-out, or the connection of a partner that gave up first)"""
    m = str(e).lower()
    return any(w in m for w in ("timed out", "timeout", "connection closed", "connection reset", "broken pipe"))


def watchdog_seconds():
    """TGP_DIST_WATCHDOG_S: seconds without progress after which a rank gives up (default 300; 0 = wait for ever)."""
    try:
        return max(float(os.environ.get("TGP_DIST_WATCHDOG_S", "300")), 0.0)
    except ValueError:
        return 300.0


class TorchComm(object):
    """Collectives over torch.distributed (nccl on GPUs; gloo for the CPU tests)."""

    def __init__(self, group=None):
        import torch.distributed as dist
        self.dist = dist
        self.group = group
        self.rank = dist.get_rank(group)
        self.size = dist.get_world_size(group)
        self.native_gather = dist.get_backend(group) == "nccl"
        # The panel exchange: all_gather_into_tensor, or ("p2p") one send + one receive per peer.  xGMI is point-to-point: a
        # ring all-gather moves (G-1)/G of every panel through ONE link per rank, direct sends use all G-1 links at once;
        # which of the two RCCL's all-gather is on a given node is measured, not assumed: probe_gather() (called by
        # dist.enable() on worlds of more than one rank) times both on a representative panel and rank 0's choice is
        # broadcast.  TGP_DIST_GATHER=allgather|p2p overrides the probe.
        self.gather_mode = os.environ.get("TGP_DIST_GATHER", "allgather")
        self.gather_probe = None             # what probe_gather() measured and chose (goes into the bench line)
        self.bytes_in = 0                    # payload bytes this rank has received through the collectives below
        self.device = None                   # where host-side reductions have to be staged (RCCL: on the GPU)
        if self.native_gather:
            import torch
            self.device = torch.device("cuda", torch.cuda.current_device())

    def probe_gather(self, elems=None, reps=3):
        """Time the two panel exchanges on one representative panel (default 2 Mi doubles = 16 MiB per rank, the size of
        a rank's share of a panel at N = 65 536 on 8 ranks), `reps` times each after one warm-up; rank 0 decides
        (the faster by its own clock), the decision is broadcast so that every rank issues the same calls.
        Collective.  Returns and stores the record; TGP_DIST_GATHER set: no timing, the override is recorded."""
        import time
        import torch
        env = os.environ.get("TGP_DIST_GATHER")
        if self.size == 1:
            self.gather_probe = {"skipped": "world of one", "chosen": self.gather_mode}
            return self.gather_probe
        dev = self.device or torch.device("cpu")
        code = torch.zeros(1, dtype=torch.float64, device=dev)
        if self.rank == 0:
            code[0] = {"allgather": 1.0, "p2p": 2.0}.get(env, 0.0)      # rank 0's environment decides whether to probe at all
        self.dist.broadcast(code, src=0, group=self.group)
        if int(code.item()) != 0:
            self.gather_mode = "allgather" if int(code.item()) == 1 else "p2p"
            self.gather_probe = {"skipped": "TGP_DIST_GATHER", "chosen": self.gather_mode}
            return self.gather_probe
        n = int(elems) if elems else (1 << 21)
        inp = torch.full((n,), float(self.rank), dtype=torch.float64, device=dev)
        out = torch.empty(n * self.size, dtype=torch.float64, device=dev)
        sync = torch.cuda.synchronize if self.native_gather else (lambda: None)
        res = {}
        keep_bytes = self.bytes_in
        for mode in ("allgather", "p2p"):
            self.gather_mode = mode
            ts = []
            for it in range(reps + 1):
                self.dist.barrier(group=self.group)
                sync()
                t0 = time.perf_counter()
                self.all_gather_start(out, inp).wait()
                sync()
                ts.append(time.perf_counter() - t0)
            ok = all(float(out[r * n]) == float(r) and float(out[(r + 1) * n - 1]) == float(r) for r in range(self.size))
            if not ok:
                raise RuntimeError("panel exchange %r delivered wrong data in the probe" % mode)
            res[mode] = 8.0 * n * (self.size - 1) / min(ts[1:]) / 1e9       # payload received per rank and second
        self.bytes_in = keep_bytes
        t = torch.tensor([res["allgather"], res["p2p"]], dtype=torch.float64, device=dev)
        self.dist.broadcast(t, src=0, group=self.group)                 # rank 0's clock decides for everybody
        ag, pp = float(t[0]), float(t[1])
        self.gather_mode = "p2p" if pp > 1.05 * ag else "allgather"     # the library collective unless direct sends clearly win
        self.gather_probe = {"allgather_GBps": ag, "p2p_GBps": pp, "chosen": self.gather_mode, "payload_MB_per_rank": 8.0 * n / 1e6,
                             "reps": reps, "this_rank": {"allgather_GBps": res["allgather"], "p2p_GBps": res["p2p"]}}
        return self.gather_probe

    def _finish(self, work, what):
        """A collective of a host-blocking backend (gloo) under the watchdog: wait with a time limit and turn the backend's
        time-out into DistStall.  RCCL's wait() only orders streams -- a stall there shows where the host synchronises
        (DistributedCholesky._await_device)."""
        limit = watchdog_seconds()
        if self.native_gather or limit <= 0.0:
            work.wait()
            return
        import datetime
        try:
            work.wait(datetime.timedelta(seconds=limit))
        except RuntimeError as e:              # ProcessGroupGloo: "Timed out waiting ... for recv/send operation to complete",
            if _is_stall(e):                   # or, when the partner has given up already, "Connection closed by peer"
                raise DistStall("rank %d of %d: %s did not complete (limit %.1f s, TGP_DIST_WATCHDOG_S): %s"
                                % (self.rank, self.size, what, limit, str(e).splitlines()[0]))
            raise

    def broadcast(self, t, src):
        if self.native_gather:
            self.dist.broadcast(t, src=src, group=self.group)
        else:
            self._finish(self.dist.broadcast(t, src=src, group=self.group, async_op=True), "a broadcast from rank %d" % src)
        if src != self.rank:
            self.bytes_in += t.numel() * t.element_size()

    def all_reduce_sum(self, t):
        if self.native_gather:
            self.dist.all_reduce(t, group=self.group)
        else:
            self._finish(self.dist.all_reduce(t, group=self.group, async_op=True), "an all-reduce")

    def all_reduce_max(self, t):
        if self.native_gather:
            self.dist.all_reduce(t, op=self.dist.ReduceOp.MAX, group=self.group)
        else:
            self._finish(self.dist.all_reduce(t, op=self.dist.ReduceOp.MAX, group=self.group, async_op=True), "an all-reduce")

    def all_gather_start(self, out, inp):
        """Non-blocking all-gather where the backend has one (RCCL runs it on its own stream, so it
        overlaps the kernels queued behind it); returns an object with .wait()."""
        if self.gather_mode == "p2p" and self.size > 1 and (self.native_gather or not inp.is_cuda):
            return self._p2p_gather(out, inp)
        if self.native_gather:
            self.bytes_in += (self.size - 1) * inp.numel() * inp.element_size()
            return _Work(self.dist.all_gather_into_tensor(out, inp, group=self.group, async_op=True), out)
        self.all_gather(out, inp)
        return _Done(out)

    def _p2p_gather(self, out, inp):
        """every rank's `inp` to every other rank, directly: out[r n : (r+1) n] <- rank r's inp"""
        dist, n = self.dist, inp.numel()
        ops = []
        for step in range(1, self.size):
            dst, src = (self.rank + step) % self.size, (self.rank - step) % self.size
            ops.append(dist.P2POp(dist.isend, inp, dst, group=self.group))
            ops.append(dist.P2POp(dist.irecv, out[src * n:(src + 1) * n], src, group=self.group))
        out[self.rank * n:(self.rank + 1) * n].copy_(inp)
        self.bytes_in += (self.size - 1) * n * inp.element_size()
        return _Works(dist.batch_isend_irecv(ops), out, stream_ordered=self.native_gather)

    def all_gather(self, out, inp):
        """out (size * len(inp)) <- concatenation of every rank's inp"""
        self.bytes_in += (self.size - 1) * inp.numel() * inp.element_size()
        if self.native_gather:
            self.dist.all_gather_into_tensor(out, inp, group=self.group)
        else:
            n = inp.numel()
            for r in range(self.size):
                chunk = out[r * n:(r + 1) * n]
                if r == self.rank:
                    chunk.copy_(inp)
                self._finish(self.dist.broadcast(chunk, src=r, group=self.group, async_op=True), "a panel exchange (from rank %d)" % r)


class _Work(object):
    """An asynchronous gather in flight (torch.distributed Work) and the tensor it fills."""

    def __init__(self, work, tensor):
        self.work, self.tensor = work, tensor

    def wait(self):
        self.work.wait()


class _Works(object):
    """Several point-to-point transfers in flight (TGP_DIST_GATHER=p2p) and the tensor they fill."""

    def __init__(self, works, tensor, stream_ordered):
        self.works, self.tensor, self.stream_ordered = works, tensor, stream_ordered

    def wait(self):
        # RCCL: wait() orders the CURRENT stream behind the transfers, so every stream that reads the panel calls it (side,
        # keep and main stream do).  gloo: wait() blocks the host until completion and must not be repeated on a finished
        # request (a second wait on a gloo send / receive never returns).
        limit = 0.0 if self.stream_ordered else watchdog_seconds()
        for w in self.works:
            if limit > 0.0:
                import datetime
                try:
                    w.wait(datetime.timedelta(seconds=limit))
                except RuntimeError as e:
                    if _is_stall(e):
                        raise DistStall("a point-to-point panel transfer did not complete (limit %.1f s, TGP_DIST_WATCHDOG_S)" % limit)
                    raise
            else:
                w.wait()
        if not self.stream_ordered:
            self.works = ()


class _Done(object):
    """Handle of a gather that was issued synchronously on the then-current stream: wait() orders the
    now-current stream behind it (no-op for CPU tensors)."""

    def __init__(self, tensor=None):
        self.ev = None
        self.tensor = tensor                 # where the gathered panel is to be read from
        if tensor is not None and getattr(tensor, "is_cuda", False):
            import torch
            self.ev = torch.cuda.Event()
            self.ev.record(torch.cuda.current_stream(tensor.device))
            self.dev = tensor.device

    def wait(self):
        if self.ev is not None:
            import torch
            torch.cuda.current_stream(self.dev).wait_event(self.ev)


class SelfComm(object):
    """World of one (lets the distributed code path run on a single GPU)."""
    rank, size = 0, 1

    def all_gather_start(self, out, inp):
        # the "gathered" panel of a world of one is the rank's own rows, in the very layout they are stored in: no copy --
        # the update reads them where they are (17 GB of copies per factorisation at N = 65 536 otherwise, 77 ms)
        return _Done(inp)

    def broadcast(self, t, src):
        pass

    def all_reduce_sum(self, t):
        pass

    def all_reduce_max(self, t):
        pass

    def all_gather(self, out, inp):
        out[:inp.numel()].copy_(inp)


def local_replicate_vote(lib, Np, G, device, env=None):
    """This rank's own view: is a replicated factor wanted (TGP_DIST_REPLICATE; default yes with more than one rank)
    and does it fit in half of the free device memory?"""
    import torch
    env = os.environ.get("TGP_DIST_REPLICATE") if env is None else env
    if not (G > 1 or env == "1") or env == "0":
        return False
    # the copy itself plus what its solves allocate beside it: inverse slabs (2 Np S doubles) and their build scratch (Np S)
    full_bytes = int(lib.tgp_panel_elems(Np)) * 8 + 3 * Np * 1024 * 8
    return full_bytes < 0.5 * torch.cuda.mem_get_info(device)[0]


def agree_replicate(comm, vote, env_rank0=None):
    """One decision for all ranks: rank 0's TGP_DIST_REPLICATE setting is broadcast (per-rank environments may
    differ), and the factor is replicated only if EVERY rank has room for it (all-reduce MIN of the local votes,
    done as MAX of the negation).  ``vote(env)`` -> this rank's bool under that setting."""
    import torch
    dev = getattr(comm, "device", None) or "cpu"
    t = torch.zeros(1, dtype=torch.float64, device=dev)
    if comm.rank == 0:
        e = os.environ.get("TGP_DIST_REPLICATE") if env_rank0 is None else env_rank0
        t[0] = {"0": 0.0, "1": 1.0}.get(e, 2.0)                 # 2 = unset
    comm.broadcast(t, 0)
    env = {0: "0", 1: "1"}.get(int(t.item()), "")
    t[0] = 0.0 if vote(env) else 1.0
    comm.all_reduce_max(t)
    return t.item() == 0.0


class RankStreams(object):
    """The two extra streams of a rank -- high-priority look-ahead stream for the panel chain, a third one for the copies
    into the replicated factor -- each with the tgp_ctx whose kernels run on it."""

    def __init__(self, device):
        import torch
        from . import _lib
        lib = _lib.load_library()
        idx = device.index if device.index is not None else 0
        self._owned = (_lib.OwnedCtx(idx), _lib.OwnedCtx(idx))          # destroyed with this object
        self.side_stream = torch.cuda.Stream(device=device, priority=-1)
        self.ctx_side = self._owned[0].handle
        lib.tgp_set_stream(self.ctx_side, C.c_void_p(self.side_stream.cuda_stream))
        self.keep_stream = torch.cuda.Stream(device=device)
        self.ctx_keep = self._owned[1].handle
        lib.tgp_set_stream(self.ctx_keep, C.c_void_p(self.keep_stream.cuda_stream))


class HipLocalOps(object):
    """Local arithmetic of one rank on its GPU through the tgp_dd_* entry points."""

    def __init__(self, ctx, spec, n, G, g, device, replicate=None, streams=None):
        """``replicate``: keep a full copy of the factor on this rank for communication-free solves.  With more than
        one rank it MUST be the value ``agree_replicate`` returned (the same on every rank: the solve branches on it and
        a mixed decision would pair collectives with no partner); None = decide locally (world of one: its own share IS
        the whole factor, in the single-GPU layout, so it is "replicated" without a copy unless TGP_DIST_REPLICATE=0).
        ``streams``: a ``RankStreams`` to reuse (the engine behind the API builds one solver per problem size)."""
        import torch
        from . import _lib
        self.torch, self._lib, self.lib = torch, _lib, _lib.load_library()
        self.ctx, self.spec, self.n, self.G, self.g, self.device = ctx, spec, int(n), int(G), int(g), device
        self.Np = int(self.lib.tgp_padded_n(n))
        self.nB = self.Np // BLK
        self.loff = np.array([self.lib.tgp_dist_panel_off(p, self.Np, G, g) for p in range(self.nB + 1)], dtype=np.int64)
        self.nloc = panel_blocks(0, self.nB, g, G)
        self.cmax0 = panel_cmax(1, self.nB, G)
        # the padded send views of the panel exchange and of the replicated finish may overrun the share: a view is as long as
        # the LONGEST rank's (panel: cmax blocks; finish from panel k0: the largest tail of any rank), so this rank's needs that
        # minus its own in slack -- taken over every possible finish start (with a finish of 64 blocks on 3 ranks the difference
        # reaches 75 blocks, more than the 65 a fixed rule allowed: ADVICE r4)
        slack_blocks = self.cmax0 + 1
        own_total = int(self.lib.tgp_dist_local_elems(self.Np, G, g))
        for k0 in range(max(self.nB - FINISH_BLOCKS_MAX, 0), self.nB):
            tails = [int(self.lib.tgp_dist_local_elems(self.Np, G, r)) - int(self.lib.tgp_dist_panel_off(k0, self.Np, G, r)) for r in range(G)]
            need = max(tails) - (own_total - int(self.loff[k0]))
            slack_blocks = max(slack_blocks, -(-need // (BLK * BLK)) + 1)
        slack = slack_blocks * BLK * BLK
        self.A = torch.empty(int(self.loff[-1]) + slack, dtype=torch.float64, device=device)
        self.W = torch.empty(self.Np * 128, dtype=torch.float64, device=device)
        # [L_kk | W0 | W1] of a panel's diagonal block, and behind it room for the owner's rows of up to three earlier panels of
        # the group (TGP_DIST_CHAIN_BCAST: the column operands of the chain's strips travel with the diagonal block)
        self.bcast_full = torch.empty(BCAST_ELEMS + 3 * BLK * BLK, dtype=torch.float64, device=device)
        self.bcast = self.bcast_full[:BCAST_ELEMS]
        self.d_loff = torch.from_numpy(self.loff).to(device)
        self.kc = spec.to_c()
        self.main_stream = torch.cuda.current_stream(device)
        self.lib.tgp_set_stream(ctx, C.c_void_p(self.main_stream.cuda_stream))
        # panel chain (diagonal block, broadcast, local solves, all-gather) runs on a second stream
        # with its own context, concurrently with the bulk trailing update on the main stream; a third stream (own
        # context) carries the copies that build the replicated factor: they only need a gather to have
        # landed and its buffer not to be reused yet -- off the panel chain, where they cost up to 0.3 ms per panel
        st = streams or RankStreams(device)
        self._streams = st                    # (owns the two contexts)
        self.side_stream, self.ctx_side, self.keep_stream, self.ctx_keep = st.side_stream, st.ctx_side, st.keep_stream, st.ctx_keep
        # replicated factor for the solves: every panel is seen by every rank anyway (broadcast + all-gather); kept
        # in the single-GPU packed layout it lets the triangular sweeps run locally, without their 2 N/256 collectives
        self.Afull = None
        self.keep_copies = False              # factorize() copies the panels it sees into Afull
        if G == 1 and replicate is None and os.environ.get("TGP_DIST_REPLICATE") != "0":
            # a world of one holds every block row: its share is the single-GPU packed factor already
            self.Afull = self.A[:int(self.lib.tgp_panel_elems(self.Np))]
        else:
            if replicate is None:
                replicate = local_replicate_vote(self.lib, self.Np, G, device)
            if replicate:
                self.Afull = torch.empty(int(self.lib.tgp_panel_elems(self.Np)), dtype=torch.float64, device=device)
                self.keep_copies = True

    def _chk(self, rc, what, ctx=None):
        self._lib.check(ctx or self.ctx, rc, what)

    # -- stream choreography -------------------------------------------------------------------------
    def on_side(self):
        return self.torch.cuda.stream(self.side_stream)

    def side_wait_main(self):
        ev = self.torch.cuda.Event()
        ev.record(self.main_stream)
        self.side_stream.wait_event(ev)

    def main_wait_side(self):
        ev = self.torch.cuda.Event()
        ev.record(self.side_stream)
        self.main_stream.wait_event(ev)

    def _p(self, t, off=0):
        return C.c_void_p(t.data_ptr() + 8 * int(off))

    def _hl(self):
        return C.c_void_p(self.loff.ctypes.data)

    def empty(self, n):
        return self.torch.empty(n, dtype=self.torch.float64, device=self.device)

    def zeros(self, n):
        return self.torch.zeros(n, dtype=self.torch.float64, device=self.device)

    def to_device(self, a):
        return self.torch.from_numpy(np.ascontiguousarray(a, dtype=np.float64)).to(self.device)

    # -- factorisation ---------------------------------------------------------------------------
    def detach_replica(self):
        """the replicated factor and W have been handed out (a borrowed tgp_factor): forget them, allocate anew on demand"""
        self.Afull = None
        self.W = None

    def ensure_replica(self):
        if self.W is None:
            self.W = self.torch.empty(self.Np * 128, dtype=self.torch.float64, device=self.device)
        if self.keep_copies and self.Afull is None:
            self.Afull = self.torch.empty(int(self.lib.tgp_panel_elems(self.Np)), dtype=self.torch.float64, device=self.device)

    def kbuild(self, dX, dyerr):
        self.ensure_replica()
        self._chk(self.lib.tgp_dd_kbuild(self.ctx, C.byref(self.kc), self._p(dX), self.n, self._p(dyerr), self._p(self.A),
                                         self._p(self.d_loff), self.G, self.g), "tgp_dd_kbuild")

    def factor_diag(self, k):              # side stream
        self._chk(self.lib.tgp_dd_factor_diag(self.ctx_side, self._p(self.A), self._hl(), self.Np, k, self.G, self.g,
                                              self._p(self.W), self._p(self.bcast)), "tgp_dd_factor_diag", self.ctx_side)

    def trsm(self, k):                     # side stream
        self._chk(self.lib.tgp_dd_trsm(self.ctx_side, self._p(self.A), self._hl(), self.Np, k, self.G, self.g,
                                       self._p(self.W), self._p(self.bcast)), "tgp_dd_trsm", self.ctx_side)

    def panel_send_view(self, k, cmax):
        skip = BLK if owner(k, self.G) == self.g else 0
        o = int(self.loff[k]) + skip * BLK
        return self.A[o:o + cmax * BLK * BLK]

    # -- panel chain with the panel exchange off it (TGP_DIST_CHAIN_BCAST) -------------------------
    def bcast_payload(self, j):
        """what the owner of the j-th panel of a group broadcasts: the diagonal block and its inverses, then its rows of the j
        earlier panels of the group"""
        return self.bcast_full[:BCAST_ELEMS + j * BLK * BLK]

    def pack_ext(self, b, kgroup):         # side stream, owner of block b: its rows of panels kgroup .. b-1 behind the diagonal block
        for m in range(kgroup, b):
            src = int(self.loff[m]) + (b // self.G - first_round(m, self.g, self.G)) * BLK * BLK
            dst = BCAST_ELEMS + (m - kgroup) * BLK * BLK
            self.bcast_full[dst:dst + BLK * BLK].copy_(self.A[src:src + BLK * BLK])

    def strip_left(self, b, kgroup, from_bcast):   # side stream: block b's columns against panels kgroup .. b-1, depth 256 (b - kgroup)
        ext = self._p(self.bcast_full, BCAST_ELEMS) if from_bcast else None
        self._chk(self.lib.tgp_dd_strip_left(self.ctx_side, self._p(self.A), self._hl(), self._p(self.d_loff), self.Np, kgroup, b,
                                             self.G, self.g, ext), "tgp_dd_strip_left", self.ctx_side)

    def update(self, k, gathered, cmax, col_lo=0, col_hi=-1, side=False):
        ctx = self.ctx_side if side else self.ctx
        self._chk(self.lib.tgp_dd_update(ctx, self._p(self.A), self._p(self.d_loff), self.Np, k, self.G, self.g,
                                         self._p(gathered), cmax, col_lo, col_hi), "tgp_dd_update", ctx)

    def update2(self, k, gathered0, cmax0, gathered1, cmax1, col_lo=0, col_hi=-1):
        """after the pair of panels (k, k+1), depth 512; tile columns count from block k+2"""
        self._chk(self.lib.tgp_dd_update2(self.ctx, self._p(self.A), self._p(self.d_loff), self.Np, k, self.G, self.g,
                                          self._p(gathered0), cmax0, self._p(gathered1), cmax1, col_lo, col_hi),
                  "tgp_dd_update2")

    def queue_reset(self):                 # once per factorisation: counters of the queued bulk launches
        self._chk(self.lib.tgp_dd_queue_reset(self.ctx), "tgp_dd_queue_reset")

    def chain_exclusive(self, on):         # the side chain's diagonal blocks take a compute unit of their own
        self.lib.tgp_dd_set_exclusive(self.ctx_side, 1 if on else 0)

    def update_group(self, k, bufs, cmaxs, col_lo=0, col_hi=-1, side=False, queue_nres=0):
        """after the group of len(bufs) consecutive panels k, k+1, ... in one pass of depth 256 len(bufs); tile columns
        count from block k + len(bufs)"""
        ns = len(bufs)
        ctx = self.ctx_side if side else self.ctx
        ptrs = (C.c_void_p * ns)(*[b.data_ptr() for b in bufs])
        cm = (C.c_int * ns)(*[int(c) for c in cmaxs])
        if queue_nres > 0 and not side:
            self._chk(self.lib.tgp_dd_update_group_queued(ctx, self._p(self.A), self._p(self.d_loff), self.Np, k, self.G, self.g,
                                                          ns, ptrs, cm, col_lo, col_hi, int(queue_nres)),
                      "tgp_dd_update_group_queued", ctx)
            return
        self._chk(self.lib.tgp_dd_update_group(ctx, self._p(self.A), self._p(self.d_loff), self.Np, k, self.G, self.g, ns,
                                               ptrs, cm, col_lo, col_hi), "tgp_dd_update_group", ctx)

    def timing_event(self):
        return self.torch.cuda.Event(enable_timing=True)

    def fused_ok(self):
        """the whole update of a group as one launch that signals the panel chain from inside (tgp_dd_update_group_fused):
        only where this process hands over between streams by flags + stream wait-value (a first-use trial decides,
        csrc/handoff.hip) and with more than one rank; TGP_DIST_FUSED=0: two launches and an event, as before round 4"""
        env = os.environ.get("TGP_DIST_FUSED")
        if env == "0" or (env is None and self.G == 1):
            # a world of one runs whole-chip launches of dozens of rounds: there the fused form loses 0.6 % (the rest tiles
            # start in lockstep when the head ends and the chain waits a tile time for a slot; LAB_NOTES A.0); TGP_DIST_FUSED=1 forces it
            return False
        return int(self.lib.tgp_handoff_mode(self.ctx)) == 1

    def update_group_fused(self, k, bufs, cmaxs, head_cols, queue_nres=0):
        """main stream: everything right of the group k .. k+len(bufs)-1 in one launch, tile columns [0, head_cols) first"""
        ns = len(bufs)
        ptrs = (C.c_void_p * ns)(*[b.data_ptr() for b in bufs])
        cm = (C.c_int * ns)(*[int(c) for c in cmaxs])
        self._chk(self.lib.tgp_dd_update_group_fused(self.ctx, self._p(self.A), self._p(self.d_loff), self.Np, k, self.G, self.g,
                                                     ns, ptrs, cm, int(head_cols), int(queue_nres)), "tgp_dd_update_group_fused")

    def side_wait_head(self):
        """the side stream proceeds once the head columns of the last fused launch are done"""
        self._chk(self.lib.tgp_dd_wait_head(self.ctx, self.ctx_side), "tgp_dd_wait_head")

    @property
    def replicated(self):
        return self.Afull is not None

    def keep_diag(self, k):                # side stream, after the broadcast of panel k
        self._chk(self.lib.tgp_dd_keep_panel(self.ctx_side, self._p(self.Afull), self.Np, k, self.G, self._p(self.bcast),
                                             None, 0), "tgp_dd_keep_panel", self.ctx_side)

    def keep_rows(self, k, gathered, cmax, handle):   # keep stream, once the all-gather of panel k (`handle`) has landed
        with self.torch.cuda.stream(self.keep_stream):
            handle.wait()
            self._chk(self.lib.tgp_dd_keep_panel(self.ctx_keep, self._p(self.Afull), self.Np, k, self.G, None,
                                                 self._p(gathered), cmax), "tgp_dd_keep_panel", self.ctx_keep)

    def keeps_done(self):                  # an event after everything queued on the keep stream so far
        ev = self.torch.cuda.Event()
        ev.record(self.keep_stream)
        return ev

    def side_wait_keeps(self, ev):         # the side stream may reuse a gather buffer only after its keeps
        if ev is not None:
            self.side_stream.wait_event(ev)

    def main_wait_keeps(self):
        self.main_stream.wait_stream(self.keep_stream)

    # -- replicated finish ------------------------------------------------------------------------
    def tail_elems(self, k0, r):
        """doubles of rank r's share from panel k0 on"""
        return int(self.lib.tgp_dist_local_elems(self.Np, self.G, r)) - int(self.lib.tgp_dist_panel_off(k0, self.Np, self.G, r))

    def tail_send_view(self, k0, stride):
        o = int(self.loff[k0])
        v = self.A[o:o + stride]
        assert v.numel() == stride, "the share's slack is too small for the finish's send view (%d of %d doubles)" % (v.numel(), stride)
        return v

    def tail_finish(self, k0, gathered, stride):
        """main stream: assemble the packed trailing matrix of order Np - 256 k0 from everybody's shares (`gathered`, [G][stride]),
        factor it with the single-GPU schedule, copy this rank's blocks of the result back into its share.  Returns the
        1-based index of the first non-positive pivot (global), or 0.  Synchronises the stream."""
        m = self.Np - BLK * k0
        own_factor = self.G == 1 and self.Afull is not None and not self.keep_copies      # world of one: the share IS the factor
        if own_factor:
            tail = self._p(self.A, self.loff[k0])
        else:
            if self.Afull is not None:
                tail = self._p(self.Afull, int(self.lib.tgp_panel_off(k0, self.Np)))
            else:
                need = int(self.lib.tgp_panel_elems(m))
                if getattr(self, "_tail_buf", None) is None or self._tail_buf.numel() < need:
                    self._tail_buf = self.empty(need)
                tail = self._p(self._tail_buf)
            self._chk(self.lib.tgp_dd_tail_assemble(self.ctx, self._p(gathered), int(stride), self.Np, k0, self.G, tail),
                      "tgp_dd_tail_assemble")
        # the single-GPU schedule's look-ahead runs on the (now idle) chain stream, lent for this call: one more stream of the
        # context's own made the bulk and chain streams share a hardware queue
        self.lib.tgp_set_side_stream(self.ctx, C.c_void_p(self.side_stream.cuda_stream))
        try:
            rc = self.lib.tgp_d_potrf(self.ctx, tail, m, self._p(self.W, 2 * k0 * 128 * 128))
        finally:
            self.lib.tgp_set_side_stream(self.ctx, None)
        self._chk(rc, "tgp_d_potrf (finish)")
        # the tail-LOCAL index of a failed pivot is in `rc`; what tgp_d_potrf left in the context's info word must not be read
        # as a global index by info() afterwards (a LinAlgError naming minor 3560 for a failure at 65 000: ADVICE r4)
        self.lib.tgp_dd_info(self.ctx, 1)
        if not own_factor:
            self._chk(self.lib.tgp_dd_tail_scatter(self.ctx, tail, self.Np, k0, self.G, self.g, self._p(self.A), self._p(self.d_loff)),
                      "tgp_dd_tail_scatter")
        return (BLK * k0 + int(rc)) if rc > 0 else 0

    def potrs_full(self, rhs):             # main stream: rhs (Np) <- L^-T L^-1 rhs with the replicated factor
        self._chk(self.lib.tgp_d_potrs(self.ctx, self._p(self.Afull), self._p(self.W), self.Np, self._p(rhs)), "tgp_d_potrs")

    def streams_idle(self):
        """everything queued on the rank's three streams has finished (polled by the watchdog instead of a blocking wait)"""
        return self.main_stream.query() and self.side_stream.query() and self.keep_stream.query()

    def info(self):
        a = int(self.lib.tgp_dd_info(self.ctx, 1))
        b = int(self.lib.tgp_dd_info(self.ctx_side, 1))
        bad = [v for v in (a, b) if v > 0]
        return min(bad) if bad else 0

    # -- triangular solves -----------------------------------------------------------------------
    def fwd_diag(self, k, yk):
        self._chk(self.lib.tgp_dd_fwd_diag(self.ctx, self._p(self.A), self._hl(), k, self._p(self.W), self._p(yk)), "fwd_diag")

    def fwd_update(self, k, zk, yloc):
        self._chk(self.lib.tgp_dd_fwd_update(self.ctx, self._p(self.A), self._hl(), self.Np, k, self.G, self.g, self._p(zk),
                                             self._p(yloc)), "fwd_update")

    def bwd_partial(self, k, aloc, s):
        self._chk(self.lib.tgp_dd_bwd_partial(self.ctx, self._p(self.A), self._hl(), self.Np, k, self.G, self.g,
                                              self._p(aloc), self._p(s)), "bwd_partial")

    def bwd_diag(self, k, ak, s=None):
        self._chk(self.lib.tgp_dd_bwd_diag(self.ctx, self._p(self.A), self._hl(), k, self._p(self.W), self._p(ak),
                                           None if s is None else self._p(s)), "bwd_diag")

    def logdet_local(self, out):
        self._chk(self.lib.tgp_dd_logdet_local(self.ctx, self._p(self.A), self._p(self.d_loff), self.Np, self.n, self.G,
                                               self.g, self._p(out)), "logdet_local")

    def predict(self, dX, dalpha, dXs, m, dys):
        self._chk(self.lib.tgp_d_gp_predict(self.ctx, C.byref(self.kc), self._p(dX), self.n, self._p(dalpha), self._p(dXs), m,
                                            self._p(dys)), "tgp_d_gp_predict")


class DistributedCholesky(object):
    """Backend-agnostic orchestration: who owns what, what is communicated, in which order."""

    def __init__(self, ops, comm, timer=None):
        self.ops, self.comm = ops, comm
        self.G, self.g = comm.size, comm.rank
        assert (ops.G, ops.g) == (self.G, self.g)
        self.nB, self.Np = ops.nB, ops.Np
        # solve() branches on ops.replicated between a communication-free path and one full of collectives: every rank
        # must have taken the same decision (agree_replicate); checked once here, by a collective all ranks reach
        if self.G > 1 and hasattr(ops, "zeros"):
            flag = ops.zeros(2)
            r = 1.0 if getattr(ops, "replicated", False) else 0.0
            flag[0], flag[1] = r, -r
            comm.all_reduce_max(flag)
            if float(flag[0]) != -float(flag[1]):
                raise RuntimeError("ranks disagree on the replicated factor (use dist.agree_replicate)")
        # panels are taken in groups of `group` (4 from N = 28672 on, else 2: the single-GPU crossover); all-gathered
        # panels: two groups of buffers (the group the bulk update reads, the group being produced underneath it)
        self.group = int(os.environ.get("TGP_DIST_GROUP", "4" if self.Np >= 28672 else "2"))
        assert self.group in (1, 2, 3, 4)
        self.gathered = [ops.empty(max(self.G * ops.cmax0, 1) * BLK * BLK) for _ in range(2 * self.group)]
        self.timer = timer            # optional callable(): returns an event-like with .record()/.elapsed_time()
        self.update_ms = 0.0          # local trailing-update kernel time of the last factorize()
        self.update_flops = 0.0       # algorithmic flops of this rank's share
        self.update_launches = 0
        self.chain_ms = 0.0           # side stream: panel chains (diagonal blocks, broadcasts, local solves, gathers, strips)
        self.wait_ms = 0.0            # main stream: stalled behind the chain / the gathers between two bulk updates
        self.chain_form = "gather"    # "bcast": the last factorize() ran with the panel exchange off the chain (TGP_DIST_CHAIN_BCAST)
        self.bytes_received = 0       # payload this rank received during the last factorize(): counted by the communicator from
                                      # the tensors handed to its collectives (TorchComm.bytes_in); for communicators that do
                                      # not count (tests' in-process ones) the schedule's own sum, ~ 4 N^2 B (G-1)/G

    def _local_update_flops(self, k):
        """2 * 256 flops per lower-triangle element of this rank's block rows > k"""
        elems = 0
        q = first_round(k + 1, self.g, self.G)
        while block_of(q, self.g, self.G) < self.nB:
            b = block_of(q, self.g, self.G)
            elems += BLK * (b - k - 1) * BLK + BLK * (BLK + 1) // 2
            q += 1
        return 2.0 * BLK * elems

    def factorize(self):
        """Right-looking factorisation, panels taken in GROUPS of `self.group`, look-ahead on two streams.

        Side stream (high priority), for the group k .. k+GS-1: for every panel j of it, its diagonal block
        on its owner, broadcast, local solves, all-gather, then the depth-256 update of the tile columns of the
        group's later panels with it (one launch).  Main stream: Ua, the 2 GS tile columns of the NEXT
        group with all GS gathered panels (depth 256 GS), after which the side stream may start on that group,
        and then Ub, the bulk of the trailing matrix in one pass of the same depth, concurrent with it.
        Per-tile fixed costs of the update (C read + write, pipeline fill) fall from 13 % at depth 256 to 6.8 %
        at 512 and 3.5 % at 1024 -- the single-GPU driver's schedule, with collectives."""
        ops, comm, G, g, nB, GS = self.ops, self.comm, self.G, self.g, self.nB, self.group
        events, chain_events, wait_events = [], [], []
        # the chain is timed in every run that can act on it (rank-local events, a few microseconds per group)
        chain_timer = self.timer or getattr(ops, "timing_event", None)
        self.update_flops, self.update_launches, self.bytes_received = 0.0, 0, 0
        bytes_in0 = getattr(comm, "bytes_in", None)
        # also build the replicated factor for the solves (a world of one's share already is that factor: no copies)
        keep = bool(getattr(ops, "replicated", False)) and bool(getattr(ops, "keep_copies", True))

        # TGP_DIST_CHAIN_BCAST=1: the panel exchange OFF the chain.  Inside a group, what panel b's factorisation and local solves
        # need from other ranks of the earlier panels of the group is only block b's rows of them (256 x 256 each, held by b's
        # owner since its own local solves): the owner appends them to the broadcast of its diagonal block, every rank brings
        # its rows of b's columns up to date in one left-looking strip (depth 256 j, tgp_dd_strip_left), and the chain per
        # panel is strip -> diagonal block -> broadcast -> strip -> local solves with no exchange of a whole panel in it: the
        # all-gathers are started as before but only the bulk update (main stream) and the replicated factor wait for them.
        # Same arithmetic in the same order as the right-looking strips: the factor is bit-identical.  Off by default until a
        # real node has measured it.
        chain_bcast = os.environ.get("TGP_DIST_CHAIN_BCAST", "0") == "1" and hasattr(ops, "strip_left") and GS > 1
        self.chain_form = "bcast" if chain_bcast else "gather"

        def factor_and_gather(k, buf, kgroup=None):
            """panel k on the side stream: diagonal block on its owner, broadcast, local solves, all-gather"""
            own = owner(k, G)
            j = (k - kgroup) if (chain_bcast and kgroup is not None) else 0
            if j > 0 and g == own:
                ops.strip_left(k, kgroup, False)                 # the owner's own rows: every operand is local
            if g == own:
                ops.factor_diag(k)
                if j > 0:
                    ops.pack_ext(k, kgroup)
            comm.broadcast(ops.bcast_payload(j) if j > 0 else ops.bcast, own)
            if g != own:
                self.bytes_received += 8 * (BCAST_ELEMS + j * BLK * BLK)
                if j > 0:
                    ops.strip_left(k, kgroup, True)
            if keep:
                ops.keep_diag(k)
            ops.trsm(k)
            rem = nB - k - 1
            if rem == 0:
                return None, 0
            cmax = panel_cmax(k + 1, nB, G)                      # most blocks > k any rank holds
            self.bytes_received += 8 * (G - 1) * cmax * BLK * BLK
            send = ops.panel_send_view(k, cmax)
            return comm.all_gather_start(buf[:G * cmax * BLK * BLK], send), cmax     # the handle knows where the panel lands

        keep_events = {}                                         # id(first buffer of a set) -> event after that set's keeps

        def side_group(k, bufs):
            """panels k .. k+GS-1 (those that exist) on the side stream; returns [(gather handle, cmax)] per panel"""
            if chain_timer is not None:
                c0 = chain_timer()
                c0.record()                                      # on the side stream (the caller's `with ops.on_side()`)
            try:
                return _side_group(k, bufs)
            finally:
                if chain_timer is not None:
                    c1 = chain_timer()
                    c1.record()
                    chain_events.append((c0, c1))

        def _side_group(k, bufs):
            if keep:
                ops.side_wait_keeps(keep_events.pop(id(bufs[0]), None))     # these buffers were last read two groups ago
            out = []
            for j in range(GS):
                if k + j >= nB:
                    break
                if j > 0 and not chain_bcast:
                    w, c = out[j - 1]
                    w.wait()                                     # side stream: panel k+j-1 is on every rank
                    # panel k+j-1 against the tile columns of ALL later panels of the group, depth 256: every strip on the
                    # chain is one tile time of depth 256 (rounds 2-4 updated panel k+j's two columns against the j panels
                    # before it, depth 256 j: 24 / 41 / 59 us instead of 3 x 24 in the chain-bound phase,
                    # profiles/r04_strips_ab.txt)
                    ops.update_group(k + j - 1, [w.tensor], [c], 0, 2 * (GS - j), side=True)
                out.append(factor_and_gather(k + j, bufs[j], k))
                if keep and out[-1][0] is not None:
                    ops.keep_rows(k + j, out[-1][0].tensor, out[-1][1], out[-1][0])   # copied on the keep stream, off this chain
            if keep:
                keep_events[id(bufs[0])] = ops.keeps_done()
            return out

        def timed(fn):
            if self.timer is not None:
                e0, e1 = self.timer(), self.timer()
                e0.record()
                fn()
                e1.record()
                events.append((e0, e1))
            else:
                fn()

        # Chain-bound steps (this rank's share of the bulk shorter than the panel chain that runs beside it): the bulk as a
        # persistent grid that keeps compute units clear for the chain, whose diagonal blocks then take a unit of their own.
        # TGP_DIST_QUEUE: 0 never (default: without communication on the chain the clear units cost more than they bring,
        # profiles/r04_rank_slice.txt), -1 decide per step from the measured chain, 1..3 always with that many units per
        # shader engine -- a knob for the first run on a real node.
        # Rank-local decisions: no collective depends on them.
        # How long a group's chain takes is MEASURED, not assumed: the side stream's span of every group is timed (it
        # contains the broadcasts and panel exchanges of a real node), and a step is chain-bound when the last chain that
        # has finished took longer than this step's bulk would with every slot.  TGP_DIST_CHAIN_US=<us per panel> replaces
        # the measurement by a constant (tests; rounds 2-3 used 600, tuned on one GPU without communication, which
        # cost 3.7 % at 8 ranks because it kept units clear for a chain that did not need them).
        queue_mode = int(os.environ.get("TGP_DIST_QUEUE", "0"))
        chain_env = os.environ.get("TGP_DIST_CHAIN_US")
        can_queue = queue_mode != 0 and hasattr(ops, "queue_reset")
        if can_queue:
            ops.queue_reset()

        def chain_us_estimate():
            if chain_env is not None:
                return float(chain_env) * GS
            for c0, c1 in reversed(chain_events):                # the most recent chain that is known to have finished
                if c1.query():
                    return 1e3 * c0.elapsed_time(c1)
            return 0.0                                           # nothing measured yet: plain launches
        fused = hasattr(ops, "update_group_fused") and ops.fused_ok()

        def bulk_queue_units(k):
            """clear units per shader engine for the bulk after group k (0 = plain launch)"""
            if not can_queue:
                return 0
            if queue_mode > 0:
                return min(queue_mode, 3)
            tiles = GS * self._local_update_flops(k + GS - 1) / (2.0 * BLK * GS * 128 * 128)
            if tiles < 64:
                return 0
            tile_us = 63.0 * GS                                   # measured: 126 us per 128 x 128 tile at depth 512
            chain_us = chain_us_estimate()
            for r in (3, 2, 1):
                if tiles * tile_us / (512 - 64 * r) <= chain_us:
                    return r
            return 0

        # Replicated finish (TGP_DIST_FINISH = blocks of 256 rows, 0 = off): the last rows are chain-bound on every
        # rank -- per panel a diagonal block, a broadcast, local solves and an all-gather for a bulk update of a handful of
        # tiles -- so from the group boundary k_fin on the ranks exchange their shares of the trailing matrix in ONE
        # all-gather and every rank factors it with the single-GPU schedule (HipLocalOps.tail_finish): the same volume over
        # the links, one collective instead of two per panel, ~6 ms of arithmetic at 8192 rows where the chain needs ~10.
        # Default: 16 blocks (4096 rows), never more than a quarter of the matrix.  Without communication, and with the chain's
        # latency kernels of the round's end, the finish COSTS 1.6 ms at 16 blocks and 3.9 ms at 32 (rank 7 of 8 at
        # N = 65 536: 184.1 ms without, 185.1 / 185.7 / 188.0 with 8 / 16 / 32; profiles/r04_rank_slice.txt); every panel it
        # takes off the chain is a broadcast and an all-gather less on a real node, so the first hardware run should sweep it.
        fin = min(int(os.environ.get("TGP_DIST_FINISH", min(16, nB // 4))), FINISH_BLOCKS_MAX)
        k_fin = None
        if fin >= GS and hasattr(ops, "tail_finish") and nB > 1:
            k_fin = GS * (-(-max(nB - fin, 0) // GS))
            if k_fin >= nB:
                k_fin = None
        ops.side_wait_main()                                     # K build (main) precedes panel 0
        cur_w = None
        if k_fin != 0:
            with ops.on_side():
                cur_w = side_group(0, self.gathered[:GS])
        k, flip = 0, 0
        while k + GS < nB and (k_fin is None or k < k_fin):
            last_regular = k_fin is not None and k + GS >= k_fin         # the group whose update completes the finish's matrix
            cur = self.gathered[flip * GS:(flip + 1) * GS]
            nxt = self.gathered[(1 - flip) * GS:(2 - flip) * GS]
            if self.timer is not None:
                w0 = self.timer()
                w0.record()
            for w, _ in cur_w:
                w.wait()                                         # main stream: the whole group is on every rank
            if self.timer is not None:
                w1 = self.timer()
                w1.record()
                wait_events.append((w0, w1))
            cm = [c for _, c in cur_w]
            cur = [w.tensor for w, _ in cur_w]                   # where each gathered panel of the group is (see _Done / _Work)
            units = bulk_queue_units(k)
            if last_regular:
                timed(lambda: ops.update_group(k, cur, cm, 0, -1))           # nothing runs beside it: one plain launch
                self.update_flops += GS * self._local_update_flops(k + GS - 1)
                self.update_launches += 1
                k += GS
                break
            if fused:
                # Ua (the next group's columns) and Ub (the bulk) in ONE launch, Ua's tiles first; the launch itself
                # releases the side stream when they are done: one ramp and one tail per group instead of two
                timed(lambda: ops.update_group_fused(k, cur, cm, 2 * GS, queue_nres=units))
                ops.side_wait_head()
            else:
                timed(lambda: ops.update_group(k, cur, cm, 0, 2 * GS))       # Ua: the next group's columns
                ops.side_wait_main()
            if can_queue:
                ops.chain_exclusive(units > 0)
            with ops.on_side():
                nxt_w = side_group(k + GS, nxt)
            if not fused:
                if units > 0:
                    timed(lambda: ops.update_group(k, cur, cm, 2 * GS, -1, queue_nres=units))   # Ub, keeping units clear
                else:
                    timed(lambda: ops.update_group(k, cur, cm, 2 * GS, -1))  # Ub: the bulk
            self.update_flops += GS * self._local_update_flops(k + GS - 1)
            self.update_launches += 1 if fused else 2
            cur_w, k, flip = nxt_w, k + GS, 1 - flip
        ops.main_wait_side()                                     # the last chain has no gather to wait on
        if can_queue:
            ops.chain_exclusive(False)
        if keep:
            ops.main_wait_keeps()                                # the replicated factor is complete before the solves
        tail_info = 0
        if k_fin is not None:
            assert k == k_fin
            stride = max(ops.tail_elems(k_fin, r) for r in range(G))
            if getattr(self, "_tail_gather", None) is None or self._tail_gather.numel() < G * stride:
                self._tail_gather = ops.empty(G * stride)
            if not hasattr(comm, "bytes_in"):
                self.bytes_received += 8 * (G - 1) * stride
            h = comm.all_gather_start(self._tail_gather[:G * stride], ops.tail_send_view(k_fin, stride))
            h.wait()
            self._await_device(chain_events)                     # (the finish synchronises the stream: watchdog first)
            tail_info = ops.tail_finish(k_fin, h.tensor, stride)
        # any rank's failure is everybody's failure; report the smallest failing index
        big = 1e18
        self._await_device(chain_events)
        mine = ops.info()                                        # synchronises the stream
        if tail_info > 0:
            mine = min(mine, tail_info) if mine > 0 else tail_info      # an earlier failure of the chain has the smaller index
        self.update_ms = sum(a.elapsed_time(b) for a, b in events) if events else 0.0
        self.chain_ms = sum(a.elapsed_time(b) for a, b in chain_events) if chain_events else 0.0
        self.wait_ms = sum(a.elapsed_time(b) for a, b in wait_events) if wait_events else 0.0
        if bytes_in0 is not None:
            self.bytes_received = comm.bytes_in - bytes_in0
        t = ops.zeros(1)
        t[0] = -(float(mine) if mine > 0 else big)
        comm.all_reduce_max(t)
        first = -float(t[0])
        return 0 if first >= big else int(first)

    def _await_device(self, chain_events):
        """The watchdog of the device side (RCCL: collectives are stream-ordered, the host runs ahead and would block for ever
        in the final synchronisation if a peer never arrived or a stream were parked on a flag nobody sets): poll instead of
        blocking, and give up -- DistStall -- when neither the panel chain (its per-group events) nor the streams as a whole
        have advanced for TGP_DIST_WATCHDOG_S seconds.  Local operations without streams (the tests' NumPy stand-in) have
        nothing to wait for here: their collectives block on the host and time out in TorchComm."""
        ops = self.ops
        idle = getattr(ops, "streams_idle", None)
        limit = watchdog_seconds()
        if idle is None or limit <= 0.0:
            return
        import time
        last, seen = time.monotonic(), -1
        while not idle():
            done = sum(1 for _, c1 in chain_events if c1.query())
            now = time.monotonic()
            if done != seen:
                seen, last = done, now
            elif now - last > limit:
                raise DistStall("rank %d of %d: the device streams have not advanced for %.1f s (%d of %d panel-chain groups "
                                "finished); TGP_DIST_WATCHDOG_S sets the limit" % (self.g, self.G, limit, max(seen, 0), len(chain_events)))
            time.sleep(0.0005 if now - last < 0.05 else 0.01)

    def solve(self, y_full):
        """alpha (Np, replicated) = (L L^T)^-1 y; y_full is the replicated right-hand side (Np).
        Forward: the owner of block k solves its diagonal block in place in z and broadcasts the 2 KB
        result; every rank then updates its own rows.  Backward: every rank contributes the partial
        sum of its rows, one 2 KB all-reduce, the owner finishes the block."""
        ops, comm, G, g, nB = self.ops, self.comm, self.G, self.g, self.nB
        if getattr(ops, "replicated", False):
            # every rank holds the whole factor (factorize() kept the panels): both sweeps locally, no communication
            alpha = ops.zeros(self.Np)
            alpha.copy_(y_full)
            ops.potrs_full(alpha)
            return alpha
        nloc = ops.nloc
        yloc = ops.zeros(max(nloc, 1) * BLK)
        yv = [yloc[lb * BLK:(lb + 1) * BLK] for lb in range(nloc)]
        for lb in range(nloc):
            b = block_of(lb, g, G)
            yv[lb].copy_(y_full[b * BLK:(b + 1) * BLK])
        z = ops.zeros(self.Np)
        zv = [z[k * BLK:(k + 1) * BLK] for k in range(nB)]
        for k in range(nB):                                      # L z = y
            own = owner(k, G)
            if g == own:
                zv[k].copy_(yv[k // G])
                ops.fwd_diag(k, zv[k])
            comm.broadcast(zv[k], own)
            ops.fwd_update(k, zv[k], yloc)
        aloc = ops.zeros(max(nloc, 1) * BLK)
        av = [aloc[lb * BLK:(lb + 1) * BLK] for lb in range(nloc)]
        s = ops.zeros(BLK)
        for k in range(nB - 1, -1, -1):                          # L^T a = z
            ops.bwd_partial(k, aloc, s)
            comm.all_reduce_sum(s)
            if g == owner(k, G):
                ak = av[k // G]
                ak.copy_(zv[k])
                ops.bwd_diag(k, ak, s)                           # a_k = L_kk^-T (z_k - s)
        alpha = ops.zeros(self.Np)
        for lb in range(nloc):
            b = block_of(lb, g, G)
            alpha[b * BLK:(b + 1) * BLK].copy_(av[lb])
        comm.all_reduce_sum(alpha)
        return alpha

    def logdet(self):
        out = self.ops.zeros(1)
        self.ops.logdet_local(out)
        self.comm.all_reduce_sum(out)
        return out


class DistributedGP(object):
    """bench.py's multi-GPU step: K build -> distributed Cholesky -> solves -> sharded predict,
    inputs resident on every GPU."""

    def __init__(self, ctx, spec, X, y, y_err, Xs, comm=None, device=None, profile=False):
        import torch
        self.torch = torch
        self.profile = profile
        if comm is None:
            import torch.distributed as dist
            comm = TorchComm() if dist.is_available() and dist.is_initialized() else SelfComm()
        self.comm = comm
        G, g = comm.size, comm.rank
        if device is None:
            device = torch.device("cuda", torch.cuda.current_device())
        self.n, self.m = len(y), len(Xs)
        from . import _lib
        lib = _lib.load_library()
        Np = int(lib.tgp_padded_n(self.n))
        replicate = agree_replicate(comm, lambda env: local_replicate_vote(lib, Np, G, device, env)) if G > 1 else None
        self.ops = HipLocalOps(ctx, spec, self.n, G, g, device, replicate=replicate)
        timer = (lambda: torch.cuda.Event(enable_timing=True)) if profile else None
        self.chol = DistributedCholesky(self.ops, comm, timer=timer)
        o = self.ops
        from ._lib import as_xy
        self.dX = o.to_device(as_xy(X))
        self.dyerr = o.to_device(y_err)
        ypad = np.zeros(o.Np)
        ypad[:self.n] = y
        self.dy = o.to_device(ypad)
        # prediction points: contiguous shards, the last one may be short
        per = -(-self.m // G)
        lo, hi = min(g * per, self.m), min((g + 1) * per, self.m)
        self.shard = (lo, hi)
        self.dXs = o.to_device(as_xy(Xs)[lo:hi]) if hi > lo else None
        self.dys = o.zeros(max(hi - lo, 1))
        self.alpha = None
        self.logdet = None

    def step(self, acc=None):
        """One full pass; when `acc` is a dict and profiling is on, phase times (ms, this rank) and
        the trailing-update roofline inputs are accumulated into it."""
        o, torch = self.ops, self.torch
        ev = [torch.cuda.Event(enable_timing=True) for _ in range(5)] if (self.profile and acc is not None) else None
        if ev: ev[0].record()
        o.kbuild(self.dX, self.dyerr)
        if ev: ev[1].record()
        info = self.chol.factorize()
        if info != 0:
            raise np.linalg.LinAlgError("%d-th leading minor of the array is not positive definite" % info)
        if ev: ev[2].record()
        self.alpha = self.chol.solve(self.dy)
        self.logdet = self.chol.logdet()
        if ev: ev[3].record()
        lo, hi = self.shard
        if hi > lo:
            o.predict(self.dX, self.alpha, self.dXs, hi - lo, self.dys)
        if ev:
            ev[4].record()
            torch.cuda.synchronize()
            for name, a, b in (("kbuild_ms", 0, 1), ("chol_ms", 1, 2), ("trsv_ms", 2, 3), ("predict_ms", 3, 4)):
                acc[name] = acc.get(name, 0.0) + ev[a].elapsed_time(ev[b])
            acc["syrk_ms"] = acc.get("syrk_ms", 0.0) + self.chol.update_ms
            acc["syrk_flops"] = acc.get("syrk_flops", 0.0) + self.chol.update_flops
            acc["syrk_launches"] = acc.get("syrk_launches", 0.0) + self.chol.update_launches
        return self.alpha, self.dys

    def gather_predictions(self):
        """full (m,) prediction vector on every rank (tests)"""
        per = -(-self.m // self.comm.size)
        buf = self.ops.zeros(per * self.comm.size)
        lo, hi = self.shard
        mine = self.ops.zeros(per)
        if hi > lo:
            mine[:hi - lo].copy_(self.dys[:hi - lo])
        self.comm.all_gather(buf, mine)
        return buf[:self.m]


# ---- the multi-GPU route behind the drop-in API ---------------------------------------------------------------------
# SPMD use: every rank runs the same script on the same data (``torchrun``), exactly as it would on one GPU.  With the
# engine enabled, ``ops.gp_solve`` / ``ops.gp_predict`` -- and through them ``GPInterpolation.predict`` /
# ``return_gp_predict`` (treegp/gp_interp.py:143-194), ``log_likelihood.log_likelihood`` (treegp/log_likelihood.py:21-41),
# ``predict_fields`` and the kept-factor calls (posterior covariance, gp_interp.py:184-192) -- factorise with
# ``DistributedCholesky`` and shard the query points; everything around them (normalize, white noise, mean function,
# ``_alpha`` cache, LinAlgError on every rank) is the single-GPU host code, untouched.
_tls = threading.local()
_process_engine = None
DEFAULT_MIN_N = 32768            # below this one GPU finishes a solve before eight have exchanged their panels


class DistEngine(object):
    """One rank's end of the distributed GP solve / predict.  ``comm``: TorchComm (default when torch.distributed is
    initialised), SelfComm, or any object with the same five collectives (the tests' in-process communicator)."""

    def __init__(self, comm=None, device=None, min_n=None, profile=False):
        import torch
        from . import _lib
        self.torch, self._lib, self.lib = torch, _lib, _lib.load_library()
        if comm is None:
            import torch.distributed as dist
            comm = TorchComm() if dist.is_available() and dist.is_initialized() else SelfComm()
        self.comm = comm
        if device is None:
            device = torch.device("cuda", torch.cuda.current_device())
        self.device = device
        self.min_n = int(os.environ.get("TGP_DIST_MIN_N", DEFAULT_MIN_N)) if min_n is None else int(min_n)
        self.profile = bool(profile)
        # never the process-wide context: its stream follows torch's.  Owned: destroyed with the engine and the last factor
        # handle that was created on it
        self._owned_ctx = _lib.OwnedCtx(device.index if device.index is not None else 0)
        self.ctx = self._owned_ctx.handle
        self.streams = RankStreams(device)
        self._solver = None                   # (n, HipLocalOps, DistributedCholesky) of the last problem size
        self._pin = {}                        # name -> pinned host staging tensor (_h2d / _d2h)
        self._replicate = {}                  # Np -> the collective replicate decision taken for that size (sticky: the same on
                                              # every rank and unaffected by factors handed out since)
        self.acc = {}                         # profile=True: phase times (ms, this rank) summed over calls; reset by the caller

    # -- plumbing -------------------------------------------------------------------------------------------------
    def _solver_for(self, spec, n):
        torch = self.torch
        G, g = self.comm.size, self.comm.rank
        if self._solver is not None and self._solver[0] == n:
            ops, chol = self._solver[1], self._solver[2]
            ops.spec, ops.kc = spec, spec.to_c()
        else:
            self._solver = None               # release the previous size's buffers before allocating
            Np = int(self.lib.tgp_padded_n(n))
            replicate = None
            if G > 1:
                # decided once per engine and size: a kept factor that is still alive (GPInterpolation._factor) lowers the free
                # memory the next vote would see, and a decision that flips under the caller turns return_cov from working into
                # NotImplementedError after all the collective work is done
                if Np not in self._replicate:
                    # what this process's single-GPU context still holds from earlier solves (17 GB per N = 65 536 factor) is
                    # given back first: memory parked there must not make ONE rank vote "no room" or run out later
                    self._lib.release_process_caches(self.device.index if self.device.index is not None else 0)
                    self._replicate[Np] = agree_replicate(self.comm, lambda env: local_replicate_vote(self.lib, Np, G, self.device, env))
                replicate = self._replicate[Np]
            ops = HipLocalOps(self.ctx, spec, n, G, g, self.device, replicate=replicate, streams=self.streams)
            timer = (lambda: torch.cuda.Event(enable_timing=True)) if self.profile else None
            chol = DistributedCholesky(ops, self.comm, timer=timer)
            self._solver = (n, ops, chol)
        ops.main_stream = torch.cuda.current_stream(self.device)
        self.lib.tgp_set_stream(self.ctx, C.c_void_p(ops.main_stream.cuda_stream))
        return ops, chol

    def _add(self, name, v):
        self.acc[name] = self.acc.get(name, 0.0) + v

    # Host arrays cross the boundary through persistent PINNED staging tensors: a transfer from / to pageable memory makes the
    # runtime pin the caller's pages on the fly and unpin them asynchronously, behind the call -- 10 - 28 ms that the NEXT call
    # then waits for (csrc/api.hip: h2d, the single-GPU side of the same finding)
    def _pinned(self, key, numel):
        t = self._pin.get(key)
        if t is None or t.numel() < numel:
            t = self.torch.empty(max(int(numel), 1), dtype=self.torch.float64).pin_memory()
            self._pin[key] = t
        return t[:numel]

    def _h2d(self, key, a):
        a = np.ascontiguousarray(a, dtype=np.float64)
        p = self._pinned(key, a.size)
        p.copy_(self.torch.from_numpy(a.reshape(-1)))
        return p.to(self.device, non_blocking=True).view(a.shape)

    def _d2h(self, key, t):
        p = self._pinned(key, t.numel())
        p.copy_(t.reshape(-1), non_blocking=True)
        self.torch.cuda.current_stream(self.device).synchronize()
        return p.numpy().copy()

    # -- seams S2 / S3 (include/tgp.h) on G GPUs --------------------------------------------------------------------
    def gp_solve(self, spec, X, y, y_err=None, keep=False, want_alpha=True):
        """(alpha, logdet, y.alpha, factor|None) like ``ops.gp_solve``, every rank returning the same values.  ``keep``
        hands out the replicated factor (an ``ops.Factor`` on borrowed memory) for the kept-factor calls."""
        from . import ops as _ops
        from ._lib import as_xy, f64
        torch = self.torch
        X2 = as_xy(X)
        n = X2.shape[0]
        y = f64(y)
        o, chol = self._solver_for(spec, n)
        dX = self._h2d("X", X2)
        de = self._h2d("yerr", np.zeros(n) if y_err is None else f64(y_err))
        ypad = np.zeros(o.Np)
        ypad[:n] = y
        dy = self._h2d("y", ypad)
        ev = [torch.cuda.Event(enable_timing=True) for _ in range(4)] if self.profile else None
        if ev: ev[0].record()
        o.kbuild(dX, de)
        if ev: ev[1].record()
        info = chol.factorize()                              # all-reduced: the same verdict on every rank
        if info != 0:
            raise np.linalg.LinAlgError("%d-th leading minor of the array is not positive definite" % info)
        if ev: ev[2].record()
        alpha = chol.solve(dy)
        logdet = chol.logdet()
        if ev: ev[3].record()
        a = self._d2h("alpha", alpha)[:n].copy()             # synchronises
        logdet = float(logdet[0])
        if ev:
            for name, i, j in (("kbuild_ms", 0, 1), ("chol_ms", 1, 2), ("trsv_ms", 2, 3)):
                self._add(name, ev[i].elapsed_time(ev[j]))
            for name, v in (("syrk_ms", chol.update_ms), ("syrk_flops", chol.update_flops), ("syrk_launches", chol.update_launches),
                            ("chain_ms", chol.chain_ms), ("gather_wait_ms", chol.wait_ms), ("bytes_received", chol.bytes_received),
                            ("solves", 1)):
                self._add(name, v)
        self.chain_form = chol.chain_form                    # which panel chain ran (goes into the N > 1 bench line)
        factor = None
        if keep:
            if not o.replicated:
                raise NotImplementedError("kept-factor calls (posterior covariance, several fields, likelihood gradient) need "
                                          "the replicated factor, which does not fit on this GPU (or TGP_DIST_REPLICATE=0)")
            h = C.c_void_p()
            self._lib.check(self.ctx, self.lib.tgp_factor_borrow(self.ctx, o._p(o.Afull), o._p(o.W), n, C.byref(h)),
                            "tgp_factor_borrow")
            if o.keep_copies:
                # G > 1: the handle takes the replica and the inverted diagonal blocks; the rank's share, the gather buffers
                # and the solver stay cached, and the next factorisation gets a fresh replica (HipLocalOps.ensure_replica)
                factor = _ops.Factor(self.ctx, h, n, keepalive=(o.Afull, o.W, self._owned_ctx))
                o.detach_replica()
            else:
                factor = _ops.Factor(self.ctx, h, n, keepalive=(o.A, o.Afull, o.W, self._owned_ctx))
                self._solver = None                          # world of one: the share IS the factor and now belongs to the handle
        return (a if want_alpha else None), logdet, float(np.dot(y, a)), factor

    def gp_predict(self, spec, X, alpha, Xs):
        """ys (m,) = k(Xs, X) alpha on every rank: the query points in contiguous shards, one all-gather of the results."""
        from ._lib import as_xy, f64
        torch = self.torch
        G, g = self.comm.size, self.comm.rank
        X2, Xs2 = as_xy(X), as_xy(Xs)
        n, m = X2.shape[0], Xs2.shape[0]
        self.lib.tgp_set_stream(self.ctx, C.c_void_p(torch.cuda.current_stream(self.device).cuda_stream))
        per = -(-m // G)
        lo, hi = min(g * per, m), min((g + 1) * per, m)
        mine = torch.zeros(max(per, 1), dtype=torch.float64, device=self.device)
        if hi > lo:
            dX = self._h2d("X", X2)
            da = self._h2d("alpha_in", f64(alpha))
            dXs = self._h2d("Xs", Xs2[lo:hi])
            kc = spec.to_c()
            rc = self.lib.tgp_d_gp_predict(self.ctx, C.byref(kc), C.c_void_p(dX.data_ptr()), n, C.c_void_p(da.data_ptr()),
                                           C.c_void_p(dXs.data_ptr()), hi - lo, C.c_void_p(mine.data_ptr()))
            self._lib.check(self.ctx, rc, "tgp_d_gp_predict")
            if self.profile:
                self._add("predict_ms", self._lib.timings(self.ctx)[3])
                self._add("predict_points", hi - lo)
        if G == 1:
            return self._d2h("ys", mine[:m])
        buf = torch.empty(per * G, dtype=torch.float64, device=self.device)
        self.comm.all_gather(buf, mine)
        return self._d2h("ys", buf[:m])


def enable(comm=None, device=None, min_n=None, profile=False, thread_local=False):
    """Switch the multi-GPU route on for this process (or, ``thread_local``, this thread: the tests' virtual ranks) and
    shard the pair binning of ``two_pcf`` over the same ranks.  Problems below ``min_n`` training points (default
    TGP_DIST_MIN_N or 32768) stay on the single-GPU path unless a ``GPInterpolation(backend="dist")`` asks otherwise.
    Collective: every rank must call it.  Returns the engine (``.acc`` holds phase timings with ``profile=True``)."""
    global _process_engine
    from . import ops as _ops
    eng = DistEngine(comm=comm, device=device, min_n=min_n, profile=profile)
    if hasattr(eng.comm, "probe_gather") and eng.comm.gather_probe is None:
        eng.comm.probe_gather()               # which panel exchange this node's links favour (collective; rank 0 decides)
    if thread_local:
        _tls.engine = eng
    else:
        _process_engine = eng
    _ops.set_pair_comm(eng.comm)
    return eng


def disable():
    global _process_engine
    from . import ops as _ops
    if getattr(_tls, "engine", None) is not None:
        _tls.engine = None
    else:
        if hasattr(_process_engine, "close"):
            _process_engine.close()                        # a worker pool: stop its processes
        _process_engine = None
    _ops.set_pair_comm(None)


_warned_no_group = False


def _current_engine(n=None):
    """The enabled engine of this thread / process.  TGP_DIST=1 enables one implicitly on first need -- a solve of at least
    TGP_DIST_MIN_N points (`n`; None: any) -- when torch.distributed is initialised; when it is not (a plain single-GPU
    script run with the variable exported globally) the process stays on its one GPU, with one warning."""
    global _process_engine, _warned_no_group
    eng = getattr(_tls, "engine", None) or _process_engine
    if eng is None and os.environ.get("TGP_DIST") == "1":
        if n is not None and n < int(os.environ.get("TGP_DIST_MIN_N", DEFAULT_MIN_N)):
            return None                               # below the threshold no engine is needed: do not demand one
        import sys
        td = sys.modules.get("torch.distributed")
        if td is None or not (td.is_available() and td.is_initialized()):
            eng = _pool_engine(explicit=False)             # an ordinary single-process script: worker pool over the visible GPUs
            if eng is not None:
                return eng
            if not _warned_no_group:
                import warnings
                warnings.warn("TGP_DIST=1 but torch.distributed is not initialised in this process and there is one GPU (or "
                              "TGP_DIST_POOL=0, or this is a rank of a torchrun job): staying on the single-GPU path "
                              "(init_process_group first, or call treegp_amd.dist.enable(comm))",
                              RuntimeWarning, stacklevel=3)
                _warned_no_group = True
            return None
        eng = enable()
    return eng


def _pool_engine(explicit):
    """The worker-pool engine of this process (dist_pool.py), started on first need: for a process that is not a rank of a
    torchrun job.  ``explicit`` (backend="dist"): even a pool of one worker is started; otherwise (TGP_DIST=1) only with more
    than one worker to gain from.  None when a pool may not or need not be started."""
    global _process_engine
    from . import dist_pool
    if not dist_pool.pool_allowed():
        return None
    if not explicit and dist_pool.default_world() < 2:
        return None
    eng = dist_pool.PoolEngine()
    _process_engine = eng
    return eng


class scope(object):
    """``with dist.scope("dist")``: every solve inside takes the multi-GPU route whatever its size (an engine must be
    enabled or torch.distributed initialised); ``"single"``: none does; ``None``: the engine's size threshold decides."""

    def __init__(self, mode):
        if mode not in (None, "dist", "single"):
            raise ValueError("backend must be None, 'dist' or 'single'; got %r" % (mode,))
        self.mode = mode

    def __enter__(self):
        self.prev = getattr(_tls, "mode", None)
        if self.mode is not None:
            _tls.mode = self.mode
        return self

    def __exit__(self, *exc):
        _tls.mode = self.prev
        return False


def engine_for(n):
    """The engine a solve with n training points should go through, or None for the single-GPU path."""
    mode = getattr(_tls, "mode", None)
    if mode == "single":
        return None
    if mode == "dist":
        eng = _current_engine()
        if eng is None:
            import sys
            td = sys.modules.get("torch.distributed")
            if td is not None and td.is_available() and td.is_initialized():
                eng = enable()
            else:
                eng = _pool_engine(explicit=True)          # an unmodified single-process script: one worker per visible GPU
                if eng is None:
                    raise RuntimeError('backend="dist" needs treegp_amd.dist.enable(comm) or an initialised torch.distributed '
                                       '(the worker pool of dist_pool.py is switched off here: TGP_DIST_POOL=0, or this process is '
                                       'a rank of a torchrun job that has not called init_process_group yet)')
        return eng
    eng = _current_engine(n)
    return eng if (eng is not None and n >= eng.min_n) else None

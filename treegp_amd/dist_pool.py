"""The multi-GPU route for an UNMODIFIED single-process script ("one process drives the node's GPUs"; the reference's
own usage, docs/treegp_gp_interp.rst:96-114, is one ``python script.py`` -- constructor treegp/gp_interp.py:54-67, ``predict``
:143, ``initialize`` :196).

``GPInterpolation(backend="dist")`` or ``TGP_DIST=1`` in a process that is NOT one rank of a ``torchrun`` job starts, once, G
fresh worker processes -- one per visible GPU, a torch.distributed group among them (nccl = RCCL) -- and keeps them for the life
of the parent.  Each worker is an ordinary rank of the SPMD route (``treegp_amd.dist.DistEngine``: row-block-cyclic Cholesky,
sharded predict); the parent ships X / y / y_err / X* to all of them through one POSIX shared-memory segment per call and reads
rank 0's alpha / predictions from it.  Problem-size caches (each rank's share of K, the gather buffers, the replicated factor)
live in the workers and are reused from call to call.

Workers are CHILD processes started with the ``spawn`` method (fresh interpreters): nothing is re-executed in a process that
has touched a GPU, and a worker that dies or stalls (``DistStall``) fails the call in the parent -- the pool is then torn
down and the next call starts a fresh one.

Not served by the pool: kept-factor calls (posterior covariance, several fields, likelihood gradient) -- the factor lives in
other processes; such a solve runs on the parent's own GPU, with one warning.  ``torchrun`` jobs keep the SPMD route.

Environment: TGP_DIST_POOL_WORLD (workers; default: visible GPUs), TGP_DIST_BACKEND (nccl | gloo), TGP_ONE_DEVICE=1 (all
workers on GPU 0: rehearsal with gloo on a one-GPU box), TGP_DIST_POOL=0 (never start a pool)."""
import atexit
import os
import socket
import warnings

import numpy as np

DEFAULT_CALL_TIMEOUT_S = 3600.0


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def pool_allowed():
    """a pool may be started here: not switched off, and this process is not itself a rank of a torchrun job"""
    if os.environ.get("TGP_DIST_POOL") == "0":
        return False
    return not any(v in os.environ for v in ("RANK", "LOCAL_RANK", "TORCHELASTIC_RUN_ID"))


def default_world():
    env = os.environ.get("TGP_DIST_POOL_WORLD")
    if env:
        return max(int(env), 1)
    from . import _lib
    return max(int(_lib.load_library().tgp_device_count()), 1)


# ---- worker side -----------------------------------------------------------------------------------------------------------
def _attach(name):
    """a shared-memory segment created by the parent (which also unlinks it: keep this process's resource tracker out of it)"""
    from multiprocessing import resource_tracker, shared_memory
    shm = shared_memory.SharedMemory(name=name)
    try:
        resource_tracker.unregister(shm._name, "shared_memory")
    except Exception:        # noqa: BLE001 - best effort: a stray warning at exit is all that is at stake
        pass
    return shm


def _views(buf, layout):
    """{name: ndarray} over one buffer; layout: [(name, shape)] of float64 arrays, back to back"""
    out, off = {}, 0
    for name, shape in layout:
        n = int(np.prod(shape)) if len(shape) else 1
        out[name] = np.ndarray(shape, dtype=np.float64, buffer=buf, offset=off)
        off += 8 * n
    return out


def _layout_bytes(layout):
    return 8 * sum(int(np.prod(s)) if len(s) else 1 for _, s in layout)


def _worker_main(rank, world, port, backend, one_device, conn, env):
    """one rank of the pool: a DistEngine over torch.distributed, serving the parent's calls until told to stop"""
    try:
        os.environ.update(env)
        os.environ.update({"MASTER_ADDR": "127.0.0.1", "MASTER_PORT": str(port), "TGP_DIST_POOL": "0"})
        local = 0 if one_device else rank
        os.environ["TGP_DEVICE"] = str(local)
        import torch
        import torch.distributed as dist
        torch.cuda.set_device(local)
        method = "tcp://127.0.0.1:%d" % port
        if backend == "nccl":
            dist.init_process_group("nccl", init_method=method, rank=rank, world_size=world, device_id=torch.device("cuda", local))
        else:
            dist.init_process_group(backend, init_method=method, rank=rank, world_size=world)
        from . import dist as tdist
        from . import ops
        eng = tdist.enable(min_n=0)
        conn.send(("ready", rank, eng.comm.size))
    except BaseException as e:           # noqa: BLE001 - the parent must hear about it
        conn.send(("error", "worker %d failed to start: %r" % (rank, e)))
        return
    while True:
        try:
            msg = conn.recv()
        except EOFError:
            break
        if msg[0] == "stop":
            break
        try:
            kind, spec_kw, shm_name, layout, opts = msg
            shm = _attach(shm_name)
            try:
                v = _views(shm.buf, layout)
                spec = ops.KernelSpec(**spec_kw)
                if kind == "solve":
                    alpha, logdet, ydota, _ = eng.gp_solve(spec, v["X"], v["y"], v["y_err"] if opts["has_err"] else None,
                                                           keep=False, want_alpha=True)
                    if rank == 0:
                        v["alpha"][:] = alpha
                    reply = ("ok", float(logdet), float(ydota))
                elif kind == "predict":
                    ys = eng.gp_predict(spec, v["X"], v["alpha"], v["Xs"])
                    if rank == 0:
                        v["ys"][:] = ys
                    reply = ("ok",)
                else:
                    reply = ("error", "unknown request %r" % (kind,))
                del v
            finally:
                shm.close()
            conn.send(reply)
        except np.linalg.LinAlgError as e:
            conn.send(("linalg", str(e)))
        except BaseException as e:       # noqa: BLE001 - DistStall, device errors, ...: the parent tears the pool down
            conn.send(("error", "%s: %s" % (type(e).__name__, e)))
            break
    try:
        import torch.distributed as dist
        if dist.is_initialized():
            dist.destroy_process_group()
    except BaseException:                # noqa: BLE001
        pass


# ---- parent side -----------------------------------------------------------------------------------------------------------
class PoolError(RuntimeError):
    """a worker died, stalled or reported an error: the pool has been torn down (the next call starts a fresh one)"""


class WorkerPool(object):
    def __init__(self, world=None, backend=None, one_device=None):
        import multiprocessing as mp
        self.world = int(world) if world else default_world()
        self.backend = backend or os.environ.get("TGP_DIST_BACKEND", "nccl")
        self.one_device = (os.environ.get("TGP_ONE_DEVICE") == "1") if one_device is None else bool(one_device)
        ctx = mp.get_context("spawn")
        port = _free_port()
        env = {k: v for k, v in os.environ.items() if k.startswith("TGP_") and k not in ("TGP_DIST", "TGP_DEVICE")}
        self.procs, self.conns = [], []
        for r in range(self.world):
            a, b = ctx.Pipe()
            p = ctx.Process(target=_worker_main, args=(r, self.world, port, self.backend, self.one_device, b, env), daemon=True)
            p.start()
            b.close()
            self.procs.append(p)
            self.conns.append(a)
        try:
            for r, c in enumerate(self.conns):
                msg = self._recv(c, r, float(os.environ.get("TGP_DIST_POOL_START_S", "600")))
                if msg[0] != "ready":
                    raise PoolError(msg[1] if len(msg) > 1 else "worker %d did not start" % r)
        except BaseException:
            self.close()
            raise
        atexit.register(self.close)

    def _recv(self, conn, rank, timeout):
        if not conn.poll(timeout):
            raise PoolError("worker %d of %d did not answer within %.0f s" % (rank, self.world, timeout))
        try:
            return conn.recv()
        except EOFError:
            raise PoolError("worker %d of %d died" % (rank, self.world))

    def call(self, kind, spec, arrays, outputs, opts=None, timeout=DEFAULT_CALL_TIMEOUT_S):
        """arrays: {name: ndarray} inputs, outputs: {name: shape}; returns ({name: ndarray of the outputs}, rank 0's reply)"""
        from multiprocessing import shared_memory
        layout = [(k, tuple(np.shape(a))) for k, a in arrays.items()] + [(k, tuple(s)) for k, s in outputs.items()]
        shm = shared_memory.SharedMemory(create=True, size=max(_layout_bytes(layout), 8))
        try:
            v = _views(shm.buf, layout)
            for k, a in arrays.items():
                v[k][...] = a
            msg = (kind, dict(kind=spec.kind, amp=spec.amp, a=spec.a, b=spec.b, c=spec.c, ell=spec.ell), shm.name, layout, opts or {})
            try:
                for c in self.conns:
                    c.send(msg)
                replies = [self._recv(c, r, timeout) for r, c in enumerate(self.conns)]
            except (PoolError, OSError, BrokenPipeError) as e:
                self.close()
                raise PoolError("multi-GPU worker pool failed (%s); it has been shut down" % (e,))
            bad = [r for r in replies if r[0] == "error"]
            if bad:
                self.close()
                raise PoolError("multi-GPU worker failed: %s; the pool has been shut down" % (bad[0][1],))
            lin = [r for r in replies if r[0] == "linalg"]
            if lin:
                raise np.linalg.LinAlgError(lin[0][1])
            out = {k: np.array(v[k]) for k in outputs}
            del v
            return out, replies[0]
        finally:
            shm.close()
            shm.unlink()

    def close(self):
        conns, procs = getattr(self, "conns", []), getattr(self, "procs", [])
        self.conns, self.procs = [], []
        for c in conns:
            try:
                c.send(("stop",))
            except Exception:            # noqa: BLE001
                pass
        for p in procs:
            p.join(timeout=5)
            if p.is_alive():
                p.terminate()            # this child only (its own pid), never a pattern
                p.join(timeout=5)
        for c in conns:
            try:
                c.close()
            except Exception:            # noqa: BLE001
                pass

    @property
    def alive(self):
        return bool(self.procs) and all(p.is_alive() for p in self.procs)


class _NoComm(object):
    rank, size = 0, 1


class PoolEngine(object):
    """The parent's end: the interface ``ops`` expects of a multi-GPU engine (``gp_solve`` / ``gp_predict`` / ``min_n``)."""

    def __init__(self, min_n=None, world=None):
        from .dist import DEFAULT_MIN_N
        self.min_n = int(os.environ.get("TGP_DIST_MIN_N", DEFAULT_MIN_N)) if min_n is None else int(min_n)
        self._world = world
        self.pool = None
        self.comm = _NoComm()            # the pair binning of two_pcf stays on the parent's GPU
        self.chain_form = "gather"
        self._warned_keep = False

    def _ensure(self):
        if self.pool is None or not self.pool.alive:
            self.pool = WorkerPool(world=self._world)
        return self.pool

    @property
    def world(self):
        return self._ensure().world

    def gp_solve(self, spec, X, y, y_err=None, keep=False, want_alpha=True):
        from . import _lib, ops
        if keep:
            if not self._warned_keep:
                warnings.warn("kept-factor solves (posterior covariance, several fields, likelihood gradient) are not served by "
                              "the multi-GPU worker pool -- the factor would live in other processes: this solve runs on this "
                              "process's own GPU (run the script under torchrun for the SPMD route, which keeps a replicated "
                              "factor on every rank)", RuntimeWarning, stacklevel=3)
                self._warned_keep = True
            return ops.gp_solve(spec, X, y, y_err, keep=True, want_alpha=want_alpha, ctx=_lib.get_ctx())
        X2, y = _lib.as_xy(X), _lib.f64(y)
        n = X2.shape[0]
        arrays = {"X": X2, "y": y, "y_err": np.zeros(n) if y_err is None else _lib.f64(y_err)}
        out, reply = self._ensure().call("solve", spec, arrays, {"alpha": (n,)}, {"has_err": y_err is not None})
        return (out["alpha"] if want_alpha else None), reply[1], reply[2], None

    def gp_predict(self, spec, X, alpha, Xs):
        from . import _lib
        X2, Xs2 = _lib.as_xy(X), _lib.as_xy(Xs)
        out, _ = self._ensure().call("predict", spec, {"X": X2, "alpha": _lib.f64(alpha), "Xs": Xs2}, {"ys": (Xs2.shape[0],)})
        return out["ys"]

    def close(self):
        if self.pool is not None:
            self.pool.close()
            self.pool = None

"""Test-only helpers for the multi-GPU driver: an in-process communicator for virtual ranks
(threads) and a NumPy stand-in for the per-rank arithmetic (CPU gloo tests)."""
import threading

import numpy as np
import torch

from treegp_amd.dist import BLK, BCAST_ELEMS, block_of, gathered_index, owner, panel_blocks, panel_cmax


class ThreadComm(object):
    """G virtual ranks = G threads of one process sharing one GPU stream."""

    class Shared(object):
        def __init__(self, size):
            self.size = size
            self.slots = [None] * size
            self.barrier = threading.Barrier(size)

    def __init__(self, shared, rank):
        self.sh, self.rank, self.size = shared, rank, shared.size

    def _exchange(self, t):
        if t.is_cuda:
            torch.cuda.synchronize()          # virtual ranks use several streams of one device
        self.sh.slots[self.rank] = t
        self.sh.barrier.wait()
        vals = list(self.sh.slots)
        return vals

    def _done(self, t):
        if t.is_cuda:
            torch.cuda.synchronize()
        self.sh.barrier.wait()

    def broadcast(self, t, src):
        vals = self._exchange(t)
        if self.rank != src:
            t.copy_(vals[src])
        self._done(t)

    def all_reduce_sum(self, t):
        vals = self._exchange(t.clone())
        tot = vals[0].clone()
        for v in vals[1:]:
            tot += v
        self._done(t)
        t.copy_(tot)

    def all_reduce_max(self, t):
        vals = self._exchange(t.clone())
        tot = vals[0].clone()
        for v in vals[1:]:
            tot = torch.maximum(tot, v)
        self._done(t)
        t.copy_(tot)

    def all_gather_start(self, out, inp):
        from treegp_amd.dist import _Done
        self.all_gather(out, inp)
        return _Done(out)

    def all_gather(self, out, inp):
        vals = self._exchange(inp)
        n = inp.numel()
        for r in range(self.size):
            out[r * n:(r + 1) * n].copy_(vals[r])
        self._done(out)


class NumpyLocalOps(object):
    """Same interface as treegp_amd.dist.HipLocalOps, dense NumPy rows on the CPU.  Exists only so
    the orchestration (ownership, collectives, gather indexing) can be tested without a GPU."""

    def __init__(self, Kfull, n, G, g, replicated=False):
        self.n, self.G, self.g = n, G, g
        self.replicated = replicated
        self.Np = (n + BLK - 1) // BLK * BLK
        self.nB = self.Np // BLK
        self.nloc = panel_blocks(0, self.nB, g, G)
        self.cmax0 = panel_cmax(1, self.nB, G)
        Kp = np.eye(self.Np)
        Kp[:n, :n] = Kfull
        self.blocks = [block_of(lb, g, G) for lb in range(self.nloc)]
        self.rows = {b: Kp[b * BLK:(b + 1) * BLK, :].copy() for b in self.blocks}     # full-width rows
        self.bcast_full = torch.zeros(BCAST_ELEMS + 3 * BLK * BLK, dtype=torch.float64)
        self.bcast = self.bcast_full[:BCAST_ELEMS]
        self._info = 0
        self.Lfull = np.zeros((self.Np, self.Np)) if replicated else None       # the replicated factor

    def keep_diag(self, k):
        self.Lfull[k * BLK:(k + 1) * BLK, k * BLK:(k + 1) * BLK] = np.tril(self.bcast[:BLK * BLK].numpy().reshape(BLK, BLK))

    def keeps_done(self):
        return None

    def side_wait_keeps(self, ev):
        pass

    def main_wait_keeps(self):
        pass

    def keep_rows(self, k, gathered, cmax, handle):
        handle.wait()
        P = gathered.numpy()
        for b in range(k + 1, self.nB):
            r, idx = gathered_index(b, k + 1, self.G)
            o = (r * cmax + idx) * BLK * BLK
            self.Lfull[b * BLK:(b + 1) * BLK, k * BLK:(k + 1) * BLK] = P[o:o + BLK * BLK].reshape(BLK, BLK)

    def potrs_full(self, rhs):
        z = np.linalg.solve(self.Lfull, rhs.numpy())
        rhs.copy_(torch.from_numpy(np.linalg.solve(self.Lfull.T, z)))

    def empty(self, n):
        return torch.empty(n, dtype=torch.float64)

    def zeros(self, n):
        return torch.zeros(n, dtype=torch.float64)

    def kbuild(self, *a):
        pass

    # no streams on the CPU: the choreography hooks are no-ops
    def on_side(self):
        import contextlib
        return contextlib.nullcontext()

    def side_wait_main(self):
        pass

    def main_wait_side(self):
        pass

    def factor_diag(self, k):
        blk = self.rows[k][:, k * BLK:(k + 1) * BLK]
        try:
            L = np.linalg.cholesky(np.tril(blk) + np.tril(blk, -1).T)
        except np.linalg.LinAlgError:
            self._info = k * BLK + 1
            L = np.eye(BLK)
        self.rows[k][:, k * BLK:(k + 1) * BLK] = L
        W0 = np.linalg.inv(L[:128, :128]); W1 = np.linalg.inv(L[128:, 128:])
        self.bcast.copy_(torch.from_numpy(np.concatenate([L.ravel(), W0.ravel(), W1.ravel()])))

    def trsm(self, k):
        L = self.bcast[:BLK * BLK].numpy().reshape(BLK, BLK)
        Li = np.linalg.inv(np.tril(L))
        for b in self.blocks:
            if b > k:
                self.rows[b][:, k * BLK:(k + 1) * BLK] = self.rows[b][:, k * BLK:(k + 1) * BLK] @ Li.T

    # panel chain with the panel exchange off it (TGP_DIST_CHAIN_BCAST): same interface as HipLocalOps
    def bcast_payload(self, j):
        return self.bcast_full[:BCAST_ELEMS + j * BLK * BLK]

    def pack_ext(self, b, kgroup):
        for m in range(kgroup, b):
            dst = BCAST_ELEMS + (m - kgroup) * BLK * BLK
            self.bcast_full[dst:dst + BLK * BLK] = torch.from_numpy(self.rows[b][:, m * BLK:(m + 1) * BLK].ravel())

    def strip_left(self, b, kgroup, from_bcast):
        """this rank's rows (blocks >= b) of block b's two tile columns -= sum over the panels kgroup .. b-1, one panel after the
        other -- the products and their order are those of the right-looking strips, so the two forms agree bit for bit"""
        for m in range(kgroup, b):
            if from_bcast:
                o = BCAST_ELEMS + (m - kgroup) * BLK * BLK
                colop = self.bcast_full[o:o + BLK * BLK].numpy().reshape(BLK, BLK)
            else:
                colop = self.rows[b][:, m * BLK:(m + 1) * BLK]
            for bi in self.blocks:
                if bi < b:
                    continue
                for half in (0, 1):
                    c0 = b * BLK + half * 128
                    sl = slice(half * 128, (half + 1) * 128)
                    self.rows[bi][:, c0:c0 + 128] -= sum([self.rows[bi][:, m * BLK:(m + 1) * BLK] @ colop[sl, :].T])

    def panel_send_view(self, k, cmax):
        out = torch.zeros(cmax * BLK * BLK, dtype=torch.float64)
        below = [b for b in self.blocks if b > k]
        for i, b in enumerate(below):
            out[i * BLK * BLK:(i + 1) * BLK * BLK] = torch.from_numpy(self.rows[b][:, k * BLK:(k + 1) * BLK].ravel())
        return out

    def update_group(self, k, bufs, cmaxs, col_lo=0, col_hi=-1, side=False):
        """group k .. k+ns-1: the blocks > k+ns-1 get all ns panels' contributions; columns count from block k+ns"""
        G, ns = self.G, len(bufs)
        Ps = [b.numpy() for b in bufs]
        ncol = 2 * (self.nB - k - ns)
        if col_hi < 0 or col_hi > ncol:
            col_hi = ncol

        def blk(s, b):
            r, idx = gathered_index(b, k + s + 1, G)
            o = (r * cmaxs[s] + idx) * BLK * BLK
            return Ps[s][o:o + BLK * BLK].reshape(BLK, BLK)
        for bi in self.blocks:
            if bi <= k + ns - 1:
                continue
            for bj in range(k + ns, bi + 1):
                for half in (0, 1):
                    tcol = 2 * (bj - k - ns) + half
                    if col_lo <= tcol < col_hi:
                        c0 = bj * BLK + half * 128
                        sl = slice(half * 128, (half + 1) * 128)
                        self.rows[bi][:, c0:c0 + 128] -= sum(blk(s, bi) @ blk(s, bj)[sl, :].T for s in range(ns))

    def update2(self, k, gathered0, cmax0, gathered1, cmax1, col_lo=0, col_hi=-1):
        """pair (k, k+1): the blocks > k+1 get both panels' contributions; columns count from block k+2"""
        G = self.G
        P0, P1 = gathered0.numpy(), gathered1.numpy()
        ncol = 2 * (self.nB - k - 2)
        if col_hi < 0 or col_hi > ncol:
            col_hi = ncol

        def blk(P, cmax, first, b):
            r, idx = gathered_index(b, first, G)
            o = (r * cmax + idx) * BLK * BLK
            return P[o:o + BLK * BLK].reshape(BLK, BLK)
        for bi in self.blocks:
            if bi <= k + 1:
                continue
            A0, A1 = blk(P0, cmax0, k + 1, bi), blk(P1, cmax1, k + 2, bi)
            for bj in range(k + 2, bi + 1):
                B0, B1 = blk(P0, cmax0, k + 1, bj), blk(P1, cmax1, k + 2, bj)
                for half in (0, 1):
                    tcol = 2 * (bj - k - 2) + half
                    if col_lo <= tcol < col_hi:
                        c0 = bj * BLK + half * 128
                        s = slice(half * 128, (half + 1) * 128)
                        self.rows[bi][:, c0:c0 + 128] -= A0 @ B0[s, :].T + A1 @ B1[s, :].T

    def update(self, k, gathered, cmax, col_lo=0, col_hi=-1, side=False):
        G = self.G
        P = gathered.numpy()
        ncol = 2 * (self.nB - k - 1)
        if col_hi < 0 or col_hi > ncol:
            col_hi = ncol

        def blk(b):
            r, idx = gathered_index(b, k + 1, G)
            o = (r * cmax + idx) * BLK * BLK
            return P[o:o + BLK * BLK].reshape(BLK, BLK)
        for bi in self.blocks:
            if bi <= k:
                continue
            Pi = blk(bi)
            for bj in range(k + 1, bi + 1):
                for half in (0, 1):                       # 128-tile columns of block bj
                    tcol = 2 * (bj - k - 1) + half
                    if col_lo <= tcol < col_hi:
                        c0 = bj * BLK + half * 128
                        self.rows[bi][:, c0:c0 + 128] -= Pi @ blk(bj)[half * 128:(half + 1) * 128, :].T

    def info(self):
        i, self._info = self._info, 0
        return i

    # -- replicated finish: the layout of a rank's share from panel k0 on (panels in order, own blocks >= p in order) ----
    def _tail_blocks(self, k0, r):
        return [(p, b) for p in range(k0, self.nB) for b in range(p, self.nB) if owner(b, self.G) == r]

    def tail_elems(self, k0, r):
        return len(self._tail_blocks(k0, r)) * BLK * BLK

    def tail_send_view(self, k0, stride):
        out = torch.zeros(stride, dtype=torch.float64)
        for i, (p, b) in enumerate(self._tail_blocks(k0, self.g)):
            out[i * BLK * BLK:(i + 1) * BLK * BLK] = torch.from_numpy(self.rows[b][:, p * BLK:(p + 1) * BLK].ravel())
        return out

    def tail_finish(self, k0, gathered, stride):
        m = self.Np - BLK * k0
        T = np.zeros((m, m))
        P = gathered.numpy()
        for r in range(self.G):
            for i, (p, b) in enumerate(self._tail_blocks(k0, r)):
                o = r * stride + i * BLK * BLK
                T[(b - k0) * BLK:(b - k0 + 1) * BLK, (p - k0) * BLK:(p - k0 + 1) * BLK] = P[o:o + BLK * BLK].reshape(BLK, BLK)
        T = np.tril(T) + np.tril(T, -1).T
        try:
            L = np.linalg.cholesky(T)
        except np.linalg.LinAlgError:
            return BLK * k0 + 1
        for b in self.blocks:
            if b >= k0:
                self.rows[b][:, k0 * BLK:] = L[(b - k0) * BLK:(b - k0 + 1) * BLK, :]
        if self.Lfull is not None:
            self.Lfull[k0 * BLK:, k0 * BLK:] = L
        return 0

    def fwd_diag(self, k, yk):
        L = np.tril(self.rows[k][:, k * BLK:(k + 1) * BLK])
        yk.copy_(torch.from_numpy(np.linalg.solve(L, yk.numpy())))

    def fwd_update(self, k, zk, yloc):
        z = zk.numpy()
        for lb, b in enumerate(self.blocks):
            if b > k:
                yloc[lb * BLK:(lb + 1) * BLK] -= torch.from_numpy(self.rows[b][:, k * BLK:(k + 1) * BLK] @ z)

    def bwd_partial(self, k, aloc, s):
        acc = np.zeros(BLK)
        for lb, b in enumerate(self.blocks):
            if b > k:
                acc += self.rows[b][:, k * BLK:(k + 1) * BLK].T @ aloc[lb * BLK:(lb + 1) * BLK].numpy()
        s.copy_(torch.from_numpy(acc))

    def bwd_diag(self, k, ak, s=None):
        L = np.tril(self.rows[k][:, k * BLK:(k + 1) * BLK])
        rhs = ak.numpy() - (0 if s is None else s.numpy())
        ak.copy_(torch.from_numpy(np.linalg.solve(L.T, rhs)))

    def logdet_local(self, out):
        s = 0.0
        for b in self.blocks:
            d = np.diag(self.rows[b][:, b * BLK:(b + 1) * BLK])
            for r in range(BLK):
                if b * BLK + r < self.n:
                    s += 2.0 * np.log(d[r])
        out[0] = s


def numpy_kk_partial(bin_type, x, y, k, w, min_sep, max_sep, nbins, part, nparts, ctx=None):
    """CPU stand-in for ops.kk_partial (tgp_kk_partial): the pairs (i, j > i) whose 256-point i-tile is
    dealt to `part`, binned by the oracle's rules; raw sums (3, nbins^2) TwoD / (5, nbins) Log."""
    x = np.asarray(x, float); y = np.asarray(y, float); k = np.asarray(k, float)
    n = len(x)
    w = np.ones(n) if w is None else np.asarray(w, float)
    acc = np.zeros((3, nbins * nbins) if bin_type == 0 else (5, nbins))
    for s in range(part * 256, n, nparts * 256):
        e = min(n, s + 256)
        dx = x[None, :] - x[s:e, None]
        dy = y[None, :] - y[s:e, None]
        rsq = dx * dx + dy * dy
        later = np.arange(n)[None, :] > np.arange(s, e)[:, None]
        ww = w[s:e, None] * w[None, :]
        kk = k[s:e, None] * k[None, :]
        if bin_type == 0:
            bs = 2.0 * max_sep / nbins
            ok = later & (rsq != 0.0) & (rsq >= min_sep * min_sep) & (np.maximum(np.abs(dx), np.abs(dy)) < max_sep)
            for sgn in (1.0, -1.0):
                ix = ((sgn * dx[ok] + max_sep) / bs).astype(np.int64)
                iy = ((sgn * dy[ok] + max_sep) / bs).astype(np.int64)
                good = (ix >= 0) & (ix < nbins) & (iy >= 0) & (iy < nbins)
                b = iy[good] * nbins + ix[good]
                acc[0] += np.bincount(b, weights=(ww * kk)[ok][good], minlength=nbins * nbins)
                acc[1] += np.bincount(b, weights=ww[ok][good], minlength=nbins * nbins)
                acc[2] += np.bincount(b, minlength=nbins * nbins)
        else:
            bs = np.log(max_sep / min_sep) / nbins
            ok = later & (rsq >= min_sep * min_sep) & (rsq < max_sep * max_sep)
            lr = 0.5 * np.log(rsq[ok])
            b = ((lr - np.log(min_sep)) / bs).astype(np.int64)
            good = (b >= 0) & (b < nbins)
            b = b[good]
            wg = ww[ok][good]
            acc[0] += np.bincount(b, weights=wg * kk[ok][good], minlength=nbins)
            acc[1] += np.bincount(b, weights=wg, minlength=nbins)
            acc[2] += np.bincount(b, weights=wg * np.sqrt(rsq[ok][good]), minlength=nbins)
            acc[3] += np.bincount(b, weights=wg * lr[good], minlength=nbins)
            acc[4] += np.bincount(b, minlength=nbins)
    return acc

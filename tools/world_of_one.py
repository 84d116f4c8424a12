#!/usr/bin/env python3
"""The multi-GPU driver on ONE GPU with a world of one rank (SelfComm): phase times and the rate of its bulk update
kernel, to compare with the single-GPU path.  usage: world_of_one.py [N=65536] [steps=2]
env: TGP_DIST_REPLICATE=1 to include the replicated-factor copies."""
import os
import sys
import time

import torch

sys.path.insert(0, __file__.rsplit("/tools/", 1)[0])
from treegp_amd import _lib, ops  # noqa: E402
from treegp_amd.dist import DistributedGP, SelfComm  # noqa: E402
from treegp_amd.synthetic import star_field, headline_invlam  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 2
iL = headline_invlam()
spec = ops.KernelSpec(_lib.TGP_ARBF, amp=1.0, a=iL[0, 0], b=iL[0, 1], c=iL[1, 1])
X, y, ye, Xs = star_field(n, 4096)
gp = DistributedGP(_lib.new_ctx(0), spec, X, y - y.mean(), ye, Xs, comm=SelfComm(), device=torch.device("cuda", 0), profile=True)
gp.step()
torch.cuda.synchronize()
acc = {}
t0 = time.perf_counter()
for _ in range(steps):
    gp.step(acc)
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / steps
print("N=%d world of one: %.1f ms per step; phases %s" % (n, dt * 1e3, {k: round(v / steps, 2) for k, v in acc.items() if k.endswith("_ms")}))
print("bulk update: %.2f TF over %d launches per step; Cholesky %.2f TF" %
      (acc["syrk_flops"] / acc["syrk_ms"] / 1e9, acc["syrk_launches"] / steps, n ** 3 / 3.0 / (acc["chol_ms"] / steps) / 1e9))

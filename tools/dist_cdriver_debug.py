#!/usr/bin/env python3
"""debug aid: per-rank comparison of the local matrices left by gptorch_amd.dist.BlockCyclicGP (Python
orchestration) and by gpn_dist_lml_forward (C driver) -- same layout, so they must agree tile by tile.
torchrun --nproc-per-node 4 tools/dist_cdriver_debug.py (GPN_SHARED_GPU=1)"""
import os, sys
import torch
import torch.distributed as dist
ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, ROOT)
from gptorch_amd import dist as gdist, rng  # noqa: E402
n, d, T = int(os.environ.get("DN", 2048)), 8, int(os.environ.get("DT", 512))
torch.cuda.set_device(0)
dev = torch.device("cuda", 0)
dist.init_process_group("gloo")
x, y = rng.make_regression(n, d, 1, seed=0)
X, Y = torch.tensor(x, device=dev), torch.tensor(y, device=dev)
one = torch.ones(1, dtype=torch.float64, device=dev)
ls = one * float(d) ** 0.5
g = gdist.BlockCyclicGP(X, Y, "Rbf", tile=T)
lml_p = g.log_likelihood(one, ls, 0.01 * one, Y)
c = gdist.NativeDistLML(X, Y, "Rbf", tile=T, comm="torch")
lml_c = c.log_likelihood(one, ls, 0.01 * one)
rows, ld = g.A.shape
Ac = c.work[:rows * ld].view(rows, ld)
msg = ["rank %d (%d,%d) lml py %.6f c %.6f" % (g.rank, g.my_r, g.my_c, lml_p.item(), lml_c.item())]
for li in range(rows // T + 1):
    for lj in range(ld // T):
        a, b = g.A[li * T:(li + 1) * T, lj * T:(lj + 1) * T], Ac[li * T:(li + 1) * T, lj * T:(lj + 1) * T]
        if a.numel():
            diff = (a - b).abs().max().item()
            if diff > 1e-9:
                msg.append("   local tile (%d,%d) max diff %.3e" % (li, lj, diff))
for r in range(dist.get_world_size()):
    dist.barrier()
    if r == g.rank:
        print("\n".join(msg), flush=True)
dist.destroy_process_group()

#!/usr/bin/env python3
"""VFE with the rows sharded over ranks (sparse_gpr.SHARD_GROUP): every rank must report the bound and
the gradients of the whole data set.  Single process: the unsharded reference values.
  python tools/vfe_shard_check.py                       # reference
  GPN_SHARED_GPU=1 python -m torch.distributed.run --nproc-per-node 2 ... tools/vfe_shard_check.py"""
import os, sys
import numpy as np
import torch
import torch.distributed as dist
ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, ROOT)
from gptorch_amd import kernels, likelihoods, mean_functions, rng  # noqa: E402
from gptorch_amd.models import VFE, sparse_gpr  # noqa: E402

world = int(os.environ.get("WORLD_SIZE", "1"))
rank = int(os.environ.get("RANK", "0"))
local = 0 if os.environ.get("GPN_SHARED_GPU") == "1" else int(os.environ.get("LOCAL_RANK", "0"))
torch.cuda.set_device(local)
if world > 1:
    dist.init_process_group("gloo" if os.environ.get("GPN_SHARED_GPU") == "1" else "nccl")
    sparse_gpr.SHARD_GROUP = True
n, d, dy, m = 3001, 3, 2, 150
x, y = rng.make_regression(n, d, dy, seed=0)
z = rng.normal(57, (m, d))
lo, hi = (n * rank) // world, (n * (rank + 1)) // world        # this rank's row shard
sparse_gpr.CHUNK_ROWS = 512
mod = VFE(x[lo:hi], y[lo:hi], kernels.Rbf(d, variance=0.8, length_scales=np.array([0.5, 0.7, 0.9]), ARD=True),
          inducing_points=z, likelihood=likelihoods.Gaussian(variance=0.1), mean_function=mean_functions.Zero(dy))
mod.cuda()
loss = mod.loss()
loss.backward()
g = [mod.kernel.variance.grad, mod.kernel.length_scales.grad, mod.likelihood.variance.grad, mod.Z.grad]
line = "rank %d: loss=%.10f grads=%s zsum=%.10e zabs=%.10e" % (
    rank, loss.item(), " ".join("%.10e" % v for t in g[:3] for v in t.flatten().tolist()), g[3].sum().item(), g[3].abs().sum().item())
if world > 1:                    # one writer: two ranks printing at once interleave inside a line
    lines = [None] * world
    dist.all_gather_object(lines, line)
    if rank == 0:
        print("\n".join(lines), flush=True)
else:
    print(line, flush=True)
if world > 1:
    dist.destroy_process_group()

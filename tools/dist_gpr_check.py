#!/usr/bin/env python3
"""GPU-box tool / test driver: gptorch_amd.models.DistGPR (GPR's call surface over the block-cyclic engine,
native tile ops) against the single-GPU GPR on the same data -- loss, raw-parameter gradients, predictions.
    python -m torch.distributed.run --nproc-per-node 2 --master-addr 127.0.0.1 tools/dist_gpr_check.py [n d tile]
GPN_SHARED_GPU=1: every rank on cuda:0 over gloo (1-GPU box); otherwise one GPU per rank over nccl."""
import os, sys
import numpy as np
import torch
import torch.distributed as dist
ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, ROOT)
from gptorch_amd import kernels, likelihoods, rng  # noqa: E402
from gptorch_amd.models import GPR, DistGPR  # noqa: E402

n, d, T = (int(a) for a in (sys.argv[1:4] + ["3000", "4", "512"][len(sys.argv) - 1:]))
shared = os.environ.get("GPN_SHARED_GPU") == "1"
local = 0 if shared else int(os.environ.get("LOCAL_RANK", "0"))
torch.cuda.set_device(local)
dist.init_process_group("gloo" if shared else "nccl", **({} if shared else {"device_id": torch.device("cuda", local)}))
x, y = rng.make_regression(n, d, 2, seed=0)
xs = rng.normal(9, (11, d))


def build(cls, **kw):
    m = cls(x, y, kernels.Matern52(d, variance=1.3, length_scales=1.1 + 0.3 * np.arange(d), ARD=True),
            likelihood=likelihoods.Gaussian(variance=0.05), **kw)
    m.cuda()
    return m


m = build(DistGPR, tile=T)
loss = m.loss()
loss.backward()
mu, var = m.predict_f(xs)
_, cov = m.predict_y(xs, diag=False)
if dist.get_rank() == 0:
    r = build(GPR)
    lr = r.loss()
    lr.backward()
    rmu, rvar = r.predict_f(xs)
    _, rcov = r.predict_y(xs, diag=False)
    errs = {"loss": abs(loss.item() - lr.item()) / abs(lr.item())}
    for nm, a, b in [("g_variance", m.kernel.variance.grad, r.kernel.variance.grad), ("g_length_scales", m.kernel.length_scales.grad, r.kernel.length_scales.grad),
                     ("g_noise", m.likelihood.variance.grad, r.likelihood.variance.grad)]:
        errs[nm] = ((a - b).abs().max() / b.abs().max().clamp(min=1.0)).item()
    errs["mean"] = float(np.abs(mu - rmu).max())
    errs["var"] = float(np.abs(var - rvar).max())
    errs["cov"] = float(np.abs(cov - rcov).max())
    print("dist_gpr_check world=%d grid=%dx%d: " % (dist.get_world_size(), m._engine.pr, m._engine.pc) +
          " ".join("%s=%.2e" % kv for kv in errs.items()), flush=True)
dist.barrier()
dist.destroy_process_group()

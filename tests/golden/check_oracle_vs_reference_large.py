#!/usr/bin/env python3
"""Build-container check (imports the reference from /root/reference; never runs on the GPU box): the CPU oracle against the REFERENCE
ITSELF at the largest sizes this container holds, for the quantities whose full-size goldens come from the oracle evaluated on the GPU
boxes' hosts (tests/sweeps/*_cpu_parity.py): the VFE bound at N = 100000, M = 1024 and the GPR loss + autograd gradients at C2's
size.  Same ATen kernels, same op sequence: the values agree bit for bit (or to the last ulp), which is what makes the oracle's
full-size values the reference's values."""
import sys
import time

import numpy as np
import torch

sys.path.insert(0, "/root/repo")
sys.path.insert(0, "/root/reference")
torch.set_default_dtype(torch.float64)
from oracle import gp_oracle as orc  # noqa: E402
from gptorch_amd import rng  # noqa: E402
from gptorch import kernels as rk, likelihoods as rl, mean_functions as rmf  # noqa: E402
from gptorch.models.gpr import GPR as RefGPR  # noqa: E402
from gptorch.models.sparse_gpr import VFE as RefVFE  # noqa: E402

n, m, d = 100000, 1024, 8
x, y = rng.make_regression(n, d, 1, seed=0)
z = rng.normal(99, (m, d))
o = orc.VFEOracle(x, y, z, kind="Rbf", variance=1.0, length_scales=float(np.sqrt(d)), noise=1e-2)
with torch.no_grad():
    vo = o.log_likelihood().item()
ref = RefVFE(x, y, rk.Rbf(d, variance=1.0, length_scales=float(np.sqrt(d))), inducing_points=z.copy(), likelihood=rl.Gaussian(variance=1e-2),
             mean_function=rmf.Zero(1))
with torch.no_grad():
    vr = ref.log_likelihood().item()
print("VFE N=%d M=%d: oracle %.10f  reference %.10f  rel diff %.1e" % (n, m, vo, vr, abs(vo - vr) / abs(vr)))

n, d = 8192, 8
x, y = rng.make_regression(n, d, 1, seed=0)
o = orc.GPROracle(x, y, kind="Rbf", variance=1.0, length_scales=float(np.sqrt(d)), noise=1e-2)
lo, go = o.loss_and_grads()
r = RefGPR(x, y, rk.Rbf(d, variance=1.0, length_scales=float(np.sqrt(d))), likelihood=rl.Gaussian(variance=1e-2))
lr = r.loss()
lr.backward()
gr = [r.kernel.variance.grad, r.kernel.length_scales.grad, r.likelihood.variance.grad]
print("GPR N=%d: loss oracle %.10f reference %.10f (diff %.1e); gradients rel diff %s" % (
    n, lo.item(), lr.item(), abs(lo.item() - lr.item()), ["%.1e" % (abs(a.item() - b.item()) / abs(b.item())) for a, b in zip(go, gr)]))

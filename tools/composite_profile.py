#!/usr/bin/env python3
"""GPU-box tool: loss() + backward() of GPR over the reference's example kernel Linear + Rbf + Constant
(examples/regression_1d.py:34-53) at N = 8192, D = 4, for rocprofv3 --kernel-trace --stats: on the fused path
(gptorch_amd/_expr.py) no at::native elementwise kernel runs over an N x N tensor -- the only N x N passes are the
expression assembly (kexpr_kernel), the factorisation / inversion contractions and one expression sweep per leaf
(kexpr_grad_kernel)."""
import os, sys
import numpy as np
import torch
ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, ROOT)
from gptorch_amd import kernels, likelihoods, rng  # noqa: E402
from gptorch_amd.models import GPR  # noqa: E402
n, d = int(sys.argv[1]) if len(sys.argv) > 1 else 8192, 4
x, y = rng.make_regression(n, d, 1, seed=0)
m = GPR(x, y, kernels.Linear(d, variance=0.1) + kernels.Rbf(d, variance=1.0, length_scales=2.0) + kernels.Constant(d, variance=0.5),
        likelihood=likelihoods.Gaussian(variance=1e-2))
m.cuda()
for _ in range(3):
    m.zero_grad()
    loss = m.loss()
    loss.backward()
torch.cuda.synchronize()
print(loss.item(), type(m.log_likelihood().grad_fn).__name__)

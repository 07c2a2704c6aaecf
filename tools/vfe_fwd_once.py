#!/usr/bin/env python3
"""GPU-box tool: ONE forward evaluation of the VFE bound at config 5 after a warm-up (for rocprofv3 traces)."""
import os, sys, time
import numpy as np, torch
ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, ROOT)
from gptorch_amd import kernels, likelihoods, mean_functions, rng
from gptorch_amd.models import VFE, sparse_gpr
sparse_gpr.LANES = int(os.environ.get('VFE_LANES', '2'))
sparse_gpr.SYRK_K_SLICE = int(os.environ.get('VFE_KSLICE', '8192'))
n, m, d = 1000000, 4096, 8
x, y = rng.make_regression(n, d, 1, seed=0)
z = rng.normal(99, (m, d))
model = VFE(x, y, kernels.Rbf(d, variance=1.0, length_scales=float(np.sqrt(d))), inducing_points=z,
            likelihood=likelihoods.Gaussian(variance=1e-2), mean_function=mean_functions.Zero(1))
model.cuda()
with torch.no_grad():
    for i in range(2):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        model.log_likelihood()
        torch.cuda.synchronize()
        print("forward %d: %.3f s" % (i, time.perf_counter() - t0), flush=True)

"""GPU-box tool: VFE config-5 forward time against the split-K / K-slice settings of the A A^T
accumulation (sparse_gpr.SPLIT_K, SYRK_K_SLICE), same process = same box."""
import os, sys, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np, torch
from gptorch_amd import kernels, likelihoods, mean_functions, rng
from gptorch_amd.models import VFE, sparse_gpr
n, m, d = 1000000, 4096, 8
x, y = rng.make_regression(n, d, 1, seed=0)
z = rng.normal(99, (m, d))
model = VFE(x, y, kernels.Rbf(d, variance=1.0, length_scales=float(np.sqrt(d))), inducing_points=z,
            likelihood=likelihoods.Gaussian(variance=1e-2), mean_function=mean_functions.Zero(1))
model.cuda()
with torch.no_grad():
    model.log_likelihood(); torch.cuda.synchronize()
    for rnd in range(2):
        for split, ks in ((1, 0), (1, 8192), (8, 8192), (4, 8192), (16, 8192)):
            sparse_gpr.SPLIT_K, sparse_gpr.SYRK_K_SLICE = split, ks
            model.log_likelihood(); torch.cuda.synchronize(); t0 = time.perf_counter()
            for _ in range(2): v = model.log_likelihood()
            torch.cuda.synchronize()
            print("split-K %2d, K slice %6d: %.3f s  elbo %.6f" % (split, ks, (time.perf_counter() - t0) / 2, v.item()), flush=True)

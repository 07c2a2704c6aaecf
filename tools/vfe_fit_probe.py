#!/usr/bin/env python3
"""GPU-box tool: a multi-start fit of B sparse (VFE) restarts -- every model's own optimize() one after the other against
multi_start_optimize (one lock-step evaluation + one optimiser call per iteration); checks that the trajectories are equal bit
for bit.  Usage: vfe_fit_probe.py"""
import contextlib, io, os, sys, time
sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), "..")))
import numpy as np, torch
from tests.test_gpu_vfe_lockstep import _models
from gptorch_amd.models import multi_start_optimize
for (B, n, m, d) in ((64, 512, 64, 2), (8, 8192, 512, 8)):
    a, b = _models(B, n, m, d, 1, "Matern52", seed=7), _models(B, n, m, d, 1, "Matern52", seed=7)
    with contextlib.redirect_stdout(io.StringIO()):
        a[0].optimize(method="Adam", max_iter=3, learning_rate=0.01)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        own = [mdl.optimize(method="Adam", max_iter=20, learning_rate=0.01)[0] for mdl in a[1:]]
        torch.cuda.synchronize(); t1 = time.perf_counter()
        multi_start_optimize(_models(2, n, m, d, 1, "Matern52", seed=9), method="Adam", max_iter=3, learning_rate=0.01)
        torch.cuda.synchronize(); t2 = time.perf_counter()
        losses, _ = multi_start_optimize(b[1:], method="Adam", max_iter=20, learning_rate=0.01)
        torch.cuda.synchronize(); t3 = time.perf_counter()
    same = all(np.array_equal(np.asarray(own[i]), losses[i]) for i in range(B - 1))
    print("B %d n %d m %d: own optimize() %.1f ms/iter, multi_start %.1f ms/iter (%.1fx), trajectories bitwise equal: %s"
          % (B - 1, n, m, 1e3 * (t1 - t0) / 20, 1e3 * (t3 - t2) / 20, (t1 - t0) / (t3 - t2), same))

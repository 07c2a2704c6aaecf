#!/usr/bin/env python3
"""Extended-precision value of the loss of the reference's example model (Linear + Rbf + Constant, examples/regression_1d.py:34-53)
at N = 16384, D = 8 -- run in the build container (CPU, a few minutes, ~10 GB); output committed as
tests/golden/composite_16k_extended.json.

Why: at N = 16384 this Kyy (a constant 0.4 * 1 1^T and a rank-8 linear part on top of the Rbf, noise 0.01) has a condition number of
a few 1e6, and the reference's own fp64 value (composite_16k_case.json, MKL on 8 threads) and the native refined value differ by
2.8e-8 -- more than north_star's 1e-8.  As for C3 (make_c3_extended.py) this pins the value both approximate: K assembled by the
oracle's op sequence (= the reference's), y^T Kyy^-1 y by iterative refinement of the fp64 Cholesky solve with the residual
accumulated in 80-bit long double, log det from the fp64 factor."""
import json
import math
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), "..", ".."))
sys.path.insert(0, ROOT)
from gptorch_amd import rng  # noqa: E402
from oracle import gp_oracle as orc  # noqa: E402

n, d, noise = 16384, 8, 0.01
torch.set_num_threads(os.cpu_count() or 8)
x, y = rng.make_regression(n, d, 1, seed=0)
t0 = time.time()
with torch.no_grad():
    X = torch.tensor(x)
    K = orc.linear_K(X, None, torch.full((d,), 0.3, dtype=torch.float64)) \
        + orc.kernel_K("Rbf", X, None, torch.tensor([1.2], dtype=torch.float64), torch.tensor([math.sqrt(d)], dtype=torch.float64)) + 0.4
    K.diagonal().add_(noise)
    L = torch.linalg.cholesky(K)
    Y = torch.tensor(y)
    alpha = torch.linalg.solve_triangular(L, Y, upper=False)
    quad_fp64 = alpha.pow(2).sum().item()
    logdet_half = L.diagonal().log().sum().item()
    a = torch.cholesky_solve(Y, L)
    Kn = K.numpy()
    yl = y[:, 0].astype(np.longdouble)
    al = a.numpy()[:, 0].astype(np.longdouble)
    quads = []
    for it in range(3):
        r = np.empty(n, dtype=np.longdouble)
        for c0 in range(0, n, 1024):
            r[c0:c0 + 1024] = yl[c0:c0 + 1024] - Kn[c0:c0 + 1024].astype(np.longdouble) @ al
        al = al + torch.cholesky_solve(torch.tensor(r.astype(np.float64))[:, None], L).numpy()[:, 0].astype(np.longdouble)
        quads.append(np.dot(yl, al))
const = -0.5 * n * math.log(2.0 * math.pi)
lml = float(-np.longdouble(0.5) * quads[-1] - np.longdouble(logdet_half) + np.longdouble(const))
ref = json.load(open(os.path.join(ROOT, "tests", "golden", "composite_16k_case.json")))
out = {"name": "linear_plus_rbf_plus_constant_16384_8", "quad_extended": repr(quads[-1]), "quad_refinement_steps": [repr(q) for q in quads],
       "quad_fp64_cpu": quad_fp64, "logdet_half_fp64_cpu": logdet_half, "loss_extended": -lml, "loss_fp64_cpu_this_script": 0.5 * quad_fp64 + logdet_half - const,
       "reference_loss": ref["loss"], "reference_abs_err_vs_extended": abs(ref["loss"] + lml), "seconds": time.time() - t0}
json.dump(out, open(os.path.join(ROOT, "tests", "golden", "composite_16k_extended.json"), "w"), indent=1)
print(json.dumps(out))

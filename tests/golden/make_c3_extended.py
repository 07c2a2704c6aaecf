#!/usr/bin/env python3
"""Extended-precision value of the C3 log marginal likelihood (N = 32768, D = 16, Matern52) -- run in
the build container (CPU, ~4 min, ~20 GB), output committed as tests/golden/lml_c3_extended.json.

Why: at this size north_star's 1e-8 ABSOLUTE tolerance is 7e-14 relative, which is the rounding
level of any fp64 factorisation (the quadratic form y^T K^-1 y = 3.2e5 reacts to a backward error E
of the factor through -a^T E a with |a|^2 = 9.5e6).  The reference's own fp64 value and ours both
carry such an error, with opposite signs; this script pins the value both are approximating:

  * K = the kernel matrix by DIRECT differences (what the native assembly kernel computes; the
    reference's Gram-trick K gives 1.2e-9 less in y^T K^-1 y, also recorded);
  * y^T K^-1 y by iterative refinement of the fp64 Cholesky solve with the residual y - K a
    accumulated in 80-bit long double (converges in one step; two are run);
  * log det from the fp64 factor (reference path and native path agree on it to 4e-11).

Usage: python tests/golden/make_c3_extended.py [--gram]   (--gram: the reference's own K instead)
"""
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

n, d, ell, noise = 32768, 16, 4.0, 1e-2
gram = "--gram" in sys.argv
torch.set_num_threads(os.cpu_count() or 8)
x, y = rng.make_regression(n, d, 1, seed=0)
t0 = time.time()
with torch.no_grad():
    if gram:
        from oracle import gp_oracle as orc
        K = orc.GPROracle(x, y, kind="Matern52", variance=1.0, length_scales=ell, noise=noise).compute_kyy(torch.tensor(x))
    else:
        xs = torch.tensor(x * (1.0 / ell))
        K = torch.empty(n, n, dtype=torch.float64)
        s5 = 2.23606797749978969641
        for c0 in range(0, n, 512):
            blk = xs[c0:c0 + 512]
            r2 = torch.zeros(blk.shape[0], n, dtype=torch.float64)
            for dd in range(d):
                df = blk[:, dd:dd + 1] - xs[:, dd][None, :]
                r2 += df * df
            r = torch.sqrt(torch.clamp(r2, min=1e-40))
            K[c0:c0 + 512] = (1.0 + s5 * r + (5.0 / 3.0) * r * r) * torch.exp(-s5 * r)
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
    for it in range(2):
        r = np.empty(n, dtype=np.longdouble)
        for c0 in range(0, n, 1024):
            r[c0:c0 + 1024] = yl[c0:c0 + 1024] - Kn[c0:c0 + 1024].astype(np.longdouble) @ al
        al = al + torch.cholesky_solve(torch.tensor(r.astype(np.float64))[:, None], L).numpy()[:, 0].astype(np.longdouble)
        quads.append(np.dot(yl, al))
const = -0.5 * n * math.log(2.0 * math.pi)
lml = float(-np.longdouble(0.5) * quads[-1] - np.longdouble(logdet_half) + np.longdouble(const))
print(json.dumps({"name": "C3_m52_32768_16", "K": "gram-trick (reference)" if gram else "direct differences (native)",
                  "quad_extended": repr(quads[-1]), "quad_refinement_steps": [repr(q) for q in quads], "quad_fp64_cpu": quad_fp64,
                  "logdet_half_fp64_cpu": logdet_half, "lml_extended": lml, "seconds": time.time() - t0}))

#!/usr/bin/env python3
"""GPU-box tool: backward error of the native Cholesky vs rocSOLVER (torch.linalg) on the C2 Gram matrix."""
import os, sys, math
import torch
ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, ROOT)
from gptorch_amd import rng, _ops, functions  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
dev = torch.device("cuda:0")
x, y = rng.make_regression(n, 8, 1, seed=0)
X, Y = torch.tensor(x, device=dev), torch.tensor(y, device=dev)
one = torch.ones(1, dtype=torch.float64, device=dev)
K = _ops.kernel_matrix("Rbf", X, None, one, one * math.sqrt(8.0), noise=one * 1e-2)
nk = K.norm().item()
for name, fn in [("rocsolver", lambda: torch.linalg.cholesky(K)), ("native", lambda: functions.cholesky(K))]:
    L = fn()
    res = (L @ L.t() - K).norm().item() / nk
    a = torch.linalg.solve_triangular(L, Y, upper=False)
    lml = (-0.5 * a.pow(2).sum() - L.diagonal().log().sum() - 0.5 * n * math.log(2 * math.pi)).item()
    print("%-10s  ||LL^T-K||/||K|| = %.3e   lml(torch trsm on this L) = %.10f" % (name, res, lml))
f = _ops.kernel_factor("Rbf", X, one, one * math.sqrt(8.0), one * 1e-2, R=Y)
print("native fused lml = %.10f   (golden -90285.1725861576 at n=8192)" % f.lml_terms()[2].item())

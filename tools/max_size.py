#!/usr/bin/env python3
"""GPU-box tool: the largest exact GP one 288 GB MI355X holds comfortably.
usage: max_size.py [N=131072] [D=8]   -- one LML evaluation (K assembly + factorisation + solve +
log-det) with a 8*N^2-byte factor (137 GB at N = 131072), timed, plus the sampled checks of
tests/test_gpu_parity.py::test_c4_full_size_factor_properties (L L^T = K, L a = y)."""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, ROOT)
from gptorch_amd import _ops, rng  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 131072
d = int(sys.argv[2]) if len(sys.argv) > 2 else 8
dev = torch.device("cuda:0")
var, ls, noise = 1.0, float(np.sqrt(d)), 1e-2
xh, yh = rng.make_regression(n, d, 1, seed=0)
x, R = torch.tensor(xh, device=dev), torch.tensor(yh, device=dev)
t = lambda v: torch.tensor([v], dtype=torch.float64, device=dev)
torch.cuda.synchronize()
t0 = time.perf_counter()
f = _ops.kernel_factor("Rbf", x, t(var), t(ls), t(noise), R=R)
terms = f.lml_terms().cpu().numpy()
sec = time.perf_counter() - t0
print("N=%d D=%d: factor buffer %.1f GB, one LML evaluation %.2f s (%.1f TFLOP/s on N^3/3), info=%d, LML=%.6f" % (
    n, d, f.A.numel() * 8 / 1e9, sec, n ** 3 / 3.0 / sec / 1e12, int(f.info.item()), terms[2]), flush=True)
rs = np.random.RandomState(11)
rows = np.unique(np.concatenate([[0, 127, 128, 1024, n - 1], rs.randint(0, n, size=24)]))
a = f.extra()[0]
wk = ws = 0.0
for i in map(int, rows):
    Li = f.A[i, :i + 1]
    ws = max(ws, abs((Li * a[:i + 1]).sum().item() - yh[i, 0]))
    for j in {0, i, max(i - 1, 0), i // 2, int(rs.randint(0, i + 1))}:
        got = (Li[:j + 1] * f.A[j, :j + 1]).sum().item()
        want = var * np.exp(-0.5 * np.sum((xh[i] - xh[j]) ** 2) / ls ** 2) + (noise if i == j else 0.0)
        wk = max(wk, abs(got - want))
print("max |(L L^T - K)_ij| over %d sampled entries: %.2e; max |(L a - y)_i|: %.2e; peak HBM %.1f GB" % (
    5 * len(rows), wk, ws, torch.cuda.max_memory_allocated() / 1e9))
# round 3: the refinement step of the quadratic form at this size (back-substitution over N/128 blocks + symmetric residual pass)
f2, terms2 = _ops.lml_forward("Rbf", x, R, t(var), t(ls), t(noise), factor=f, refine=True)
torch.cuda.synchronize()
t0 = time.perf_counter()
f2, terms2 = _ops.lml_forward("Rbf", x, R, t(var), t(ls), t(noise), factor=f, refine=True)
torch.cuda.synchronize()
sec2 = time.perf_counter() - t0
tr = terms2.cpu().numpy()
print("with gpn_lml_refine: %.2f s, LML %.6f (plain %.6f, difference %.2e), quadratic form %.9e vs %.9e" % (
    sec2, tr[2], terms[2], tr[2] - terms[2], tr[1], terms[1]), flush=True)

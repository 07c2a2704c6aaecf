#!/usr/bin/env python3
"""GPU-box sweep (round 5): lock-step forward + backward (gpn_lml_forward_batched / gpn_lml_backward_batched) against the sequential entry
points at LARGE sizes (3000 ... 11000 rows, 2 ... 8 models) -- where the batch picks other tile shapes than a single model does (128 x 128
tiles from 4096 of them): terms, constrained gradients and dLML/d(y - m) BITWISE.  No oracle involved.  usage: fuzz_lockstep_big.py"""
import sys; sys.path.insert(0, __import__('os').path.abspath(__import__('os').path.join(__import__('os').path.dirname(__file__), '..', '..')))
import numpy as np, torch
from gptorch_amd import _backward, _ops, rng
dev = torch.device("cuda:0")
bad = 0
for n, d, dy, batch, kind, ard in [(3000, 4, 1, 8, "Rbf", False), (4096, 8, 1, 6, "Matern52", True), (5000, 3, 2, 5, "Rbf", False), (6144, 6, 1, 4, "Matern32", False),
                                   (8192, 8, 1, 8, "Rbf", False), (8192, 8, 2, 2, "Matern52", True), (10240, 4, 1, 3, "Rbf", False), (11000, 5, 1, 2, "Exp", False), (7000, 20, 1, 3, "Rbf", True)]:
    g = torch.Generator().manual_seed(n + batch)
    x, y = rng.make_regression(n, d, dy, seed=3)
    X, Y = torch.as_tensor(x).to(dev), torch.as_tensor(y).to(dev)
    var = (0.5 + torch.rand(batch, generator=g, dtype=torch.float64)).to(dev)
    ls = (0.7 + torch.rand(batch, d if ard else 1, generator=g, dtype=torch.float64)).to(dev) * float(np.sqrt(d))
    nz = (0.01 + 0.05 * torch.rand(batch, generator=g, dtype=torch.float64)).to(dev)
    fb, terms = _ops.lml_forward_batched(kind, X, Y, var, ls, nz)
    assert int(fb.info.cpu().abs().max()) == 0
    grads, g_R = _ops.lml_backward_batched(kind, X, var, ls, fb, need_resid=True)
    nls = ls.shape[1]
    ok = True
    for b in range(batch):
        f, t = _ops.lml_forward(kind, X, Y, var[b:b + 1], ls[b], nz[b:b + 1], refine=False)
        gv, gl, gn, gr = _backward.lml_backward(kind, X, var[b:b + 1], ls[b], nz[b:b + 1], f)
        ok = ok and torch.equal(t, terms[b]) and torch.equal(grads[b, 0:1], gv) and torch.equal(grads[b, 1:1 + nls], gl) and torch.equal(grads[b, 1 + nls:], gn) and torch.equal(g_R[b], gr)
    print(n, d, dy, batch, kind, ard, "bit-identical", ok, flush=True)
    bad += int(not ok)
    del fb, grads, g_R
    torch.cuda.empty_cache()
print("violations", bad)
sys.exit(1 if bad else 0)

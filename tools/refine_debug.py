#!/usr/bin/env python3
"""GPU-box tool: gpn_lml_refine piece by piece against torch -- a_hat = L^-T alpha, Kyy a_hat, the refined quadratic form --
for a few sizes around the blocking edges (usage: tools/refine_debug.py [n ...])."""
import os, sys
import numpy as np
import torch
ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, ROOT)
from gptorch_amd import _native, _ops, rng  # noqa: E402

dev = torch.device("cuda:0")
lib = _native.lib()
t = lambda v: torch.tensor(np.atleast_1d(v), dtype=torch.float64, device=dev)
for n in [int(a) for a in sys.argv[1:]] or [128, 256, 896, 1000, 1024, 2048, 2500]:
    for kind, d, dy in (("Rbf", 3, 1), ("Matern52", 16, 2)):
        x, y = rng.make_regression(n, d, dy, seed=11)
        X, Y = torch.tensor(x, device=dev), torch.tensor(y, device=dev)
        var, ls, nz = 1.3, 1.5 * np.sqrt(d / 3.0), 0.05
        tv, tl, tn = t(var), t(ls), t(nz)
        f, terms = _ops.lml_forward(kind, X, Y, tv, tl, tn, refine=False)
        quad0 = terms[1].item()
        work = torch.zeros(int(lib.gpn_lml_refine_work_bytes(n, dy)) // 8, dtype=torch.float64, device=dev)
        out = terms.clone()
        st = lib.gpn_lml_refine(_ops._stream(dev), _ops.KINDS[kind], _ops._ptr(X), n, d, _ops._ptr(Y), None, dy, _ops._ptr(tv),
                                _ops._ptr(tl), 1, _ops._ptr(tn), _ops._ptr(f.A), f.ld, _ops._ptr(f.winv), _ops._ptr(work), _ops._ptr(out))
        _native.check(st, "gpn_lml_refine")
        torch.cuda.synchronize()
        lds = (n + 127) // 128 * 128
        a_nat = work[dy * lds:2 * dy * lds].view(dy, lds)[:, :n]
        L = torch.tril(f.A[:n, :n])
        alpha = f.A[n:n + dy, :n]
        a_ref = torch.linalg.solve_triangular(L.t(), alpha.t(), upper=True).t()
        K = _ops.kernel_matrix(kind, X, None, tv, tl, noise=tn)
        ka_ref = (K @ a_ref.t()).t()
        nseg = 1                                          # refine_gather_kernel's per-row sums (hi, lo)
        ka_nat = work[2 * dy * lds:2 * dy * lds + dy * lds * 2].view(dy, lds, 2).sum(-1)[:, :n]
        print("n=%5d %-8s d=%2d dy=%d nseg=%d | a err %.2e | Ka err %.2e | quad plain %.12g refined %.12g (rel diff %.2e) | max|r| %.2e"
              % (n, kind, d, dy, nseg, (a_nat - a_ref).abs().max().item() / a_ref.abs().max().item(),
                 (ka_nat - ka_ref).abs().max().item() / ka_ref.abs().max().item(), quad0, out[1].item(),
                 abs(out[1].item() - quad0) / abs(quad0), work[2 * dy * lds + dy * lds * 2].item()))

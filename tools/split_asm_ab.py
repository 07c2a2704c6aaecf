#!/usr/bin/env python3
"""GPU-box A/B (tools' build): gpn_lml_forward with the K assembly of the columns right of the first top-level panel on a side
stream underneath that panel's chain (gpn_debug_set_split_assembly 1) against one assembly launch up front (0, shipped:
the split measured neutral).
split_asm_ab.py [n,d ...]"""
import os, sys, time
import numpy as np
import torch
ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, ROOT)
from gptorch_amd import _native, _ops, rng  # noqa: E402
cases = [tuple(int(v) for v in a.split(",")) for a in sys.argv[1:]] or [(32768, 16), (20480, 8), (24000, 8)]
dev = torch.device("cuda:0")
lib = _native.debug_begin()
for n, d in cases:
    x, y = rng.make_regression(n, d, 1, seed=0)
    X, Y = torch.as_tensor(x).to(dev), torch.as_tensor(y).to(dev)
    var = torch.tensor([1.0], dtype=torch.float64, device=dev)
    ls = torch.tensor([float(np.sqrt(d))], dtype=torch.float64, device=dev)
    nz = torch.tensor([1e-2], dtype=torch.float64, device=dev)
    f, res = None, {}
    reps = 6 if n <= 32768 else 3
    for mode in (0, 1, 0, 1):
        lib.gpn_debug_set_split_assembly(mode)
        for _ in range(2):
            f, t = _ops.lml_forward("Matern52", X, Y, var, ls, nz, factor=f, refine=False)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(reps):
            f, t = _ops.lml_forward("Matern52", X, Y, var, ls, nz, factor=f, refine=False)
        torch.cuda.synchronize()
        res.setdefault(mode, []).append(((time.perf_counter() - t0) / reps, t.clone(), torch.tril(f.A[:n, :n]).sum().item()))
    same = all(torch.equal(res[0][0][1], r[1]) and res[0][0][2] == r[2] for rs in res.values() for r in rs)
    print("N %6d D %2d: one assembly launch %.3f / %.3f ms | split, rest under the first panel %.3f / %.3f ms | bit-identical %s"
          % (n, d, res[0][0][0] * 1e3, res[0][1][0] * 1e3, res[1][0][0] * 1e3, res[1][1][0] * 1e3, same), flush=True)
    del f, X, Y
    torch.cuda.empty_cache()

#!/usr/bin/env python3
"""GPU-box tool: K assembly alone (kmat_kernel, lower tiles into a factor buffer), ms and TB/s of algorithmic bytes
8 (N (N + 1) / 2 + N D).  usage: kmat_bench.py [c2|c3|c4 ...]"""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), "..")))
from gptorch_amd import _ops, rng
dev = torch.device("cuda:0")
CASES = {"c2": (8192, 8, "Rbf"), "c3": (32768, 16, "Matern52"), "c4": (65536, 32, "Rbf"), "c3rbf": (32768, 16, "Rbf"), "m32_16k": (16384, 32, "Matern52")}
for name in (sys.argv[1:] or ["c2", "c3", "c4"]):
    n, d, kind = CASES[name]
    x, _ = rng.make_regression(n, d, 1, seed=0)
    X = torch.as_tensor(x).to(dev)
    one = lambda v: torch.tensor([v], dtype=torch.float64, device=dev)
    f = _ops.Factor(n, 1, dev)
    args = (kind, X, None, one(1.0), one(float(np.sqrt(d))))
    for _ in range(2):
        _ops.kernel_matrix(*args, noise=one(1e-2), out=f.A, ldk=f.ld, lower=True)
    torch.cuda.synchronize()
    ts = []
    for _ in range(7):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        _ops.kernel_matrix(*args, noise=one(1e-2), out=f.A, ldk=f.ld, lower=True)
        e1.record()
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1))
    ms = sorted(ts)[len(ts) // 2]
    byt = 8.0 * (n * (n + 1) / 2 + n * d)
    chk = float(f.A[:n, :n].diagonal().sum().item()) + float(f.A[n - 1, :n].sum().item())
    print("%-8s n %6d d %2d %-8s: %.3f ms  %.2f TB/s = %.3f of 8 TB/s   checksum %.15e" % (name, n, d, kind, ms, byt / ms / 1e9, byt / ms / 1e9 / 8.0, chk), flush=True)
    del f

#!/usr/bin/env python3
"""GPU-box A/B (tools' build): the K-clipped contractions of the lock-step backward (triangular inversion nodes x models,
U U^T) on 128x128 tiles from a given K / tile count, against the shipped rule.  tri_tile_ab.py [B]"""
import ctypes, os, sys, time
import numpy as np
import torch
ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, ROOT)
from gptorch_amd import _native, _ops, rng  # noqa: E402
B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
n, d = 8192, 8
dev = torch.device("cuda:0")
x, y = rng.make_regression(n, d, 1, seed=0)
X, Y = torch.as_tensor(x).to(dev), torch.as_tensor(y).to(dev)
var = torch.linspace(1.0, 1.1, B, dtype=torch.float64, device=dev)
ls = (float(np.sqrt(d)) * torch.linspace(1.0, 1.2, B, dtype=torch.float64, device=dev))[:, None]
nz = torch.full((B,), 1e-2, dtype=torch.float64, device=dev)
lib = _native.debug_begin()
lib.gpn_debug_set_tri_big.restype = ctypes.c_int
lib.gpn_debug_set_tri_big.argtypes = [ctypes.c_int, ctypes.c_int]
fb, terms = _ops.lml_forward_batched("Rbf", X, Y, var, ls, nz)
ref = None
for k, tiles in [(0, 0), (4096, 4096), (2048, 4096), (1024, 4096), (1024, 2048), (512, 1024), (0, 0), (2048, 4096)]:
    lib.gpn_debug_set_tri_big(k, tiles)
    ts = []
    for it in range(5):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        grads, _ = _ops.lml_backward_batched("Rbf", X, var, ls, fb)
        torch.cuda.synchronize()
        ts.append(time.perf_counter() - t0)
    if ref is None:
        ref = grads.clone()
    print("tri big from K >= %5d, tiles >= %5d: backward of %d models %.2f ms (min of 5; %.1f %% of peak on 2N^3/3)  bit-identical %s"
          % (k, tiles, B, min(ts[1:]) * 1e3, 100 * B * 2 * n ** 3 / 3 / min(ts[1:]) / 78.6e12, torch.equal(ref, grads)), flush=True)

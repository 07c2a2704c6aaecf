#!/usr/bin/env python3
"""GPU-box diagnostic: where the 128x128 leaf kernel spends its cycles (s_memtime stamps)."""
import os, sys
import numpy as np
import torch
ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, ROOT)
from gptorch_amd import _native, _ops
from gptorch_amd._ops import _ptr, _stream
dev = torch.device("cuda:0")
n = 128
a = torch.randn(n, n, dtype=torch.float64, device=dev)
spd = a @ a.t() / n + 0.5 * torch.eye(n, dtype=torch.float64, device=dev)
f = _ops.Factor(n, 0, dev)
diag = torch.zeros(72, dtype=torch.int64, device=dev)
lib = _native.lib()
for it in range(3):
    f.A[:n, :n] = spd
    f.info.zero_()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    _native.debug_begin().gpn_debug_leaf_timing(_stream(dev), _ptr(f.A), f.ld, _ptr(f.winv), _ptr(f.info), _ptr(diag))
    e1.record()
    torch.cuda.synchronize()
    d = diag.cpu().numpy().reshape(9, 8)
    print("launch %.1f us; cycles per pivot block, rows = waves (8 = pivot wave), cols = [top, A, bar, B, bar, C, bar | tail total]" % (e0.elapsed_time(e1) * 1e3))
    print(np.round(d[:, :7] / 15.0).astype(int), d[:, 7])
L = torch.linalg.cholesky(spd)
print("max err L:", (torch.tril(f.A[:n, :n]) - L).abs().max().item(),
      " W:", (f.winv[:n * n].reshape(n, n) - torch.linalg.inv(L)).abs().max().item())

#!/usr/bin/env python3
"""GPU-box diagnostic: where the 64x64 leaf kernel spends its cycles (s_memtime stamps)."""
import os, sys
import numpy as np
import torch
ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, ROOT)
from gptorch_amd import _native, _ops
from gptorch_amd._ops import _ptr, _stream
dev = torch.device("cuda:0")
a = torch.randn(64, 64, dtype=torch.float64, device=dev)
spd = a @ a.t() / 64 + 0.5 * torch.eye(64, dtype=torch.float64, device=dev)
f = _ops.Factor(64, 0, dev)
diag = torch.zeros(24, dtype=torch.int64, device=dev)
lib = _native.lib()
for it in range(3):
    f.A[:64, :64] = spd
    f.info.zero_()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    lib.gpn_debug_leaf_timing(_stream(dev), _ptr(f.A), f.ld, _ptr(f.winv), _ptr(f.info), _ptr(diag))
    e1.record()
    torch.cuda.synchronize()
    d = diag.cpu().numpy().reshape(4, 6)
    print("launch %.1f us; cycles per column by wave x segment [barrier, pivot+reads, publish, park, bulk, tail(total)]:" % (e0.elapsed_time(e1) * 1e3))
    print(np.round(d[:, :5] / 64.0, 1), d[:, 5])
L = torch.linalg.cholesky(spd)
print("max err vs torch:", (torch.tril(f.A[:64, :64]) - L).abs().max().item())

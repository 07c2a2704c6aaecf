#!/usr/bin/env python3
"""GPU-box tool: wall time of the 128x128 leaf kernel alone (factor + inverse), back-to-back launches."""
import os, sys
import torch
ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, ROOT)
from gptorch_amd import _native, _ops
from gptorch_amd._ops import _ptr, _stream
dev = torch.device("cuda:0")
n = 128
a = torch.randn(n, n, dtype=torch.float64, device=dev)
spd = a @ a.t() / n + 0.5 * torch.eye(n, dtype=torch.float64, device=dev)
lib = _native.lib()
R = 200
fs = []
for _ in range(R):
    f = _ops.Factor(n, 0, dev)
    f.A[:n, :n] = spd
    fs.append(f)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for f in fs:
    lib.gpn_potrf_lower(_stream(dev), _ptr(f.A), n, 0, f.ld, _ptr(f.winv), _ptr(f.info))
e1.record()
torch.cuda.synchronize()
L = torch.linalg.cholesky(spd)
print("leaf: %.2f us per launch (back-to-back, incl. ~launch gap); max err %.2e" % (
    e0.elapsed_time(e1) * 1e3 / R, (fs[-1].lower() - L).abs().max().item()))

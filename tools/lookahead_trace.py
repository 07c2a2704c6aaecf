#!/usr/bin/env python3
"""GPU-box tool (round 6): a few LML evaluations with the outer-panel look-ahead forced on / off, for rocprofv3 --kernel-trace
(tools/trace_timeline.py prints the window).  usage: lookahead_trace.py <n> <mode> [extra nwg pad]   (tools' build)"""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), "..")))
from gptorch_amd import _native, _ops, rng
n, mode = int(sys.argv[1]), int(sys.argv[2])
extra, nwg, pad = ([int(v) for v in sys.argv[3:6]] + [0, 0, 20])[:3] if len(sys.argv) > 3 else (0, 0, 20)
dev = torch.device("cuda:0")
lib = _native.debug_begin()
lib.gpn_debug_set_outer_lookahead(mode, extra, nwg, pad)
x, y = rng.make_regression(n, 8, 1, seed=0)
X, Y = torch.as_tensor(x).to(dev), torch.as_tensor(y).to(dev)
var = torch.tensor([1.0], dtype=torch.float64, device=dev)
ls = torch.tensor([float(np.sqrt(8))], dtype=torch.float64, device=dev)
nz = torch.tensor([1e-2], dtype=torch.float64, device=dev)
f = None
for _ in range(4):
    f, t = _ops.lml_forward("Rbf", X, Y, var, ls, nz, factor=f, refine=False)
torch.cuda.synchronize()
print(t[2].item())
_native.debug_end()

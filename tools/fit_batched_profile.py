#!/usr/bin/env python3
"""GPU-box tool: a few lock-step loss + backward passes (gpn_lml_forward_batched + gpn_lml_backward_batched) for
`rocprofv3 --kernel-trace --stats`:  fit_batched_profile.py c2|c1 B [bwd]   (bwd: backward launches only after one forward)."""
import os, sys
import numpy as np
import torch
ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, ROOT)
from gptorch_amd import _ops, rng  # noqa: E402
what, B = sys.argv[1], int(sys.argv[2])
only_bwd = len(sys.argv) > 3 and sys.argv[3] == "bwd"
n, d = (8192, 8) if what == "c2" else (512, 2)
reps = 6 if what == "c2" else 50
dev = torch.device("cuda:0")
x, y = rng.make_regression(n, d, 1, seed=0)
X, Y = torch.as_tensor(x).to(dev), torch.as_tensor(y).to(dev)
var = torch.linspace(1.0, 1.1, B, dtype=torch.float64, device=dev)
ls = (float(np.sqrt(d)) * torch.linspace(1.0, 1.2, B, dtype=torch.float64, device=dev))[:, None]
nz = torch.full((B,), 1e-2, dtype=torch.float64, device=dev)
fb, terms = _ops.lml_forward_batched("Rbf", X, Y, var, ls, nz)
for it in range(reps):
    if not only_bwd:
        fb, terms = _ops.lml_forward_batched("Rbf", X, Y, var, ls, nz, fb=fb)
    grads, _ = _ops.lml_backward_batched("Rbf", X, var, ls, fb)
    torch.cuda.synchronize()
print(terms[:, 2].cpu().numpy(), fb.info.cpu().numpy(), grads[0].cpu().numpy())

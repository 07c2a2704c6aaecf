#!/usr/bin/env python3
"""GPU-box tool: GPR._predict at N* = 1024 with the factor cached, for rocprofv3 --kernel-trace --stats (predict_profile.py c2|c3 [chain|blocked])."""
import os, sys
import torch
ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from gptorch_amd import rng  # noqa: E402
import gptorch_amd.models.gpr as gpr_mod  # noqa: E402
w = bench.WORKLOADS[sys.argv[1] if len(sys.argv) > 1 else "c2"]
mode = sys.argv[2] if len(sys.argv) > 2 else "blocked"          # chain | blocked
gpr_mod.INVERSE_AFTER_CALLS = 10 ** 9
gpr_mod.BLOCKED_AFTER_CALLS = 10 ** 9 if mode == "chain" else 1
m, _, _ = bench.build_model(w, 0, torch.device("cuda:0"))
xs = torch.tensor(rng.normal(77, (1024, w["d"])), device="cuda:0")
with torch.no_grad():
    for _ in range(8):
        mu, var = m._predict(xs)
    torch.cuda.synchronize()
print(float(mu.sum()), float(var.sum()))

#!/usr/bin/env python3
"""GPU-box tool: C3-sized LML evaluations with the refinement step, for rocprofv3 --kernel-trace --stats."""
import os, sys
import torch
ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
w = bench.WORKLOADS[sys.argv[1] if len(sys.argv) > 1 else "c3"]
m, _, _ = bench.build_model(w, 0, torch.device("cuda:0"))
with torch.no_grad():
    for _ in range(4):
        v = m.log_likelihood()
torch.cuda.synchronize()
print(v.item())

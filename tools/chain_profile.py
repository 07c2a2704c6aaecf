#!/usr/bin/env python3
"""GPU-box tool: a few LML evaluations of a bench workload for rocprofv3 --kernel-trace --stats (GPN_REFINE_MIN_N=0 to leave
the refinement out; POTRF_VARIANT=<int> selects a driver variant through the tools' build)."""
import os, sys
import torch
ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from gptorch_amd import _native  # noqa: E402
w = bench.WORKLOADS[sys.argv[1] if len(sys.argv) > 1 else "c3"]
v = int(os.environ.get("POTRF_VARIANT", "0"), 0)
if v:
    _native.debug_begin().gpn_debug_set_potrf_variant(v)
m, _, _ = bench.build_model(w, 0, torch.device("cuda:0"))
with torch.no_grad():
    for _ in range(int(sys.argv[2]) if len(sys.argv) > 2 else 4):
        val = m.log_likelihood()
torch.cuda.synchronize()
print(val.item())

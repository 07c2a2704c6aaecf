#!/usr/bin/env python3
"""GPU-box tool: loss()+backward() steps at a bench workload (for rocprofv3 kernel traces)."""
import os, sys, time
import torch
ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
w = bench.WORKLOADS[sys.argv[1] if len(sys.argv) > 1 else "c2"]
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 4
m, _, _ = bench.build_model(w, 0, torch.device("cuda:0"))
for i in range(steps + 1):
    if i == 1:
        torch.cuda.synchronize(); t0 = time.perf_counter()
    m.zero_grad(); m.loss().backward()
torch.cuda.synchronize()
print("loss+backward: %.2f ms/step" % ((time.perf_counter() - t0) / steps * 1e3))

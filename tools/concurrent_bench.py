#!/usr/bin/env python3
"""GPU-box tool: LML evals/s with R independent models in flight on R HIP streams."""
import os, sys, time
import torch
ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from gptorch_amd.models import batched_log_likelihood  # noqa: E402

w = bench.WORKLOADS[sys.argv[1] if len(sys.argv) > 1 else "c2"]
dev = torch.device("cuda:0")
for R in [1, 2, 3, 4, 6, 8]:
    models = [bench.build_model(w, seed=r, device=dev)[0] for r in range(R)]
    streams = [torch.cuda.Stream(device=dev) for _ in range(R)]
    for _ in range(2):
        out = batched_log_likelihood(models, streams)
    torch.cuda.synchronize()
    rounds = 6
    t0 = time.perf_counter()
    for _ in range(rounds):
        out = batched_log_likelihood(models, streams)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print("R=%d: %.1f evals/s  (%.2f ms per round of %d)  lml[0]=%.6f" % (R, R * rounds / dt, dt / rounds * 1e3, R, out[0].item()), flush=True)
    del models

#!/usr/bin/env python3
"""GPU-box tool: BASELINE config 3 end to end -- GPR + Matern52, N = 32768, D = 16, 50 Adam steps of
hyper-parameter optimisation through GPModel.optimize (base.py:260-269 loop: loss, backward, step,
loss.item())."""
import json, os, sys, time
import torch
ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
wl = sys.argv[1] if len(sys.argv) > 1 else "c3"
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 50
w = bench.WORKLOADS[wl]
m, _, _ = bench.build_model(w, 0, torch.device("cuda:0"))
m.loss().backward(); m.zero_grad()          # warm-up (allocations, side streams)
torch.cuda.synchronize()
t0 = time.perf_counter()
losses, _ = m.optimize(method="Adam", max_iter=steps, verbose=False, learning_rate=0.01)
torch.cuda.synchronize()
dt = time.perf_counter() - t0
print(json.dumps({"workload": w["name"] + ", %d Adam steps (lr 0.01)" % steps, "seconds": dt, "ms_per_step": dt / steps * 1e3,
                  "loss_first": float(losses[0]), "loss_last": float(losses[-1]),
                  "peak_hbm_gb": torch.cuda.max_memory_allocated() / 1e9}))

#!/usr/bin/env python3
"""GPU-box tool: B = 8 C2 restarts as ONE lock-step group vs several groups on separate HIP streams."""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), "..")))
from gptorch_amd import _ops, rng
dev = torch.device("cuda:0")
n, d, B = 8192, 8, int(sys.argv[1]) if len(sys.argv) > 1 else 8
x, y = rng.make_regression(n, d, 1, seed=0)
X, Y = torch.as_tensor(x).to(dev), torch.as_tensor(y).to(dev)
var = torch.linspace(1.0, 1.1, B, dtype=torch.float64, device=dev)
ls = (float(np.sqrt(d)) * torch.linspace(1.0, 1.2, B, dtype=torch.float64, device=dev))[:, None]
nz = torch.full((B,), 1e-2, dtype=torch.float64, device=dev)
for groups in (1, 2, 4, 1):
    g = B // groups
    streams = [torch.cuda.Stream(device=dev) for _ in range(groups)]
    fbs = [None] * groups

    def run():
        outs = []
        for i, st in enumerate(streams):
            with torch.cuda.stream(st):
                fbs[i], t = _ops.lml_forward_batched("Rbf", X, Y, var[i * g:(i + 1) * g], ls[i * g:(i + 1) * g], nz[i * g:(i + 1) * g], fb=fbs[i])
                outs.append(t)
        return outs
    for _ in range(2):
        run()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(5):
        outs = run()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 5
    print("%d group(s) of %d on %d stream(s): %.2f ms per %d = %.1f evals/s  lml[0] %.10f" % (groups, g, groups, dt * 1e3, B, B / dt, outs[0][0, 2].item()), flush=True)

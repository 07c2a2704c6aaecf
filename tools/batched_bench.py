#!/usr/bin/env python3
"""GPU-box tool: evaluations/s of `batched_log_likelihood` (lock step, gpn_lml_forward_batched) against the same models one after
the other and against the round-3 two-stream placement.  Usage: batched_bench.py [c2|c1|n=<N>,d=<D>] [B ...]"""
import os, sys, time
import numpy as np
import torch
ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, ROOT)
from gptorch_amd import kernels, likelihoods, rng  # noqa: E402
from gptorch_amd.models import GPR, batched_log_likelihood  # noqa: E402
from gptorch_amd.models.gpr import two_lane_streams  # noqa: E402

what = sys.argv[1] if len(sys.argv) > 1 else "c2"
Bs = [int(v) for v in sys.argv[2:]] or [1, 2, 4, 8, 16]
n, d = (8192, 8) if what == "c2" else (512, 2) if what == "c1" else tuple(int(t.split("=")[1]) for t in what.split(","))
dev = torch.device("cuda:0")
x, y = rng.make_regression(n, d, 1, seed=0)


def models(B):
    ms = []
    for b in range(B):
        m = GPR(x, y, kernels.Rbf(d, variance=1.0 + 0.01 * b, length_scales=float(np.sqrt(d)) * (1.0 + 0.02 * b)),
                likelihood=likelihoods.Gaussian(variance=1e-2))
        m.cuda()
        ms.append(m)
    for m in ms[1:]:
        m.X, m.Y = ms[0].X, ms[0].Y            # restarts share the data
    return ms


def timed(fn, reps, warm=2):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps


for B in Bs:
    ms = models(B)
    with torch.no_grad():
        t_seq = timed(lambda: [m.log_likelihood() for m in ms], 5)
        t_bat = timed(lambda: batched_log_likelihood(ms), 5)
        t_two = timed(lambda: batched_log_likelihood(ms, two_lane_streams(ms)), 5) if B > 1 else t_seq
        same = [a.item() for a in batched_log_likelihood(ms)] == [m.log_likelihood().item() for m in ms]
    print("N %d D %d B %3d: sequential %9.1f evals/s | two streams %9.1f | lock step %9.1f evals/s (%.3f ms per batch) bit-identical %s"
          % (n, d, B, B / t_seq, B / t_two, B / t_bat, t_bat * 1e3, same), flush=True)
    del ms
    torch.cuda.empty_cache()

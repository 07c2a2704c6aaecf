#!/usr/bin/env python3
"""GPU-box tool (round 6): ms per optimiser step of ONE model (the reference's loop, base.py:260-269), ordinary loop against
optimize(capture=True) -- the step as one hipGraph replay.  usage: capture_bench.py [n ...] [--iters K] [--method Adam]"""
import argparse, contextlib, io, os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), "..")))
from gptorch_amd import kernels, likelihoods, rng
from gptorch_amd.models import GPR
ap = argparse.ArgumentParser()
ap.add_argument("sizes", type=int, nargs="*", default=[512, 2048, 4096])
ap.add_argument("--iters", type=int, default=200)
ap.add_argument("--method", default="Adam")
args = ap.parse_args()
for n in args.sizes:
    d = 2 if n <= 512 else 8
    x, y = rng.make_regression(n, d, 1, seed=0)
    res = {}
    for cap in (False, True, False, True):
        m = GPR(x, y, kernels.Rbf(d, variance=1.0, length_scales=float(np.sqrt(d))), likelihood=likelihoods.Gaussian(variance=1e-2))
        m.cuda()
        with contextlib.redirect_stdout(io.StringIO()):
            m.optimize(method=args.method, max_iter=8, verbose=False, capture=cap)          # warm caches, plans, allocator
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            losses, _ = m.optimize(method=args.method, max_iter=args.iters, verbose=False, capture=cap)
            torch.cuda.synchronize()
            dt = time.perf_counter() - t0
        res.setdefault(cap, []).append((dt / args.iters * 1e3, losses[-1]))
    a, b = min(v[0] for v in res[False]), min(v[0] for v in res[True])
    print("n %6d d %d %s: ordinary loop %.3f ms/step | captured %.3f ms/step (%.2f x) | final loss %.10f vs %.10f" % (
        n, d, args.method, a, b, a / b, res[False][0][1], res[True][0][1]), flush=True)

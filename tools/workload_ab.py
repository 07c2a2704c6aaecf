#!/usr/bin/env python3
"""GPU-box tool: same-box A/B of whole workloads under contraction-kernel debug variants
(gpn_debug_set_gemm_variant), interleaved round by round.
    python tools/workload_ab.py 0,20 [c3,c2,c4,c5,bwd3,bwd2]
Variant 0 is the shipped path; see gemm_f64.hip for the others."""
import os, sys, time
import numpy as np
import torch
ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from gptorch_amd import _native  # noqa: E402

variants = [int(v) for v in (sys.argv[1] if len(sys.argv) > 1 else "0").split(",")]
legs = (sys.argv[2] if len(sys.argv) > 2 else "c3,c2,c4,c5,bwd3").split(",")
rounds = int(os.environ.get("GPN_AB_ROUNDS", "3"))
dev = torch.device("cuda:0")
lib = _native.lib()


def timed(fn, steps):
    fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        out = fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / steps * 1e3, out


def vfe_model():
    from gptorch_amd import kernels, likelihoods, mean_functions, rng
    from gptorch_amd.models import VFE
    n, m, d = 1000000, 4096, 8
    xv, yv = rng.make_regression(n, d, 1, seed=0)
    z = rng.normal(99, (m, d))
    mod = VFE(xv, yv, kernels.Rbf(d, variance=1.0, length_scales=float(np.sqrt(d))), inducing_points=z,
              likelihood=likelihoods.Gaussian(variance=1e-2), mean_function=mean_functions.Zero(1))
    mod.cuda()
    return mod


for leg in legs:
    key = {"bwd3": "c3", "bwd2": "c2"}.get(leg, leg)
    model = vfe_model() if leg == "c5" else bench.build_model(bench.WORKLOADS[key], 0, dev)[0]
    steps = {"c2": 20, "c3": 3, "c4": 1, "c5": 1, "bwd3": 1, "bwd2": 5}[leg]

    def fwd():
        with torch.no_grad():
            return model.log_likelihood()

    def fb():
        for p in model.parameters():
            p.grad = None
        loss = model.loss()
        loss.backward()
        return loss.detach()
    fn = fb if leg.startswith("bwd") else fwd
    times = {v: [] for v in variants}
    vals = {}
    for r in range(rounds):
        for v in variants:
            _native.debug_begin().gpn_debug_set_gemm_variant(v)
            t, o = timed(fn, steps)
            times[v].append(t)
            vals[v] = o.item()
    _native.debug_end()
    print(leg, "  ".join("v%d %.3f ms [%.3f-%.3f] val %.10f" % (v, sorted(times[v])[len(times[v]) // 2], min(times[v]), max(times[v]), vals[v])
                         for v in variants), flush=True)
    del model
    torch.cuda.empty_cache()

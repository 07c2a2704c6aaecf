#!/usr/bin/env python3
"""GPU-box tool: predict path timing at the bench workload (C2: N*=1024 test points)."""
import os, sys, time
import numpy as np
import torch
ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from gptorch_amd import rng  # noqa: E402
w = bench.WORKLOADS[sys.argv[1] if len(sys.argv) > 1 else "c2"]
ns = int(sys.argv[2]) if len(sys.argv) > 2 else 1024
m, _, _ = bench.build_model(w, 0, torch.device("cuda:0"))
xs = torch.tensor(rng.normal(77, (ns, w["d"])), device="cuda:0")


def t(fn, reps=5):
    fn(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e3


def cold():
    m._predict_cache = None
    return m.predict_y(xs)


import gptorch_amd.models.gpr as gpr_mod
from gptorch_amd import _ops  # noqa: E402
gpr_mod.INVERSE_AFTER_CALLS = gpr_mod.BLOCKED_AFTER_CALLS = 10 ** 9
print("%s N*=%d: predict_y diag, factor cached   %.3f ms (right-solve chain down to the 128-wide leaf inverses)" % (w["name"][:2], ns, t(lambda: m.predict_y(xs))))
gpr_mod.BLOCKED_AFTER_CALLS = 0
print("%s N*=%d: predict_y diag, factor cached   %.3f ms (inverted 1024 x 1024 diagonal blocks, after their one-off construction)" % (w["name"][:2], ns, t(lambda: m.predict_y(xs))))
f_ = m._predict_cache[1]
f_._wblock = None
torch.cuda.synchronize(); t0 = time.perf_counter(); _ops.block_inverses(f_); torch.cuda.synchronize()
print("%s: forming the block inverses once: %.3f ms" % (w["name"][:2], (time.perf_counter() - t0) * 1e3))
gpr_mod.BLOCKED_AFTER_CALLS = 10 ** 9
_ops.BLOCKED_PREDICT_MIN_N, keep = 10 ** 9, _ops.BLOCKED_PREDICT_MIN_N
gpr_mod.INVERSE_AFTER_CALLS = 0
print("%s N*=%d: predict_y diag, factor cached   %.3f ms (explicit full inverse, after its one-off construction)" % (w["name"][:2], ns, t(lambda: m.predict_y(xs))))
_ops.BLOCKED_PREDICT_MIN_N = keep
gpr_mod.INVERSE_AFTER_CALLS = 10 ** 9
print("%s N*=%d: predict_y full cov, cached      %.3f ms" % (w["name"][:2], ns, t(lambda: m.predict_y(xs, diag=False))))
print("%s N*=%d: predict_y diag, incl. re-factor %.3f ms (the reference re-factorises every call)" % (w["name"][:2], ns, t(cold)))

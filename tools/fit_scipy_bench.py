#!/usr/bin/env python3
"""GPU-box tool: B restarts of scipy L-BFGS-B (what the reference's example runs, examples/regression_1d.py:53) optimised AT ONCE
(multi_start_optimize -> _multi_start_scipy: every round of function evaluations is one lock-step loss + backward) against the same
restarts one after the other.  fit_scipy_bench.py [c2|c1|n=..,d=..] [B] [max_iter]"""
import contextlib, io, os, sys, time
import numpy as np
import torch
ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, ROOT)
from gptorch_amd import kernels, likelihoods, rng  # noqa: E402
from gptorch_amd.models import GPR, multi_start_optimize  # noqa: E402
what = sys.argv[1] if len(sys.argv) > 1 else "c2"
B = int(sys.argv[2]) if len(sys.argv) > 2 else 8
iters = int(sys.argv[3]) if len(sys.argv) > 3 else 10
n, d = (8192, 8) if what == "c2" else (512, 2) if what == "c1" else tuple(int(t.split("=")[1]) for t in what.split(","))
x, y = rng.make_regression(n, d, 1, seed=0)


def models():
    ms = []
    for b in range(B):
        m = GPR(x, y, kernels.Rbf(d, variance=1.0 + 0.05 * b, length_scales=float(np.sqrt(d)) * (1.0 + 0.1 * b)), likelihood=likelihoods.Gaussian(variance=1e-2))
        m.cuda()
        ms.append(m)
    for m in ms[1:]:
        m.X, m.Y = ms[0].X, ms[0].Y
    return ms


def wall(fn):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    out = fn()
    torch.cuda.synchronize()
    return time.perf_counter() - t0, out


with contextlib.redirect_stdout(io.StringIO()):
    multi_start_optimize(models(), method="L-BFGS-B", max_iter=1)          # warm-up
    a, b = models(), models()
    t_bat, (res, _) = wall(lambda: multi_start_optimize(a, method="L-BFGS-B", max_iter=iters))
    t_seq, ref = wall(lambda: [m.optimize(method="L-BFGS-B", max_iter=iters) for m in b])
same = all(np.array_equal(r.x, q.x) and r.nfev == q.nfev for r, q in zip(res, ref))
nfev = sum(r.nfev for r in res)
print("N %d D %d, %d restarts x L-BFGS-B (max_iter %d, %d evaluations in all): one after the other %.2f s | at once %.2f s -> %.2fx; results bit-identical %s"
      % (n, d, B, iters, nfev, t_seq, t_bat, t_seq / t_bat, same))

#!/usr/bin/env python3
"""GPU-box tool: one optimiser step (loss + backward + Adam) of B restarts in lock step (multi_start_optimize /
batched_loss_and_grad -> gpn_lml_forward_batched + gpn_lml_backward_batched) against the same restarts stepped one after the
other by their own optimize().  Usage: fit_batched_bench.py [c2|c1|n=<N>,d=<D>] [B ...] [--steps K] [--parts] [--composite]
--composite: the reference's example kernel Linear + Rbf + Constant (examples/regression_1d.py:34-53) instead of Rbf: the lock-step
path of composite kernels (_expr.BatchedExprLogLik), timed through batched_loss_and_grad + one Adam step per model.
--capture: also the stacked lock-step loop with its iteration as one hipGraph replay (multi_start_optimize(capture=True))."""
import contextlib, io, os, sys, time
import numpy as np
import torch
ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, ROOT)
from gptorch_amd import _ops, kernels, likelihoods, rng  # noqa: E402
from gptorch_amd.models import GPR, batched_loss_and_grad, multi_start_optimize  # noqa: E402

args = [a for a in sys.argv[1:] if not a.startswith("--")]
steps = 10
if "--steps" in sys.argv:
    steps = int(sys.argv[sys.argv.index("--steps") + 1])
    args.remove(str(steps))
what = args[0] if args else "c2"
Bs = [int(v) for v in args[1:]] or [1, 2, 4, 8]
n, d = (8192, 8) if what == "c2" else (512, 2) if what == "c1" else tuple(int(t.split("=")[1]) for t in what.split(","))
PEAK = 78.6e12
x, y = rng.make_regression(n, d, 1, seed=0)


COMPOSITE = "--composite" in sys.argv


def models(B):
    ms = []
    for b in range(B):
        k = kernels.Rbf(d, variance=1.0 + 0.01 * b, length_scales=float(np.sqrt(d)) * (1.0 + 0.02 * b))
        if COMPOSITE:
            k = kernels.Linear(d, variance=0.1 + 0.01 * b) + k + kernels.Constant(d)
        m = GPR(x, y, k, likelihood=likelihoods.Gaussian(variance=1e-2))
        m.cuda()
        ms.append(m)
    for m in ms[1:]:
        m.X, m.Y = ms[0].X, ms[0].Y
    return ms


def quiet():
    return contextlib.redirect_stdout(io.StringIO())


def wall(fn):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    fn()
    torch.cuda.synchronize()
    return time.perf_counter() - t0


def composite_step_times(ms, steps):
    """composite kernels have no stacked-parameter loop: one lock-step loss + backward, then every model's own Adam step"""
    from gptorch_amd.models import batched_loss_and_grad
    opts = [torch.optim.Adam([p for p in m.parameters() if p.requires_grad], lr=0.01) for m in ms]
    def lock():
        for o in opts:
            o.zero_grad()
        batched_loss_and_grad(ms)
        for o in opts:
            o.step()
    def seq():
        for m, o in zip(ms, opts):
            o.zero_grad()
            m.loss().backward()
            o.step()
    for fn in (lock, seq):
        fn(); fn()
    t_lock = wall(lambda: [lock() for _ in range(steps)]) / steps
    t_seq = wall(lambda: [seq() for _ in range(steps)]) / steps
    return t_seq, t_lock


for B in Bs:
    ms = models(B)
    if COMPOSITE:
        t_seq, t_bat = composite_step_times(ms, steps)
        flops = B * float(n) ** 3
        print("N %d D %d B %3d Linear + Rbf + Constant: sequential %8.2f ms / step (%.1f %% of peak on N^3) | lock step %8.2f ms / step (%.1f %%)  -> %.2fx"
              % (n, d, B, t_seq * 1e3, 100 * flops / t_seq / PEAK, t_bat * 1e3, 100 * flops / t_bat / PEAK, t_seq / t_bat), flush=True)
        del ms
        torch.cuda.empty_cache()
        continue
    with quiet():
        multi_start_optimize(ms, max_iter=2)                                         # warm-up (allocations, first launches)
        for m in ms:
            m.optimize(method="Adam", max_iter=2, verbose=False)
        t_bat = wall(lambda: multi_start_optimize(ms, method="Adam", max_iter=steps)) / steps
        t_seq = wall(lambda: [m.optimize(method="Adam", max_iter=steps, verbose=False) for m in ms]) / steps
    flops = B * float(n) ** 3
    print("N %d D %d B %3d: sequential %8.2f ms / step (%.1f %% of peak on N^3) | lock step %8.2f ms / step (%.1f %%)  -> %.2fx; "
          "fits/s of 50 steps: %.3f vs %.3f" % (n, d, B, t_seq * 1e3, 100 * flops / t_seq / PEAK, t_bat * 1e3, 100 * flops / t_bat / PEAK,
                                                 t_seq / t_bat, B / (50 * t_seq), B / (50 * t_bat)), flush=True)
    if "--capture" in sys.argv and B > 1:
        it = max(steps, 100)
        with quiet():
            multi_start_optimize(ms, method="Adam", max_iter=8, stacked=True, capture=True)
            t_cap = wall(lambda: multi_start_optimize(ms, method="Adam", max_iter=it, stacked=True, capture=True)) / it
            t_stk = wall(lambda: multi_start_optimize(ms, method="Adam", max_iter=it, stacked=True)) / it
        print("    stacked loop, %d iterations: %.3f ms / step | as one hipGraph replay %.3f ms / step (%.2fx); fits/s of 50 steps: %.0f vs %.0f"
              % (it, t_stk * 1e3, t_cap * 1e3, t_stk / t_cap, B / (50 * t_stk), B / (50 * t_cap)), flush=True)
    if "--parts" in sys.argv and B > 1:
        k = ms[0]._stationary()
        var = torch.stack([m.kernel.variance.transform().reshape(()) for m in ms])
        ls = torch.stack([m.kernel.length_scales.transform().reshape(-1) for m in ms])
        nz = torch.stack([m.likelihood.variance.transform().reshape(()) for m in ms])
        fb, _ = _ops.lml_forward_batched("Rbf", ms[0].X, ms[0].Y, var, ls, nz)
        _ops.lml_backward_batched("Rbf", ms[0].X, var, ls, fb)
        tf = min(wall(lambda: _ops.lml_forward_batched("Rbf", ms[0].X, ms[0].Y, var, ls, nz, fb=fb)) for _ in range(3))
        tb = min(wall(lambda: _ops.lml_backward_batched("Rbf", ms[0].X, var, ls, fb)) for _ in range(3))
        print("    parts: forward %.2f ms (%.1f %% of peak on N^3/3), backward %.2f ms (%.1f %% on 2N^3/3)"
              % (tf * 1e3, 100 * flops / 3 / tf / PEAK, tb * 1e3, 100 * 2 * flops / 3 / tb / PEAK), flush=True)
    del ms
    torch.cuda.empty_cache()

#!/usr/bin/env python3
"""GPU-box tool: `loss(); backward()` of B sparse (VFE) restarts in lock step (batched_loss_and_grad ->
models/_vfe_lockstep.py) against the same restarts evaluated one after the other, and one Adam step per model on top of
both (multi_start_optimize's loop for models that keep their own optimiser).
Usage: vfe_batched_bench.py [n=<N>,m=<M>,d=<D>,B=<B> ...] [--steps K] [--parts]
--parts: the lock-step step split into forward, backward and the host's share (a run with the stream left to drain after
every phase)."""
import os, sys, time
import numpy as np
import torch
ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, ROOT)
from gptorch_amd import kernels, likelihoods, mean_functions, rng  # noqa: E402
from gptorch_amd.models import VFE, batched_loss_and_grad  # noqa: E402

args = [a for a in sys.argv[1:] if not a.startswith("--")]
steps = 10
if "--steps" in sys.argv:
    steps = int(sys.argv[sys.argv.index("--steps") + 1])
    args.remove(str(steps))
cases = args or ["n=512,m=64,d=2,B=64", "n=2048,m=256,d=4,B=16", "n=8192,m=512,d=8,B=8"]
LS = float(os.environ.get("VFE_BENCH_LS", 0.5))    # length scale / sqrt(D): K(Z) well away from the jitter ladder


def models(n, m, d, B):
    g = np.random.default_rng(0)
    x, y = rng.make_regression(n, d, 1, seed=0)
    ms = []
    for b in range(B):
        z = x[g.choice(n, m, replace=False)]
        k = kernels.Rbf(d, variance=1.0 + 0.01 * b, length_scales=LS * float(np.sqrt(d)) * (1.0 + 0.02 * b))
        v = VFE(x, y, k, inducing_points=z, likelihood=likelihoods.Gaussian(variance=0.05), mean_function=mean_functions.Zero(1))
        v.cuda()
        ms.append(v)
    for v in ms[1:]:
        v.X, v.Y = ms[0].X, ms[0].Y
    return ms


def wall(fn, reps):
    fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps


for case in cases:
    kv = dict(t.split("=") for t in case.split(","))
    n, m, d, B = int(kv["n"]), int(kv["m"]), int(kv["d"]), int(kv["B"])
    ms = models(n, m, d, B)
    opts = [torch.optim.Adam([p for p in v.parameters() if p.requires_grad], lr=0.01) for v in ms]

    def seq():
        for v, o in zip(ms, opts):
            o.zero_grad()
            loss = v.loss()
            loss.backward()
            o.step()

    def lock():
        for o in opts:
            o.zero_grad()
        batched_loss_and_grad(ms)
        for o in opts:
            o.step()

    def lock_eval():
        for v in ms:
            v.zero_grad()
        batched_loss_and_grad(ms)

    def seq_eval():
        for v in ms:
            v.zero_grad()
            v.loss().backward()

    from gptorch_amd.models import _vfe_lockstep
    r0 = _vfe_lockstep.LADDER_CLIMBS
    wall(lock_eval, 1)
    replays = (_vfe_lockstep.LADDER_CLIMBS - r0) // 2
    t_seq, t_lock = wall(seq, steps), wall(lock, steps)
    e_seq, e_lock = wall(seq_eval, steps), wall(lock_eval, steps)
    flops = B * 4.0 * n * m * m            # forward + backward of the collapsed bound (sparse_gpr.py header)
    print("N %6d M %5d D %2d B %3d: loss+backward one after the other %9.3f ms | lock step %9.3f ms -> %5.2fx"
          "   (+ one Adam step per model: %9.3f vs %9.3f ms -> %5.2fx; lock step %.1f TFLOP/s on 4 N M^2)"
          % (n, m, d, B, 1e3 * e_seq, 1e3 * e_lock, e_seq / e_lock, 1e3 * t_seq, 1e3 * t_lock, t_seq / t_lock,
             flops / e_lock / 1e12), flush=True)
    if replays:
        print("    (%d factorisations of the %d models climbed the jitter ladder, together)" % (replays, B), flush=True)
    if "--parts" in sys.argv:
        from gptorch_amd.models import gpr as gpr_mod

        def fwd():
            with torch.no_grad():
                for key, g in gpr_mod._vfe_groups(ms):
                    gpr_mod._vfe_group_bound([ms[i] for i in g], key, differentiable=False)
        t_f = wall(fwd, steps)
        print("    parts: forward %.3f ms, backward + autograd plumbing %.3f ms" % (1e3 * t_f, 1e3 * (e_lock - t_f)), flush=True)
    del ms, opts
    torch.cuda.empty_cache()

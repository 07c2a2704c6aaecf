#!/usr/bin/env python3
"""GPU-box tool: LML evals/s with the whole evaluation captured in a hipGraph, R graphs in flight."""
import os, sys, time
import torch
ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from gptorch_amd import _ops  # noqa: E402

w = bench.WORKLOADS[sys.argv[1] if len(sys.argv) > 1 else "c2"]
dev = torch.device("cuda:0")


def capture(m):
    k = m._stationary()
    with torch.no_grad():
        resid = (m.Y - m.mean_function(m.X)).contiguous()
        var, ls, nz = k.variance.transform().clone(), k.length_scales.transform().clone(), m.likelihood.variance.transform().clone()
        f = _ops.kernel_factor_async(k._kind, m.X, var, ls, nz, R=resid)     # warm-up (attributes, allocs)
        f.lml_terms()
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            _ops.kernel_factor_async(k._kind, m.X, var, ls, nz, R=resid, factor=f)
            terms = f.lml_terms()
    return g, f, terms, (resid, var, ls, nz)   # keep every captured input alive: the graph holds raw pointers


for R in [1, 2, 4, 8]:
    models = [bench.build_model(w, seed=r, device=dev)[0] for r in range(R)]
    caps = [capture(m) for m in models]
    streams = [torch.cuda.Stream(device=dev) for _ in range(R)]
    torch.cuda.synchronize()
    rounds = 8
    t0 = time.perf_counter()
    for _ in range(rounds):
        for (g, f, terms, _keep), st in zip(caps, streams):
            with torch.cuda.stream(st):
                g.replay()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print("graphs R=%d: %.1f evals/s (%.2f ms per round) lml=%.6f info=%d" % (
        R, R * rounds / dt, dt / rounds * 1e3, caps[0][2][2].item(), int(caps[0][1].info.item())), flush=True)
    del models, caps

#!/usr/bin/env python3
"""GPU-box tool: BASELINE config 5 -- sparse VFE GP, N = 1e6, M = 4096 inducing points, D = 8,
one MI355X.  Times one evaluation of the collapsed bound (streamed Kuf, TRSM, SYRK, chol(Kuu),
chol(B)) and one loss+backward (all hyper-parameters and Z), prints one JSON line.

    python tools/vfe_bench.py [--n 1000000] [--m 4096] [--d 8] [--steps 3] [--chunk 65536]
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, ROOT)
from gptorch_amd import kernels, likelihoods, mean_functions, rng  # noqa: E402
from gptorch_amd.models import VFE, sparse_gpr  # noqa: E402

PEAK = 78.6e12


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--n", type=int, default=1000000)
    ap.add_argument("--m", type=int, default=4096)
    ap.add_argument("--d", type=int, default=8)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--chunk", type=int, default=sparse_gpr.CHUNK_ROWS)
    ap.add_argument("--kind", default="Rbf")
    args = ap.parse_args()
    if os.environ.get("POTRF_VARIANT"):          # A/B of driver variants through the tools' build (libgpnative_dbg.so)
        from gptorch_amd import _native
        _native.debug_begin().gpn_debug_set_potrf_variant(int(os.environ["POTRF_VARIANT"], 0))
    sparse_gpr.CHUNK_ROWS = args.chunk
    n, m, d = args.n, args.m, args.d
    x, y = rng.make_regression(n, d, 1, seed=0)
    z = rng.normal(99, (m, d))        # "Z = first M rows of an independent draw" 
    model = VFE(x, y, getattr(kernels, args.kind)(d, variance=1.0, length_scales=float(np.sqrt(d))),
                inducing_points=z, likelihood=likelihoods.Gaussian(variance=1e-2), mean_function=mean_functions.Zero(1))
    model.cuda()
    torch.cuda.synchronize()

    def timed(fn):
        fn()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            out = fn()
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / args.steps, out

    def fwd():
        with torch.no_grad():
            return model.log_likelihood()

    def fwd_bwd():
        model.zero_grad()
        loss = model.loss()
        loss.backward()
        return loss

    torch.cuda.reset_peak_memory_stats()
    t_f, elbo = timed(fwd)
    t_fb, loss = timed(fwd_bwd)
    peak_gb = torch.cuda.max_memory_allocated() / 1e9
    f_flops = 2.0 * n * m * m + m ** 3 * 2.0 / 3.0           # TRSM N M^2 + SYRK N M^2 + two Choleskys
    b_flops = 2.0 * n * m * m                                   # dense [N,M] x [M,M]
    grads = {k: float(p.grad.abs().max().item()) for k, p in model.named_parameters() if p.grad is not None}
    print(json.dumps({
        "workload": "C5: VFE %s N=%d M=%d D=%d fp64, one MI355X" % (args.kind, n, m, d),
        "bound_eval_s": t_f, "bound_evals_per_s": 1.0 / t_f,
        "bound_tflops": f_flops / t_f / 1e12, "bound_frac_fp64_mfma_peak": f_flops / t_f / PEAK,
        "loss_backward_s": t_fb, "loss_backward_tflops": (f_flops + b_flops) / t_fb / 1e12,
        "elbo": float(elbo.item()), "loss": float(loss.item()), "chunk_rows": args.chunk,
        "peak_hbm_gb": peak_gb, "max_abs_grads": grads,
        "flops_model": "forward 2 N M^2 + 2/3 M^3, backward 2 N M^2 (sweeps and transposes not counted)"}))


if __name__ == "__main__":
    main()

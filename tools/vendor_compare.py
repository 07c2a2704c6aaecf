#!/usr/bin/env python3
"""GPU-box tool: vendor-library context for the factorisation alone -- torch.linalg.cholesky
(rocSOLVER/hipSOLVER via PyTorch-ROCm) vs gpn_potrf_lower on the same SPD matrix."""
import os, sys, time, math
import torch
ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, ROOT)
from gptorch_amd import rng, _ops  # noqa: E402

dev = torch.device("cuda:0")
for n in [int(a) for a in sys.argv[1:]] or [2048, 4096, 8192, 16384, 32768]:
    x = torch.tensor(rng.normal(0, (n, 8)), device=dev)
    one = torch.ones(1, dtype=torch.float64, device=dev)
    K = _ops.kernel_matrix("Rbf", x, None, one, one * math.sqrt(8.0), noise=one * 1e-2)
    def t(fn, reps):
        fn(); torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(reps):
            fn()
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / reps
    reps = 5 if n <= 8192 else 2
    tv = t(lambda: torch.linalg.cholesky(K), reps)
    f = _ops.Factor(n, 0, dev)
    def ours():
        f.A[:n, :n].copy_(K)
        f.potrf(check=False)
    def copy_only():
        f.A[:n, :n].copy_(K)
    to = t(ours, reps) - t(copy_only, reps)
    fl = n ** 3 / 3
    print("N=%6d  rocSOLVER (torch.linalg.cholesky): %9.3f ms (%5.1f TF)   gpn_potrf_lower: %9.3f ms (%5.1f TF)" % (
        n, tv * 1e3, fl / tv / 1e12, to * 1e3, fl / to / 1e12), flush=True)
    del K, f

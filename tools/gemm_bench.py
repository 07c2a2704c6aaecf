#!/usr/bin/env python3
"""GPU-box tool: throughput of the fp64 MFMA contraction kernel alone.
usage: gemm_bench.py [M N K lower]...   (defaults: a sweep)"""
import os
import sys
import time

import torch

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, ROOT)
from gptorch_amd import _ops, _native  # noqa: E402


def run(M, N, K, lower, reps=5):
    dev = torch.device("cuda:0")
    A = torch.randn(M + 16, K, dtype=torch.float64, device=dev)
    B = A if lower else torch.randn(N + 16, K, dtype=torch.float64, device=dev)
    C = torch.zeros(M, N, dtype=torch.float64, device=dev)
    _ops.gemm_nt(A, B, M, N, K, alpha=-1.0, beta=1.0, C=C, lower=lower)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        _ops.gemm_nt(A, B, M, N, K, alpha=-1.0, beta=1.0, C=C, lower=lower)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / reps
    flops = (M * (M + 1) if lower else 2.0 * M * N) * K
    print("M=%6d N=%6d K=%6d lower=%d: %8.3f ms  %6.2f TFLOP/s (%.1f%% of 78.6)" % (
        M, N, K, lower, ms, flops / ms / 1e9, flops / ms / 1e9 / 78.6 * 100), flush=True)


if __name__ == "__main__":
    if sys.argv[1:2] == ["--variant"]:
        _native.debug_begin().gpn_debug_set_gemm_variant(int(sys.argv[2]))
        print("gemm variant", sys.argv[2])
        del sys.argv[1:3]
    args = [int(a) for a in sys.argv[1:]]
    if args:
        for i in range(0, len(args), 4):
            run(*args[i:i + 4])
    else:
        for (M, N, K, lo) in [(4096, 4096, 4096, 0), (8192, 8192, 8192, 0), (8192, 8192, 256, 0), (8192, 8192, 64, 0),
                              (16384, 16384, 16384, 1), (16384, 16384, 512, 1), (4096, 4096, 4096, 1),
                              (8192, 64, 64, 0), (8192, 256, 256, 0), (2048, 2048, 2048, 1), (1024, 1024, 1024, 1)]:
            run(M, N, K, lo)

#!/usr/bin/env python3
"""GPU-box tool: 128x128 vs 64x64 tiles on large contractions (variants 3 / 4)."""
import os, sys
import torch
ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, ROOT)
from gptorch_amd import _ops, _native  # noqa: E402
lib = _native.lib()
dev = torch.device("cuda:0")


def t(M, N, K, lower, variant, reps=3):
    _native.debug_begin().gpn_debug_set_gemm_variant(variant)
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
    _native.debug_end()
    return e0.elapsed_time(e1) / reps * 1e3


for (M, N, K, lo) in [(8192, 8192, 8192, 0), (4096, 4096, 4096, 0), (16384, 16384, 16384, 1), (16384, 16384, 4096, 1),
                      (8192, 8192, 4096, 1), (4096, 4096, 65536, 1), (65536, 4096, 4096, 0), (65536, 2048, 2048, 0),
                      (16384, 8192, 8192, 0), (4096, 4096, 2048, 1), (24576, 24576, 8192, 1)]:
    fl = (M * (M + 1.0) if lo else 2.0 * M * N) * K
    a, b = t(M, N, K, lo, 3), t(M, N, K, lo, 4)
    print("M=%6d N=%6d K=%6d lower=%d: 128: %9.1f us (%.1f TF)   64: %9.1f us (%.1f TF)" % (M, N, K, lo, a, fl / a / 1e6, b, fl / b / 1e6), flush=True)

#!/usr/bin/env python3
"""GPU-box tool: sensitivity of the contraction kernel to the leading dimensions (address hashing /
channel aliasing check).  usage: ld_stride_test.py"""
import os, sys
sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), "..")))
import torch
from gptorch_amd import _ops
dev = torch.device("cuda:0")
def run(M, N, K, lower, ldk, ldc, reps=5):
    A = torch.randn(M + 16, ldk, dtype=torch.float64, device=dev)[:, :K]
    B = A if lower else torch.randn(N + 16, ldk, dtype=torch.float64, device=dev)[:, :K]
    C = torch.zeros(M, ldc, dtype=torch.float64, device=dev)[:, :N]
    out = []
    for rnd in range(2):
        _ops.gemm_nt(A, B, M, N, K, alpha=-1.0, beta=1.0, C=C, lower=lower)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            _ops.gemm_nt(A, B, M, N, K, alpha=-1.0, beta=1.0, C=C, lower=lower)
        e1.record(); torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / reps
        fl = (M * (M + 1) if lower else 2.0 * M * N) * K
        out.append("%.2f" % (fl / ms / 1e9))
    print("M=%d N=%d K=%d lower=%d ld(A,B)=%d ldc=%d: %s TFLOP/s" % (M, N, K, lower, ldk, ldc, out), flush=True)
for (M, K) in [(7168, 1536), (6656, 1536), (8192, 1536)]:
    for lower in (0, 1):
        for ldk, ldc in [(K, M), (K + 16, M), (K, M + 16), (K, M + 128), (K, 8320), (8320, 8320)]:
            run(M, M, K, lower, ldk, ldc)

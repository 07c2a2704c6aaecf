#!/usr/bin/env python3
"""GPU-box tool: tile-shape A/B of the contraction kernel (tools' build: gpn_debug_set_gemm_variant; 3 = 128x128, 4 = 64x64
2-stage, 5 = 64x64 8-deep ring, 6 = 32x32 8-deep ring, 0 = the launcher's heuristic).
    gemm_sweep.py            the shapes the look-ahead factorisation launches (trailing SYRK, next-column, rest-of-panel)
    gemm_sweep.py large      128x128 vs 64x64 tiles on large contractions (was gemm_sweep2.py)"""
import os, sys
import torch
ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, ROOT)
from gptorch_amd import _ops, _native  # noqa: E402
lib = _native.lib()
dev = torch.device("cuda:0")


def t(M, N, K, lower, variant, reps=20):
    _native.debug_begin().gpn_debug_set_gemm_variant(variant)
    A = torch.randn(M + 16, K, dtype=torch.float64, device=dev)
    B = A if lower else torch.randn(N + 16, K, dtype=torch.float64, device=dev)
    C = torch.zeros(M, N, dtype=torch.float64, device=dev)
    for _ in range(3):
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


if len(sys.argv) > 1 and sys.argv[1] == "large":
    for (M, N, K, lo) in [(8192, 8192, 8192, 0), (4096, 4096, 4096, 0), (16384, 16384, 16384, 1), (16384, 16384, 4096, 1),
                          (8192, 8192, 4096, 1), (4096, 4096, 65536, 1), (65536, 4096, 4096, 0), (65536, 2048, 2048, 0),
                          (16384, 8192, 8192, 0), (4096, 4096, 2048, 1), (24576, 24576, 8192, 1)]:
        fl = (M * (M + 1.0) if lo else 2.0 * M * N) * K
        a, b = t(M, N, K, lo, 3, reps=3), t(M, N, K, lo, 4, reps=3)
        print("M=%6d N=%6d K=%6d lower=%d: 128: %9.1f us (%.1f TF)   64: %9.1f us (%.1f TF)" % (M, N, K, lo, a, fl / a / 1e6, b, fl / b / 1e6), flush=True)
    sys.exit(0)
print("== trailing SYRK (lower), K = panel width")
for K in (1024, 2048):
    for M in ([1024, 2048, 3072, 4096, 5120, 6144, 7168] if K == 1024 else [2048, 6144, 10240, 14336, 22528, 30720]):
        r = {v: t(M, M, K, 1, v, reps=5) for v in (0, 3, 4)}
        fl = M * (M + 1.0) * K
        print("M=%6d K=%5d: heur %8.1f us  128: %8.1f (%.1f TF)  64: %8.1f (%.1f TF)" % (
            M, K, r[0], r[3], fl / r[3] / 1e6, r[4], fl / r[4] / 1e6), flush=True)
print("== next-column update  [M,128] -= [M,128][128,128]^T")
for M in (8064, 6016, 4096, 2048, 1024, 512, 32768, 16384):
    r = {v: t(M, 128, 128, 0, v) for v in (0, 4, 5, 6)}
    print("M=%6d: heur %6.1f  64x2 %6.1f  64ring %6.1f  32ring %6.1f us" % (M, r[0], r[4], r[5], r[6]), flush=True)
print("== rest-of-panel update  [M,N] -= [M,128][N,128]^T")
for (M, N) in ((7936, 768), (6000, 512), (4096, 768), (2048, 256), (1024, 768), (32000, 1792), (16000, 1024)):
    r = {v: t(M, N, 128, 0, v) for v in (0, 3, 4, 5)}
    print("M=%6d N=%5d: heur %6.1f  128 %6.1f  64x2 %6.1f  64ring %6.1f us" % (M, N, r[0], r[3], r[4], r[5]), flush=True)

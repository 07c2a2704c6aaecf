#!/usr/bin/env python3
"""GPU-box tool: the vendor library's fp64 GEMM (torch.mm -> rocBLAS / hipBLASLt) on the shapes of tools/gemm_ab.py, for context
next to gemm_nt_kernel's numbers (not a product path)."""
import torch
dev = torch.device("cuda:0")
def tm(fn, reps=5):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    ts = []
    for _ in range(3):
        e0.record()
        for _ in range(reps): fn()
        e1.record(); torch.cuda.synchronize(); ts.append(e0.elapsed_time(e1) / reps)
    return sorted(ts)[1]
for (M, N, K) in ((8192, 8192, 8192), (30720, 30720, 2048), (16384, 16384, 4096), (7168, 7168, 1024)):
    A = torch.randn(M, K, dtype=torch.float64, device=dev)
    B = torch.randn(N, K, dtype=torch.float64, device=dev)
    C = torch.randn(M, N, dtype=torch.float64, device=dev)
    ms = tm(lambda: torch.addmm(C, A, B.t(), beta=1.0, alpha=-1.0, out=C))
    print("torch.addmm (C -= A B^T, full %d x %d, K = %d): %.3f ms = %.2f TFLOP/s on 2MNK" % (M, N, K, ms, 2.0 * M * N * K / ms / 1e9), flush=True)
    del A, B, C

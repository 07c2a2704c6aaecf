#!/usr/bin/env python3
"""GPU-box tool: same-box A/B of contraction-kernel variants (gpn_debug_set_gemm_variant) with a
correctness check of every variant against the first one.   python tools/gemm_ab.py 0,3,4 [M N K lower ...]"""
import os, sys
import torch
ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, ROOT)
from gptorch_amd import _ops, _native  # noqa: E402

variants = [int(v) for v in (sys.argv[1] if len(sys.argv) > 1 else "0,3,4").split(",")]
shapes = [(8192, 8192, 8192, 0), (8192, 8192, 2048, 0), (30720, 30720, 2048, 1), (16384, 16384, 4096, 1), (6656, 6656, 1536, 1), (12345, 777, 1024, 0)]
if len(sys.argv) > 2:
    a = [int(v) for v in sys.argv[2:]]
    shapes = [tuple(a[i:i + 4]) for i in range(0, len(a), 4)]
ROUNDS = int(os.environ.get("GPN_AB_ROUNDS", "5"))
dev = torch.device("cuda:0")
lib = _native.lib()
for (M, N, K, lower) in shapes:
    torch.manual_seed(0)
    A = torch.randn(M + 16, K, dtype=torch.float64, device=dev)
    B = A if lower else torch.randn(N + 16, K, dtype=torch.float64, device=dev)
    C0 = torch.randn(M, N, dtype=torch.float64, device=dev)
    ref = None
    line = "M=%6d N=%6d K=%5d lower=%d:" % (M, N, K, lower)
    flops = (M * (M + 1) if lower else 2.0 * M * N) * K
    times = {v: [] for v in variants}
    errs = {}
    C = C0.clone()
    for rnd in range(ROUNDS):                    # variants interleaved round by round: clock / box drift hits all alike
        for v in variants:
            _native.debug_begin().gpn_debug_set_gemm_variant(v)
            if rnd == 0:
                C.copy_(C0)
                _ops.gemm_nt(A, B, M, N, K, alpha=-1.0, beta=1.0, C=C, lower=bool(lower))
                torch.cuda.synchronize()
                errs[v] = 0.0 if ref is None else (torch.tril(C - ref) if lower else (C - ref)).abs().max().item()
                if ref is None:
                    ref = C.clone()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(5):
                _ops.gemm_nt(A, B, M, N, K, alpha=-1.0, beta=1.0, C=C, lower=bool(lower))
            e1.record()
            torch.cuda.synchronize()
            times[v].append(e0.elapsed_time(e1) / 5)
    for v in variants:
        ms = sorted(times[v])[len(times[v]) // 2]
        line += "   v%d %8.3f ms %6.2f TF [%5.2f-%5.2f] (maxdiff %.1e)" % (
            v, ms, flops / ms / 1e9, flops / max(times[v]) / 1e9, flops / min(times[v]) / 1e9, errs[v])
    print(line, flush=True)
    del A, B, C0, ref, C
_native.debug_end()

#!/usr/bin/env python3
"""GPU-box tool: the in-panel trapezoid / left-looking update shapes of C3 IN SITU (operands and C are column ranges of one factor
buffer, leading dimension 32896) under contraction-kernel variants: TFLOP/s per shape.  trap_bench.py [variants]"""
import os, sys
import torch
ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, ROOT)
from gptorch_amd import _ops, _native  # noqa: E402
variants = [int(v) for v in (sys.argv[1] if len(sys.argv) > 1 else "0,11,8").split(",")]
dev = torch.device("cuda:0")
n, ld = 32768, 32896
buf = torch.randn(n + 144, ld, dtype=torch.float64, device=dev) * 1e-3
lib = _native.debug_begin()
# (rows below, columns updated, K, column offset of the K operand, lower): right-looking trapezoids after the 1st / 2nd / 3rd inner
# panel of the first and of a middle outer panel; the left-looking shapes (512 columns, K = 512 / 1024 / 1536)
shapes = [(32256, 1536, 512, 2), (31744, 1024, 512, 2), (31232, 512, 512, 2), (16384, 1536, 512, 2), (8192, 1536, 512, 2),
          (32256, 512, 512, 2), (31744, 512, 1024, 2), (31232, 512, 1536, 2), (30720, 30720, 2048, 1)]
for (M, N, K, lower) in shapes:
    r0 = n - M                                  # the update's first row = its first column
    P = buf[r0:, r0 - K:r0]                     # [M, K] solved panel rows, lda = ld
    C = buf[r0:, r0:r0 + N]
    flops = (M * (M + 1.0) if lower == 1 else 2.0 * M * N - (N * (N - 1.0) if lower == 2 else 0.0)) * K
    line = "M=%6d N=%6d K=%5d lower=%d:" % (M, N, K, lower)
    for v in variants:
        lib.gpn_debug_set_gemm_variant(v)
        ts = []
        for rnd in range(4):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(3):
                _ops.gemm_nt(P, P, M, N, K, alpha=-1e-9, beta=1.0, C=C, lower=lower)
            e1.record()
            torch.cuda.synchronize()
            ts.append(e0.elapsed_time(e1) / 3)
        ms = sorted(ts)[1]
        line += "   v%-2d %7.3f ms %5.1f TF" % (v, ms, flops / ms / 1e9)
    print(line, flush=True)
_native.debug_end()

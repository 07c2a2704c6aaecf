#!/usr/bin/env python3
"""GPU-box tool (round-3 review item 8): C3's first trailing update (M = 30720 lower-tile, K = 2048, C -= A A^T) with ONE
contraction-kernel variant, three launches -- the program rocprofv3 --pmc wraps in tools/macro_tile_ab.sh.
usage: macro_tile_run.py <gemm variant: 11 shipped 128x128 pipelined | 12 128x128 plain loop | 13 256x128 macro tile, plain loop>"""
import os, sys
import torch
ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, ROOT)
from gptorch_amd import _ops, _native  # noqa: E402
v = int(sys.argv[1])
M, K = 30720, 2048
dev = torch.device("cuda:0")
torch.manual_seed(0)
A = torch.randn(M + 16, K, dtype=torch.float64, device=dev)
C = torch.randn(M, M, dtype=torch.float64, device=dev)
_native.debug_begin().gpn_debug_set_gemm_variant(v)
for _ in range(3):
    _ops.gemm_nt(A, A, M, M, K, alpha=-1.0, beta=1.0, C=C, lower=True)
torch.cuda.synchronize()
_native.debug_end()
print("variant", v, "done")

#!/usr/bin/env python3
"""GPU-box tool: the 128x128 leaf's duration against the leading dimension of the matrix it sits in (its 128 rows are
lda*8 bytes apart: 1 KB when packed, 256 KB at C3) -- back to back on distinct blocks along the diagonal of one buffer."""
import os, sys
import torch
ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, ROOT)
from gptorch_amd import _native
from gptorch_amd._ops import _ptr, _stream
dev = torch.device("cuda:0")
torch.manual_seed(0)
lib = _native.lib()
a = torch.randn(128, 128, dtype=torch.float64, device=dev)
m = a @ a.t() / 128 + torch.eye(128, dtype=torch.float64, device=dev)
R = 200
for lda in (128, 1024, 8192, 8192 + 128, 16384, 32768, 32768 + 128, 32768 + 16 * 128, 65536):
    rows = min(lda, 128 * R) if lda > 128 else 128 * R
    nblk = rows // 128 if lda > 128 else R
    A = torch.zeros(rows * lda, dtype=torch.float64, device=dev)
    Av = A.view(rows, lda)
    winv = torch.zeros(nblk * 128 * 128, dtype=torch.float64, device=dev)
    info = torch.zeros(1, dtype=torch.int32, device=dev)
    for mode in ("warm", "cold"):
        for b in range(nblk):
            c = (b * 128) % lda if lda > 128 else 0
            Av[b * 128:(b + 1) * 128, c:c + 128] = m
        if mode == "cold":
            junk = torch.empty(1 << 27, dtype=torch.float64, device=dev).fill_(1.0)   # 1 GB through L2 / MALL
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for b in range(nblk):
            c = (b * 128) % lda if lda > 128 else 0
            off = (b * 128 * lda + c) * 8
            lib.gpn_potrf_lower(_stream(dev), A.data_ptr() + off, 128, 0, lda, winv.data_ptr() + b * 131072, _ptr(info))
        e1.record()
        torch.cuda.synchronize()
        print("lda %6d (%4d KB between rows) %s: %.2f us per leaf, %d blocks, info %d" % (lda, lda * 8 // 1024, mode, e0.elapsed_time(e1) * 1e3 / nblk, nblk, int(info.item())))
    del A, Av, winv

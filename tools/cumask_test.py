#!/usr/bin/env python3
"""GPU-box tool: what does a CU-masked stream (hipExtStreamCreateWithCUMask) do to a large contraction?
Times C3's first SYRK (M = 30720 lower, K = 2048) on: the torch stream, a plain created stream, and
masked streams with R of the 256 CUs switched off in two candidate bit layouts."""
import ctypes, os, sys, time
import torch
ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, ROOT)
from gptorch_amd import _native, _ops  # noqa: E402

lib = _native.lib()
dev = torch.device("cuda:0")
M, K = 30720, 2048
A = torch.randn(M + 16, K, dtype=torch.float64, device=dev)
C = torch.zeros(M, M, dtype=torch.float64, device=dev)
hip = ctypes.CDLL("libamdhip64.so")


def run(stream_ptr, label):
    def go():
        st = lib.gpn_gemm_nt(stream_ptr, M, M, K, -1.0, _ops._ptr(A), A.stride(0), _ops._ptr(A), A.stride(0), 1.0, _ops._ptr(C), C.stride(0), 1, 0)
        assert st == 0, st
    go()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(3):
        go()
    hip.hipStreamSynchronize(stream_ptr)
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / 3 * 1e3
    print("%-46s %8.2f ms  %6.2f TFLOP/s" % (label, ms, M * (M + 1) * K / ms / 1e9), flush=True)


run(_ops._stream(dev), "torch current stream")
s = ctypes.c_void_p()
assert hip.hipStreamCreateWithFlags(ctypes.byref(s), 1) == 0
run(s, "plain non-blocking stream")
for layout in (0, 1):
    for R in (0, 8, 32, 64, 128):
        mask = [0xffffffff] * 8
        for i in range(R):
            bit = 255 - i if layout == 0 else (i % 8) * 32 + 31 - i // 8
            mask[bit >> 5] &= ~(1 << (bit & 31))
        arr = (ctypes.c_uint32 * 8)(*mask)
        ms_ = ctypes.c_void_p()
        rc = _native.debug_begin().gpn_debug_masked_stream(arr, 8, ctypes.byref(ms_))
        if rc != 0:
            print("masked stream creation failed", rc)
            continue
        run(ms_, "masked: layout %d, %3d CUs off (%s)" % (layout, R, " ".join("%08x" % w for w in mask)))

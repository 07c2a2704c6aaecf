#!/usr/bin/env python3
"""GPU-box experiment: does a latency-bound chain of tiny kernels on one stream survive a big
contraction running on another stream (no CU mask)?"""
import os, sys, time
import torch
ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, ROOT)
from gptorch_amd import _ops  # noqa: E402

dev = torch.device("cuda:0")
big = torch.randn(4112, 4096, dtype=torch.float64, device=dev)
Cb = torch.zeros(4096, 4096, dtype=torch.float64, device=dev)
sm = torch.randn(144, 128, dtype=torch.float64, device=dev)
Cs = torch.zeros(128, 128, dtype=torch.float64, device=dev)
f = _ops.Factor(128, 0, dev)
a = torch.randn(128, 128, dtype=torch.float64, device=dev)
spd = a @ a.t() / 128 + 0.5 * torch.eye(128, dtype=torch.float64, device=dev)
s_main, s_side = torch.cuda.Stream(), torch.cuda.Stream()


def chain(n=40):
    for _ in range(n):
        _ops.gemm_nt(sm, sm, 128, 128, 128, alpha=-1.0, beta=1.0, C=Cs, lower=True)
        f.A[:128, :128].copy_(spd)
        f.potrf(check=False)


def bigk():
    _ops.gemm_nt(big, big, 4096, 4096, 4096, alpha=-1.0, beta=1.0, C=Cb, lower=True)


def timed(fn):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) * 1e3


chain(5); bigk()
print("chain alone      %.2f ms" % timed(lambda: chain()))
print("big alone        %.2f ms" % timed(bigk))


def both():
    with torch.cuda.stream(s_side):
        bigk(); bigk()
    with torch.cuda.stream(s_main):
        chain()


print("2x big alone     %.2f ms" % timed(lambda: (bigk(), bigk())))
print("chain || 2x big  %.2f ms" % timed(both))


# ---- with CU masks: side stream on 7/8 of the CUs, main stream on the remaining 1/8 ----
import ctypes
from gptorch_amd import _native
lib = _native.lib()


def masked(pred):
    words = (ctypes.c_uint32 * 8)()
    for cu in range(256):
        if pred(cu):
            words[cu // 32] |= (1 << (cu % 32))
    out = ctypes.c_void_p()
    st = _native.debug_begin().gpn_debug_masked_stream(words, 8, ctypes.byref(out))
    assert st == 0, (st, lib.gpn_last_hip_error())
    return torch.cuda.ExternalStream(out.value, device=dev)


for name, pm, ps in [("bits%8==0 vs rest", lambda c: c % 8 == 0, lambda c: c % 8 != 0),
                     ("first 32 vs rest", lambda c: c < 32, lambda c: c >= 32)]:
    s_main, s_side = masked(pm), masked(ps)

    def both2():
        with torch.cuda.stream(s_side):
            bigk(); bigk()
        with torch.cuda.stream(s_main):
            chain()

    def chain_masked():
        with torch.cuda.stream(s_main):
            chain()

    def big_masked():
        with torch.cuda.stream(s_side):
            bigk(); bigk()
    print("[%s] chain on masked main %.2f ms | 2x big on masked side %.2f ms | both %.2f ms" % (
        name, timed(chain_masked), timed(big_masked), timed(both2)))

#!/usr/bin/env python3
"""GPU-box tool: the contractions of the closed-form backward at C2 under forced tile shapes."""
import os, sys
import torch
ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from gptorch_amd import _backward, _native, _ops  # noqa: E402
lib = _native.lib()
wl = sys.argv[1] if len(sys.argv) > 1 else "c2"
if wl.startswith("n="):           # ad-hoc size: Rbf, D = 8
    w = dict(name=wl, kind="Rbf", n=int(wl[2:]), d=8, dy=1, variance=1.0, length_scales=8.0 ** 0.5, noise=1e-2)
else:
    w = bench.WORKLOADS[wl]
m, _, _ = bench.build_model(w, 0, torch.device("cuda:0"))
with torch.no_grad():
    m.log_likelihood()
f = m._holder["factor"]
n, dy = f.n, f.e
U = _backward._upper_inverse(f)


def t(fn, reps=5):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


for v in (0, 3, 4, 0, 3, 4):
    _native.debug_begin().gpn_debug_set_gemm_variant(v)
    a = t(lambda: _backward._kinv_lower(f, U))
    _native.debug_end()
    print("Kinv = U U^T (K-clipped SYRK), variant %d: %8.1f us" % (v, a))
for v in (0, 4, 5, 6):
    _native.debug_begin().gpn_debug_set_gemm_variant(v)
    a = t(lambda: _ops.gemm_nt(f.A[n:], U, dy, n, _ops.round_up(n, 16), tri=_ops.TRI_B_UPPER))
    _native.debug_end()
    print("a^T = alpha^T U^T (skinny), variant %d: %8.1f us" % (v, a))
for v in (0, 3, 4, 0, 3, 4):
    _native.debug_begin().gpn_debug_set_gemm_variant(v)
    a = t(lambda: _backward._upper_inverse(f, True), 3)
    _native.debug_end()
    print("U = L^-T (level-wise, incl. zeroing two buffers), variant %d: %8.1f us" % (v, a))
print("U = L^-T: chain-based recursion %8.1f us" % t(lambda: _backward._upper_inverse(f, False), 3))

#!/usr/bin/env python3
"""GPU-box A/B: the refined evaluation with the residual pass RE-COMPUTING Kyy (gpn_lml_forward + gpn_lml_refine, shipped) against
the assembly keeping a pristine copy that the residual pass reads back (gpn_lml_forward_saving + gpn_lml_refine_dense;
GPN_REFINE_SAVED_K=1).  saved_k_ab.py [n,d ...]"""
import os, sys, time
import numpy as np
import torch
ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, ROOT)
from gptorch_amd import _ops, rng  # noqa: E402
cases = [tuple(int(v) for v in a.split(",")) for a in sys.argv[1:]] or [(32768, 16), (16384, 8)]
dev = torch.device("cuda:0")
for n, d in cases:
    x, y = rng.make_regression(n, d, 1, seed=0)
    X, Y = torch.as_tensor(x).to(dev), torch.as_tensor(y).to(dev)
    var = torch.tensor([1.0], dtype=torch.float64, device=dev)
    ls = torch.tensor([float(np.sqrt(d))], dtype=torch.float64, device=dev)
    nz = torch.tensor([1e-2], dtype=torch.float64, device=dev)
    f, res = None, {}
    for mode in ("0", "1", "0", "1"):
        os.environ["GPN_REFINE_SAVED_K"] = mode
        for _ in range(2):
            f, t = _ops.lml_forward("Matern52", X, Y, var, ls, nz, factor=f, refine=True)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(6):
            f, t = _ops.lml_forward("Matern52", X, Y, var, ls, nz, factor=f, refine=True)
        torch.cuda.synchronize()
        res.setdefault(mode, []).append(((time.perf_counter() - t0) / 6, t.clone()))
    same = all(torch.equal(res["0"][0][1], r[1]) for rs in res.values() for r in rs)
    print("N %6d D %2d: residual pass re-computes K %.3f / %.3f ms | reads the saved copy %.3f / %.3f ms | refined terms bit-identical %s"
          % (n, d, res["0"][0][0] * 1e3, res["0"][1][0] * 1e3, res["1"][0][0] * 1e3, res["1"][1][0] * 1e3, same), flush=True)
    del f
    torch.cuda.empty_cache()

# parts at C3: the two assemblies and the two refinement calls alone (HIP events)
n, d = 32768, 16
x, y = rng.make_regression(n, d, 1, seed=0)
X, Y = torch.as_tensor(x).to(dev), torch.as_tensor(y).to(dev)
var = torch.tensor([1.0], dtype=torch.float64, device=dev)
ls = torch.tensor([4.0], dtype=torch.float64, device=dev)
nz = torch.tensor([1e-2], dtype=torch.float64, device=dev)
os.environ["GPN_REFINE_SAVED_K"] = "1"
f, t = _ops.lml_forward("Matern52", X, Y, var, ls, nz, refine=True)
from gptorch_amd import _native
lib = _native.lib()
s = _ops._stream(dev)
K1 = torch.empty(n, f.ld, dtype=torch.float64, device=dev)


def timed(fn, reps=5):
    fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


out = t.clone()
t_ref = timed(lambda: lib.gpn_lml_refine(s, 1, _ops._ptr(X), n, d, _ops._ptr(Y), None, 1, _ops._ptr(var), _ops._ptr(ls), 1, _ops._ptr(nz), _ops._ptr(f.A), f.ld,
                                         _ops._ptr(f.winv), _ops._ptr(f._refine_work), _ops._ptr(out)))
t_den = timed(lambda: lib.gpn_lml_refine_dense(s, _ops._ptr(f._ksave), f.ld, 0.0, n, _ops._ptr(Y), None, 1, _ops._ptr(f.A), f.ld, _ops._ptr(f.winv),
                                               _ops._ptr(f._refine_work), _ops._ptr(out)))
t_k1 = timed(lambda: _ops.kernel_matrix("Matern52", X, None, var, ls, noise=nz, out=K1, ldk=f.ld, lower=True))
print("C3 parts: gpn_lml_refine %.3f ms | gpn_lml_refine_dense on the saved copy %.3f ms | assembly (one output) %.3f ms" % (t_ref, t_den, t_k1))

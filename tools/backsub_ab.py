#!/usr/bin/env python3
"""GPU-box A/B (tools' build): the refinement's back-substitution as ONE persistent launch (flags in device memory) against one
launch per 128-column block -- bit-identity of the refined terms and time of gpn_lml_refine.  backsub_ab.py [n,d,dy ...]"""
import ctypes, os, sys, time
import numpy as np
import torch
ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, ROOT)
from gptorch_amd import _native, _ops, rng  # noqa: E402
cases = [tuple(int(v) for v in a.split(",")) for a in sys.argv[1:]] or [(12288, 8, 1), (16001, 8, 3), (32768, 16, 1), (700, 3, 5), (129, 2, 1)]
dev = torch.device("cuda:0")
lib = _native.debug_begin()
lib.gpn_debug_set_backsub_persistent.restype = ctypes.c_int
lib.gpn_debug_set_backsub_persistent.argtypes = [ctypes.c_int]
for n, d, dy in cases:
    x, y = rng.make_regression(n, d, dy, seed=0)
    X, Y = torch.as_tensor(x).to(dev), torch.as_tensor(y).to(dev)
    var = torch.tensor([1.0], dtype=torch.float64, device=dev)
    ls = torch.tensor([float(np.sqrt(d))], dtype=torch.float64, device=dev)
    nz = torch.tensor([1e-2], dtype=torch.float64, device=dev)
    res = {}
    for mode in (0, 1, 0, 1):
        lib.gpn_debug_set_backsub_persistent(mode)
        f, terms = _ops.lml_forward("Matern52", X, Y, var, ls, nz, refine=True)
        torch.cuda.synchronize()
        ts = []
        work = f._refine_work
        for it in range(6):
            out = terms.clone()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            st = lib.gpn_lml_refine(_ops._stream(dev), _ops.KINDS["Matern52"], _ops._ptr(X), n, d, _ops._ptr(Y), None, dy, _ops._ptr(var), _ops._ptr(ls), 1,
                                    _ops._ptr(nz), _ops._ptr(f.A), f.ld, _ops._ptr(f.winv), _ops._ptr(work), _ops._ptr(out))
            torch.cuda.synchronize()
            ts.append(time.perf_counter() - t0)
            assert st == 0
        res.setdefault(mode, []).append((min(ts[1:]), terms.clone()))
    same = all(torch.equal(res[0][0][1], r[1]) for rs in res.values() for r in rs)
    print("N %6d D %2d dy %d: refine with one launch per block %.3f ms | persistent %.3f ms | terms bit-identical %s  (LML %.10f)"
          % (n, d, dy, min(r[0] for r in res[0]) * 1e3, min(r[0] for r in res[1]) * 1e3, same, res[1][0][1][2].item()), flush=True)

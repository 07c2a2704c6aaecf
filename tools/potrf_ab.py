#!/usr/bin/env python3
"""GPU-box tool: A/B of the factorisation drivers (0 = look-ahead panels, 1 = plain recursion)
on the bench workloads: ms per LML evaluation and the LML values."""
import os, sys, time
import torch
ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from gptorch_amd import _native  # noqa: E402

lib = _native.lib()
for wl in (sys.argv[1:] or ["c2", "c3"]):
    if wl.startswith("n="):          # ad-hoc size: Rbf, D = 8
        nn = int(wl[2:])
        w = dict(name=wl, kind="Rbf", n=nn, d=8, dy=1, variance=1.0, length_scales=8.0 ** 0.5, noise=1e-2)
    else:
        w = bench.WORKLOADS[wl]
    m, _, _ = bench.build_model(w, 0, torch.device("cuda:0"))
    for variant in [int(v, 0) for v in os.environ.get("VARIANTS", "1,0,1,0").split(",")]:
        _native.debug_begin().gpn_debug_set_potrf_variant(variant)
        with torch.no_grad():
            for _ in range(3):
                out = m.log_likelihood()
            torch.cuda.synchronize()
            reps = 20 if w["n"] <= 8192 else 5
            t0 = time.perf_counter()
            for _ in range(reps):
                out = m.log_likelihood()
            torch.cuda.synchronize()
        print("%s variant 0x%x: %.3f ms  lml %.10f" % (wl, variant, (time.perf_counter() - t0) / reps * 1e3, out.item()), flush=True)
    _native.debug_end()
    del m
    torch.cuda.empty_cache()

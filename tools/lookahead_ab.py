#!/usr/bin/env python3
"""GPU-box tool (round 6): look-ahead over OUTER panels with a persistent bulk (potrf.hip g_outer_lookahead) against the plain
schedule -- ms per LML evaluation, same box, interleaved; the factor buffer, the leaf inverses and the terms must be
bit-identical.  usage: lookahead_ab.py <n> [<n> ...] [--variants "mode:extra:nwg:pad ..."] [--d D] [--kind Rbf]   (tools' build)"""
import argparse, os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), "..")))
from gptorch_amd import _native, _ops, rng

ap = argparse.ArgumentParser()
ap.add_argument("sizes", type=int, nargs="+")
ap.add_argument("--variants", default="0:0:0:20 1:0:0:20")
ap.add_argument("--d", type=int, default=8)
ap.add_argument("--dy", type=int, default=1)
ap.add_argument("--kind", default="Rbf")
ap.add_argument("--passes", type=int, default=2)
args = ap.parse_args()
dev = torch.device("cuda:0")
lib = _native.debug_begin()
variants = [tuple(int(v) for v in s.split(":")) for s in args.variants.split()]
for n in args.sizes:
    x, y = rng.make_regression(n, args.d, args.dy, seed=0)
    X, Y = torch.as_tensor(x).to(dev), torch.as_tensor(y).to(dev)
    var = torch.tensor([1.0], dtype=torch.float64, device=dev)
    ls = torch.tensor([float(np.sqrt(args.d))], dtype=torch.float64, device=dev)
    nz = torch.tensor([1e-2], dtype=torch.float64, device=dev)
    ref = None
    reps = max(5, int(40 * (8192 / n) ** 3))
    for rep in range(args.passes):
        for v in variants:
            lib.gpn_debug_set_outer_lookahead(*v)
            f = None
            for _ in range(3):
                f, t = _ops.lml_forward(args.kind, X, Y, var, ls, nz, factor=f, refine=False)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(reps):
                f, t = _ops.lml_forward(args.kind, X, Y, var, ls, nz, factor=f, refine=False)
            torch.cuda.synchronize()
            dt = (time.perf_counter() - t0) / reps
            L = torch.tril(f.A[:n, :n]).clone()
            got = (L, f.A[n:n + args.dy, :n].clone(), f.winv.clone(), t.clone())
            same = "ref"
            if ref is None:
                ref = got
            else:
                same = "bitwise-equal" if all(torch.equal(a, b) for a, b in zip(ref, got)) else \
                    "DIFFERENT (max |dL| %.3e, |dterms| %.3e)" % ((ref[0] - got[0]).abs().max().item(), (ref[3] - got[3]).abs().max().item())
            print("n %6d  lookahead mode %2d extra %d nwg %3d pad %2d : %8.3f ms  lml %.10f  %s" % ((n,) + v + (dt * 1e3, t[2].item(), same)), flush=True)
            del L, got
lib.gpn_debug_set_outer_lookahead(-1, 0, 0, 20)
_native.debug_end()

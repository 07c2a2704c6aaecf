#!/usr/bin/env python3
"""GPU-box tool (round 6): gpn_potrf_lower_persistent (csrc/ppotrf.hip: the factorisation as one persistent dataflow launch)
against gpn_potrf_lower on the same matrix -- factor, extra rows, leaf inverses and info must be bit-identical; ms per
factorisation by HIP events around the call alone (the matrix is restored from a copy before each one).
usage: persistent_ab.py <n> [<n> ...] [--dy 1] [--reps R] [--chain R1,R2,...] [--grid G] [--soak K]   (tools' build for --chain / --grid)"""
import argparse, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), "..")))
from gptorch_amd import _native, _ops, rng

ap = argparse.ArgumentParser()
ap.add_argument("sizes", type=int, nargs="+")
ap.add_argument("--dy", type=int, default=1)
ap.add_argument("--d", type=int, default=8)
ap.add_argument("--reps", type=int, default=0)
ap.add_argument("--chain", default="0")
ap.add_argument("--grid", type=int, default=0)
ap.add_argument("--soak", type=int, default=0, help="extra persistent factorisations compared bitwise against the first")
args = ap.parse_args()
dev = torch.device("cuda:0")
dbg = args.chain != "0" or args.grid
lib = _native.debug_begin() if dbg else _native.lib()
st = _ops._stream(dev)
ptr = _ops._ptr


def run(f, saved, which, reps):
    e0 = [torch.cuda.Event(enable_timing=True) for _ in range(reps)]
    e1 = [torch.cuda.Event(enable_timing=True) for _ in range(reps)]
    for r in range(reps):
        f.A.copy_(saved)
        f.info.zero_()
        e0[r].record()
        if which == "persistent":
            rc = lib.gpn_potrf_lower_persistent(st, ptr(f.A), f.n, f.e, f.ld, ptr(f.winv), ptr(f.info))
        else:
            rc = lib.gpn_potrf_lower(st, ptr(f.A), f.n, f.e, f.ld, ptr(f.winv), ptr(f.info))
        assert rc == 0, (which, rc)
        e1[r].record()
    torch.cuda.synchronize()
    ts = sorted(a.elapsed_time(b) for a, b in zip(e0, e1))
    return ts[len(ts) // 2], ts[0]


for n in args.sizes:
    x, y = rng.make_regression(n, args.d, args.dy, seed=0)
    X, Y = torch.as_tensor(x).to(dev), torch.as_tensor(y).to(dev)
    var = torch.tensor([1.0], dtype=torch.float64, device=dev)
    ls = torch.tensor([float(np.sqrt(args.d))], dtype=torch.float64, device=dev)
    nz = torch.tensor([1e-2], dtype=torch.float64, device=dev)
    f = _ops.Factor(n, args.dy, dev)
    _ops.kernel_matrix("Rbf", X, None, var, ls, noise=nz, out=f.A, ldk=f.ld, lower=True)
    f.pack_rhs(Y)
    saved = f.A.clone()
    reps = args.reps or max(5, int(30 * (8192 / n) ** 3))
    if not lib.gpn_potrf_persistent_supported(n, args.dy):
        print("n %d: not a size of the persistent driver" % n)
        continue
    med, mn = run(f, saved, "launches", reps)
    ref = (torch.tril(f.A[:n, :n]).clone(), f.A[n:n + args.dy, :n].clone(), f.winv.clone(), int(f.info.item()))
    print("n %6d  gpn_potrf_lower            : median %8.3f ms  min %8.3f   info %d" % (n, med, mn, ref[3]), flush=True)
    for R in [int(v) for v in args.chain.split(",")]:
        if dbg:
            lib.gpn_debug_set_persistent(R, args.grid)
        run(f, saved, "persistent", 2)
        med, mn = run(f, saved, "persistent", reps)
        got = (torch.tril(f.A[:n, :n]), f.A[n:n + args.dy, :n], f.winv, int(f.info.item()))
        same = all(torch.equal(a, b) for a, b in zip(ref[:3], got[:3])) and ref[3] == got[3]
        msg = "bitwise-equal" if same else "DIFFERENT: max |dL| %.3e  max |d extra| %.3e  max |dW| %.3e  info %d" % (
            (ref[0] - got[0]).abs().max().item(), (ref[1] - got[1]).abs().max().item() if args.dy else 0.0,
            (ref[2] - got[2]).abs().max().item(), got[3])
        print("n %6d  persistent (chain wgs %2d)  : median %8.3f ms  min %8.3f   %s" % (n, R, med, mn, msg), flush=True)
        bad = 0
        for k in range(args.soak):
            run(f, saved, "persistent", 1)
            if not (torch.equal(ref[0], torch.tril(f.A[:n, :n])) and torch.equal(ref[1], f.A[n:n + args.dy, :n]) and torch.equal(ref[2], f.winv)):
                bad += 1
        if args.soak:
            print("n %6d  soak: %d of %d persistent factorisations differ from gpn_potrf_lower" % (n, bad, args.soak), flush=True)
    del f, saved, ref
if dbg:
    lib.gpn_debug_set_persistent(0, 0)
    _native.debug_end()

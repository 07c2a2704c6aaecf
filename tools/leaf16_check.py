#!/usr/bin/env python3
"""GPU-box tool: the second-generation 128x128 leaf (csrc/leaf16.hip) against torch -- factor and inverse at full and
ragged block sizes, LAPACK-style info on non-positive pivots, back-to-back timing against the first-generation leaf
(tools' build, variant bit 1) and its per-phase cycle budget (s_memtime stamps)."""
import os, sys
import numpy as np
import torch
ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, ROOT)
from gptorch_amd import _native, _ops
from gptorch_amd._ops import _ptr, _stream
dev = torch.device("cuda:0")
torch.manual_seed(0)


def spd(n, cond=1e3):
    a = torch.randn(n, n, dtype=torch.float64, device=dev)
    q, _ = torch.linalg.qr(a)
    ev = torch.logspace(0, -np.log10(cond), n, dtype=torch.float64, device=dev)
    m = (q * ev) @ q.t()
    return (m + m.t()) / 2


def run(lib, m):
    n = m.shape[0]
    f = _ops.Factor(n, 0, dev)
    f.A.zero_()
    f.A[:n, :n] = torch.tril(m) + torch.triu(torch.full_like(m, 777.0), 1)     # the upper triangle must never be read
    f.winv.fill_(float("nan"))
    f.info.zero_()
    rc = lib.gpn_potrf_lower(_stream(dev), _ptr(f.A), n, 0, f.ld, _ptr(f.winv), _ptr(f.info))
    torch.cuda.synchronize()
    assert rc == 0, rc
    return f


ok = True
with _native.debug_library() as lib:
    for gen in (2,):
        for n in (128, 127, 113, 100, 64, 33, 17, 16, 15, 1):
            for cond in (1e2, 1e8):
                m = spd(n, cond)
                f = run(lib, m)
                L = torch.linalg.cholesky(m)
                got = torch.tril(f.A[:n, :n])
                errL = ((got - L).abs().max() / L.abs().max()).item()
                iu = torch.triu_indices(n, n, 1, device=dev)
                up = (f.A[:n, :n][iu[0], iu[1]] - 777.0).abs().max().item() if n > 1 else 0.0
                W = f.winv[:128 * 128].reshape(128, 128)
                Wref = torch.zeros(128, 128, dtype=torch.float64, device=dev)
                Wref[:n, :n] = torch.linalg.inv(L)
                errW = ((W - Wref).abs().max() / Wref.abs().max()).item()
                resid = ((got @ got.t() - m).abs().max() / m.abs().max()).item()
                info = int(f.info.item())
                bad = not (errL < 1e-9 * max(1.0, cond ** 0.5) and errW < 1e-9 * max(1, cond ** 0.5) and resid < 1e-14 and up == 0.0 and info == 0)
                ok &= not bad
                print("gen %d n %3d cond %.0e  errL %.2e errW %.2e resid %.2e upper-touched %.1e info %d %s" % (gen, n, cond, errL, errW, resid, up, info, "BAD" if bad else ""))
        # failure: the pivot of column j goes negative
        for n, j in ((128, 0), (128, 5), (128, 16), (128, 77), (128, 127), (100, 99), (40, 17)):
            m = spd(n, 10.0)
            L = torch.linalg.cholesky(m)
            m2 = m.clone()
            m2[j, j] = (L[j, :j] ** 2).sum() - 0.5          # d_j = -0.5
            f = run(lib, m2)
            info = int(f.info.item())
            wz = f.winv[:128 * 128].abs().max().item()
            bad = info != j + 1 or wz != 0.0 or not torch.isfinite(f.A).all().item()
            ok &= not bad
            print("gen %d n %3d negative pivot at %3d -> info %d, |winv| %.1e, finite %s %s" % (gen, n, j, info, wz, torch.isfinite(f.A).all().item(), "BAD" if bad else ""))
        m = spd(128, 10.0); m[50, 3] = float("nan")
        f = run(lib, m)
        print("gen %d NaN entry -> info %d" % (gen, int(f.info.item())))
        ok &= int(f.info.item()) > 0
    # determinism
    m = spd(128, 1e6)
    f1, f2 = run(lib, m), run(lib, m)
    same = torch.equal(f1.A, f2.A) and torch.equal(f1.winv, f2.winv)
    ok &= same
    print("bitwise repeatable:", same)
    # timing, back to back
    for gen in (2, 2):
        R = 400
        m = spd(128, 1e3)
        fs = []
        for _ in range(R):
            f = _ops.Factor(128, 0, dev)
            f.A[:128, :128] = m
            fs.append(f)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for f in fs:
            lib.gpn_potrf_lower(_stream(dev), _ptr(f.A), 128, 0, f.ld, _ptr(f.winv), _ptr(f.info))
        e1.record()
        torch.cuda.synchronize()
        print("gen %d leaf: %.2f us per launch back to back" % (gen, e0.elapsed_time(e1) * 1e3 / R))
print("LEAF16 CHECK", "OK" if ok else "FAILED")
sys.exit(0 if ok else 1)

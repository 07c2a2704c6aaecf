#!/usr/bin/env python3
"""GPU-box parity sweep (round 5): random lock-step groups -- sizes on and around the blocking edges (16-pivot blocks, the
128-wide leaf, 256-column panels, the 256-row switch to the level-parallel inversion), all native stationary kinds, ARD /
isotropic, and (3 groups in 10) composite kernels of one structure (Linear + Rbf + Constant, Matern52 * Rbf-ARD, Matern32 + White),
dy 1..3, shared or per-model data, 2..6 models per group, two groups per call -- through
batched_loss_and_grad (gpn_lml_forward_batched + gpn_lml_backward_batched) against each model's own loss(); backward()
(the reference's optimiser-step closure, gptorch/models/base.py:260-269).  The bar is BITWISE equality of every loss and
every gradient.  No oracle involved (both sides are native); exits 1 on the first mismatch.
usage: fuzz_lockstep.py [cases = 40] [seed = 0]"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), "..", ".."))
sys.path.insert(0, ROOT)
from gptorch_amd import kernels, likelihoods, rng  # noqa: E402
from gptorch_amd.models import GPR, batched_loss_and_grad  # noqa: E402

cases = int(sys.argv[1]) if len(sys.argv) > 1 else 40
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 0
rs = np.random.RandomState(seed)
dev = torch.device("cuda:0")
KINDS = {"Rbf": kernels.Rbf, "Matern52": kernels.Matern52, "Matern32": kernels.Matern32, "Exp": kernels.Exp}
EDGES = [16, 64, 128, 129, 255, 256, 257, 384, 511, 512, 640, 1000, 1024, 1025, 1536, 2048, 2176]


def composite(d, which):
    if which == 0:        # the reference's example model (examples/regression_1d.py:34-53)
        return kernels.Linear(d, variance=float(0.1 + rs.rand())) + kernels.Rbf(d, length_scales=float(0.5 + rs.rand())) + kernels.Constant(d, variance=float(0.2 + rs.rand()))
    if which == 1:
        return kernels.Matern52(d, variance=float(0.5 + rs.rand()), length_scales=float(0.7 + rs.rand())) * kernels.Rbf(d, length_scales=(0.8 + rs.rand(d)) * 1.5, ARD=True)
    return kernels.Matern32(d, length_scales=float(0.7 + rs.rand())) + kernels.White(d, variance=float(0.01 + 0.05 * rs.rand()))


def group(n, d, dy, kind, ard, shared, count, base_seed, comp=-1):
    ms = []
    X0 = Y0 = None
    for b in range(count):
        if X0 is None or not shared:
            x, y = rng.make_regression(n, d, dy, seed=base_seed + (0 if shared else b))
            X, Y = torch.as_tensor(x).to(dev), torch.as_tensor(y).to(dev)
            if X0 is None:
                X0, Y0 = X, Y
        else:
            X, Y = X0, Y0
        ls = (0.6 + rs.rand(d)) * np.sqrt(d) if ard else float((0.6 + rs.rand()) * np.sqrt(d))
        kern = composite(d, comp) if comp >= 0 else KINDS[kind](d, variance=float(0.5 + rs.rand()), length_scales=ls, ARD=ard)
        m = GPR(X, Y, kern, likelihood=likelihoods.Gaussian(variance=float(10.0 ** rs.uniform(-2.5, -1.0))))
        m.cuda()
        m.X, m.Y = X, Y
        ms.append(m)
    return ms


bad = 0
for case in range(cases):
    models, desc = [], []
    for g in range(2):
        n = int(EDGES[rs.randint(len(EDGES))] + rs.randint(-2, 3) * (rs.rand() < 0.3))
        n = max(n, 8)
        d, dy = int(rs.randint(1, 7)), int(rs.randint(1, 4))
        kind = list(KINDS)[rs.randint(len(KINDS))]
        ard, shared, count = bool(rs.rand() < 0.4), bool(rs.rand() < 0.6), int(rs.randint(2, 7))
        comp = int(rs.randint(0, 3)) if rs.rand() < 0.3 else -1        # a composite-kernel group (_expr.BatchedExprLogLik)
        models += group(n, d, dy, kind, ard, shared, count, 1000 * case + 10 * g, comp)
        desc.append((n, d, dy, kind, ard, shared, count, comp))
    order = rs.permutation(len(models))
    models = [models[i] for i in order]                      # the two groups interleaved in the call
    ref = []
    for m in models:
        m.zero_grad()
        loss = m.loss()
        loss.backward()
        ref.append((loss.detach().clone(), [None if p.grad is None else p.grad.clone() for p in m.parameters()]))
        m.zero_grad()
    out = batched_loss_and_grad(models)
    for i, m in enumerate(models):
        ok = torch.equal(out[i], ref[i][0])
        for p, g in zip(m.parameters(), ref[i][1]):
            ok = ok and ((p.grad is None) == (g is None)) and (g is None or torch.equal(p.grad, g))
        if not ok:
            bad += 1
            print("MISMATCH case %d model %d groups %s" % (case, i, desc), flush=True)
            print("  loss", out[i].item(), ref[i][0].item(), [None if p.grad is None else p.grad.tolist() for p in m.parameters()], [None if g is None else g.tolist() for g in ref[i][1]])
            break
    if bad:
        break
print("cases %d (two lock-step groups each), violations %d" % (cases, bad))
sys.exit(1 if bad else 0)

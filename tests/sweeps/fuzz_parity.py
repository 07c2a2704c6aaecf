#!/usr/bin/env python3
"""GPU-box tool: randomized parity sweep against the CPU oracle -- sizes around every blocking edge
(leaf 128, panel 1024, tile 64), all native kinds, ARD / isotropic, dy 1..4, well- and ill-conditioned
noise.  Prints the worst relative errors; exits 1 on a violation."""
import os, sys
import numpy as np
import torch
ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), "..", ".."))
sys.path.insert(0, ROOT)
from gptorch_amd import kernels, likelihoods, rng  # noqa: E402
from gptorch_amd.models import GPR  # noqa: E402
from oracle import gp_oracle as orc  # noqa: E402

cases = int(sys.argv[1]) if len(sys.argv) > 1 else 120
rs = np.random.RandomState(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
edges = [1, 2, 63, 64, 65, 127, 128, 129, 255, 256, 257, 383, 384, 511, 512, 513, 1023, 1024, 1025, 1151, 1152, 1280, 2047, 2048, 2049, 2176]
worst = {"lml": 0.0, "grad": 0.0, "mean": 0.0, "var": 0.0}
bad = 0
for it in range(cases):
    n = int(rs.choice(edges)) if rs.rand() < 0.6 else int(rs.randint(1, 2600))
    d = int(rs.choice([1, 2, 3, 5, 8, 16, 17, 33]))
    dy = int(rs.choice([1, 1, 2, 4]))
    kind = str(rs.choice(["Rbf", "Matern52", "Matern32"]))
    ard = bool(rs.rand() < 0.5)
    noise = float(rs.choice([1e-3, 1e-2, 0.1]))
    x, y = rng.make_regression(n, d, dy, seed=1000 + it)
    ls = (np.sqrt(d) * (0.5 + rs.rand(d))) if ard else float(np.sqrt(d) * (0.6 + rs.rand()))
    m = GPR(x, y, getattr(kernels, kind)(d, variance=1.3, length_scales=ls, ARD=ard), likelihood=likelihoods.Gaussian(variance=noise))
    m.cuda()
    o = orc.GPROracle(x, y, kind=kind, variance=1.3, length_scales=ls, noise=noise, ARD=ard)
    lo = o.loss(); lo.backward()
    l = m.loss(); l.backward()
    e_l = abs(l.item() - lo.item()) / max(1.0, abs(lo.item()))
    e_g = 0.0
    for got, ref in [(m.kernel.variance.grad, o.raw_variance.grad), (m.kernel.length_scales.grad, o.raw_length_scales.grad),
                     (m.likelihood.variance.grad, o.raw_noise.grad)]:
        e_g = max(e_g, (got.cpu() - ref).abs().max().item() / max(1.0, ref.abs().max().item()))
    xs = rng.normal(5000 + it, (7, d))
    mu, var = m.predict_f(xs)
    with torch.no_grad():
        omu, ovar = o.predict_f(xs)
    e_m = np.abs(mu - omu.numpy()).max()
    e_v = np.abs(var - ovar.numpy()).max()
    worst["lml"] = max(worst["lml"], e_l); worst["grad"] = max(worst["grad"], e_g)
    worst["mean"] = max(worst["mean"], e_m); worst["var"] = max(worst["var"], e_v)
    if e_l > 1e-9 or e_g > 1e-6 or e_m > 1e-7 or e_v > 1e-8:
        bad += 1
        print("VIOLATION n=%d d=%d dy=%d %s ard=%s noise=%g: lml %.2e grad %.2e mean %.2e var %.2e" % (n, d, dy, kind, ard, noise, e_l, e_g, e_m, e_v), flush=True)
print("cases %d, violations %d, worst rel errors: %s" % (cases, bad, {k: "%.2e" % v for k, v in worst.items()}))
sys.exit(1 if bad else 0)

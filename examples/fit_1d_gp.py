#!/usr/bin/env python3
"""Switching a gptorch script to gptorch_amd: the workflow of the reference's 1-D regression example
(GPR over a sum kernel -- or a sparse VFE model --, scipy L-BFGS-B, predictions and posterior samples) on an MI355X.  Only the
import lines differ from a gptorch script, plus ONE line: `settings.auto_device = True` (or GPTORCH_AMD_AUTO_DEVICE=1 in the
environment) lets the CPU-constructed model place itself on the GPU at its first call, like the reference's example, which
never calls .cuda() (examples/regression_1d.py:89-95).  Without it `model.cuda()` is mandatory: there is no CPU path.

    python examples/fit_1d_gp.py [--sparse] [--n 100] [--restarts 6]

--restarts K (not in the reference, which fits one model per optimize() call): K exact-GP restarts with an Rbf kernel from
different initial length scales, all K L-BFGS-B runs AT ONCE -- every round of function evaluations is one lock-step
loss + backward on the GPU (gptorch_amd.models.multi_start_optimize), each restart ends where its own optimize() would --
and the best one predicts.
"""
import argparse
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))   # run from a checkout without installing

from gptorch_amd import kernels, settings  # was: from gptorch import kernels
from gptorch_amd.models import GPR, VFE    # was: from gptorch.models.gpr import GPR / sparse_gpr import VFE
from gptorch_amd.models import multi_start_optimize


def target(x):
    return np.sin(2.0 * np.pi * x) + np.cos(3.5 * np.pi * x) - 3.0 * x + 5.0


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--sparse", action="store_true", help="variational sparse GP (VFE) instead of the exact one")
    ap.add_argument("--n", type=int, default=100)
    ap.add_argument("--restarts", type=int, default=0, help="multi-start: this many Rbf restarts optimised in lock step")
    args = ap.parse_args()
    rs = np.random.RandomState(42)
    x = np.linspace(0.0, 1.0, args.n).reshape(-1, 1)
    y = target(x) + 0.1 * rs.randn(args.n, 1)

    if args.sparse:      # K(Z) of a sum with rank-one terms on 20 points is singular to working precision: Matern52 here
        model = VFE(x, y, kernels.Matern52(1), num_inducing_points=20)
    else:
        model = GPR(x, y, kernels.Linear(1) + kernels.Rbf(1) + kernels.Constant(1))
    settings.auto_device = True          # opt-in: the model moves itself to the GPU at its first loss() / predict call
    if args.restarts > 1 and not args.sparse:
        xs_, ys_ = torch.as_tensor(x).cuda(), torch.as_tensor(y).cuda()          # the restarts share the data on the device
        restarts = []
        for ell in np.geomspace(0.02, 2.0, args.restarts):
            m = GPR(xs_, ys_, kernels.Rbf(1, length_scales=float(ell)))
            m.cuda()
            restarts.append(m)
        results, seconds = multi_start_optimize(restarts, method="L-BFGS-B", max_iter=100)
        best = int(np.argmin([r.fun for r in results]))
        print("multi-start: %d restarts in %.2f s, final losses %s -> restart %d" % (args.restarts, seconds, [round(float(r.fun), 3) for r in results], best))
        model = restarts[best]
    else:
        model.optimize(method="L-BFGS-B", max_iter=100)
    print(model)

    x_test = np.linspace(-1.0, 2.0, 200).reshape(-1, 1)
    with torch.no_grad():
        mu, var = model.predict_y(x_test)
        samples = model.predict_y_samples(x_test, n_samples=5)
    inside = np.abs(mu[50:150] - target(x_test[50:150])) < 3.0 * np.sqrt(var[50:150])
    print("predictive mean within 3 sigma of the truth on [0.25, 1.25]: %.0f %%; samples %s"
          % (100.0 * inside.mean(), samples.shape))


if __name__ == "__main__":
    main()

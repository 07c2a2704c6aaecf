#!/usr/bin/env python3
"""k-fold cross-validation of a GP regression model with every fold in lock step (needs an MI355X).

The reference fits and scores one model at a time (gptorch/models/base.py:260-269 per fold, gpr.py:88-117 per prediction).  Here
the k training sets -- of UNEQUAL length when N is not a multiple of k -- form one ragged lock-step group: every iteration of the
fit is one loss + backward over all folds, the factorisations the predictions start from are one more lock-step call, and every
number is bit-identical to fitting and scoring the folds one after the other.
Usage: python examples/cross_validation.py [N=3001] [k=5] [iterations=30]"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), "..")))
from gptorch_amd import kernels, rng  # noqa: E402
from gptorch_amd.models import GPR, batched_factorise, multi_start_optimize  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 3001
k = int(sys.argv[2]) if len(sys.argv) > 2 else 5
iters = int(sys.argv[3]) if len(sys.argv) > 3 else 30
d = 3
x, y = rng.make_regression(n, d, 1, seed=0)
order = np.random.default_rng(0).permutation(n)
held = np.array_split(order, k)                                  # folds of n // k and n // k + 1 rows
splits = [(np.setdiff1d(order, te), te) for te in held]

folds = [GPR(x[tr], y[tr], kernels.Rbf(d, length_scales=float(np.sqrt(d)))) for tr, _ in splits]
for m in folds:
    m.cuda()
losses, seconds = multi_start_optimize(folds, method="Adam", max_iter=iters, learning_rate=0.05)
print("fitted %d folds (%s training rows) in lock step: %.2f s, final losses %s"
      % (k, ", ".join(str(len(tr)) for tr, _ in splits), seconds, np.round(losses[:, -1], 3).tolist()))
batched_factorise(folds)
rmse = []
for m, (_, te) in zip(folds, splits):
    mean, var = m.predict_y(x[te])
    rmse.append(float(np.sqrt(np.mean((mean - y[te]) ** 2))))
print("held-out RMSE per fold: %s   mean %.4f" % (np.round(rmse, 4).tolist(), float(np.mean(rmse))))

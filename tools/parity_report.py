#!/usr/bin/env python3
"""GPU-box diagnostic: abs error of LML / predictions vs the goldens for every
case, and a first timing of the LML pipeline stages at the bench sizes."""
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, ROOT)
from gptorch_amd import kernels, likelihoods, mean_functions, rng, _ops  # noqa: E402
from gptorch_amd.models import GPR  # noqa: E402

KERN = {"Rbf": kernels.Rbf, "Matern52": kernels.Matern52}


def model(case):
    x, y = rng.make_regression(case["n"], case["d"], case["dy"], seed=0)
    d = x.shape[1]
    ls = case["length_scales"]
    if case["ARD"]:
        ls = np.asarray(ls, dtype=np.float64) * np.ones(d)
    kern = KERN[case["kind"]](d, variance=case["variance"], length_scales=ls, ARD=case["ARD"])
    mean = None
    if case.get("mean") is not None:
        mean = mean_functions.Constant(y.shape[1], val=torch.tensor(case["mean"], dtype=torch.float64))
        mean.val.requires_grad_(False)
    m = GPR(x, y, kern, likelihood=likelihoods.Gaussian(variance=case["noise"]), mean_function=mean)
    m.cuda()
    return m


def main():
    cases = json.load(open(os.path.join(ROOT, "tests/golden/lml_cases.json")))
    c3 = os.path.join(ROOT, "tests/golden/lml_c3.json")
    if os.path.exists(c3) and "--big" in sys.argv:
        cases.append(json.load(open(c3)))
    for case in cases:
        m = model(case)
        with torch.no_grad():
            lml = -m.loss().item()
            torch.cuda.synchronize()
            t0 = time.time()
            reps = 3
            for _ in range(reps):
                m.loss()
            torch.cuda.synchronize()
            dt = (time.time() - t0) / reps
        line = "%-22s n=%6d lml=%.10f err=%.3e rel=%.2e  %.3f ms/eval" % (
            case["name"], case["n"], lml, abs(lml - case["lml"]), abs(lml - case["lml"]) / abs(case["lml"]), dt * 1e3)
        if "predict" in case:
            xs = rng.normal(case["predict"]["seed_xs"], (16, case["d"]))
            mf, vf = m.predict_f(xs)
            _, cf = m.predict_f(xs, diag=False)
            line += "  pred err mean=%.2e var=%.2e cov=%.2e" % (
                np.max(np.abs(mf - np.asarray(case["predict"]["mean_f"]))),
                np.max(np.abs(vf - np.asarray(case["predict"]["var_f"]))),
                np.max(np.abs(cf - np.asarray(case["predict"]["cov_f"]))))
        print(line, flush=True)


if __name__ == "__main__":
    main()

#!/usr/bin/env python3
"""GPU-box tool: the native VFE bound at the C5-shaped quarter size against its extended-precision value
(tests/golden/vfe_extended_262144_2048.json) -- relative distance, next to the CPU oracle's own."""
import json, os, sys
import torch
ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, ROOT)
from gptorch_amd import kernels, likelihoods, mean_functions, rng  # noqa: E402
from gptorch_amd.models import VFE  # noqa: E402
case = json.load(open(os.path.join(ROOT, "tests", "golden", "vfe_extended_262144_2048.json")))
x, y = rng.make_regression(case["n"], case["d"], 1, seed=case["seed_x"])
z = rng.normal(case["seed_z"], (case["m"], case["d"]))
m = VFE(x, y, kernels.Rbf(case["d"], variance=case["variance"], length_scales=case["length_scales"]), inducing_points=z,
        likelihood=likelihoods.Gaussian(variance=case["noise"]), mean_function=mean_functions.Zero(1))
m.cuda()
with torch.no_grad():
    elbo, st = m._bound(m.X)
e = elbo.item()
print("native %.10f extended %.10f: rel %.3e (abs %.3e); cpu oracle rel %.3e; rung %d / %d"
      % (e, case["elbo_extended"], abs(e - case["elbo_extended"]) / abs(case["elbo_extended"]), abs(e - case["elbo_extended"]),
         case["oracle_rel_err_vs_extended"], st.f_uu.jitter_rung, case["jitter_rung"]))

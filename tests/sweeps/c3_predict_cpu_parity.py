#!/usr/bin/env python3
"""GPU-box one-off: GPR._predict (gpr.py:88-117) at BASELINE configs[2]'s size (C3: N = 32768, D = 16, Matern52), 1024 test points:
predictive mean and variance (diag) and a 64 x 64 full covariance by the CPU oracle on the box's host cores against the
native path on the same data.  (The goldens generated from the reference stop at N = 8192.)
    python tests/sweeps/c3_predict_cpu_parity.py [threads]        -> one JSON line"""
import json
import os
import subprocess
import sys
import time
ROOT = os.environ.get("GRAFT_REPO_ROOT", os.path.abspath(os.path.join(os.path.dirname(__file__), "..", "..")))
sys.path.insert(0, ROOT)

CHILD = r'''
import sys, time, json, resource, numpy as np, torch
sys.path.insert(0, %(root)r)
from oracle import gp_oracle as orc
from gptorch_amd import rng
w = json.loads(%(w)r)
torch.set_num_threads(%(th)d)
x, y = rng.make_regression(w["n"], w["d"], w["dy"], seed=0)
xs = rng.normal(7, (1024, w["d"]))
o = orc.GPROracle(x, y, kind=w["kind"], variance=w["variance"], length_scales=w["length_scales"], noise=w["noise"])
t0 = time.time()
with torch.no_grad():
    mean, var = o.predict_f(xs, diag=True)
    m2, cov = o.predict_f(xs[:64], diag=False)
np.savez(%(out)r, mean=mean.numpy(), var=var.numpy(), cov=cov.numpy())
print("C3P_CHILD " + json.dumps({"seconds": time.time() - t0, "peak_rss_gb": resource.getrusage(resource.RUSAGE_SELF).ru_maxrss / 1e6, "threads": %(th)d}))
'''


def main():
    threads = int(sys.argv[1]) if len(sys.argv) > 1 else 64
    import bench
    import numpy as np
    import torch
    from gptorch_amd import rng
    w = bench.WORKLOADS[os.environ.get("WORKLOAD", "c3")]
    m, _, _ = bench.build_model(w, 0, torch.device("cuda:0"))
    xs = rng.normal(7, (1024, w["d"]))
    with torch.no_grad():
        mean, var = m.predict_f(xs, diag=True)
        _, cov = m.predict_f(xs[:64], diag=False)
    mean, var, cov = [np.asarray(t.cpu() if hasattr(t, "cpu") else t) for t in (mean, var, cov)]
    del m
    torch.cuda.empty_cache()
    out_npz = os.environ.get("C3P_OUT", "/tmp/c3_predict_oracle.npz")      # (the oracle's values: tests/golden/predict_c3_cpu_oracle.npz)
    env = dict(os.environ, HIP_VISIBLE_DEVICES="", ROCR_VISIBLE_DEVICES="")
    spec = json.dumps({k: w[k] for k in ("n", "d", "dy", "kind", "variance", "length_scales", "noise")})
    out = subprocess.run([sys.executable, "-c", CHILD % {"root": ROOT, "w": spec, "th": threads, "out": out_npz}], capture_output=True, text=True,
                         timeout=2400, env=env)
    r = None
    for ln in out.stdout.splitlines():
        if ln.startswith("C3P_CHILD "):
            r = json.loads(ln[len("C3P_CHILD "):])
    if r is None:
        sys.exit("cpu child failed (%d): %s" % (out.returncode, out.stderr[-500:]))
    ref = np.load(out_npz)
    print(json.dumps({"workload": w["name"].replace("LML eval", "predict_f at 1024 points (+ 64 x 64 full covariance)"),
                      "mean_max_abs_diff": float(np.abs(mean - ref["mean"]).max()), "mean_max_abs": float(np.abs(ref["mean"]).max()),
                      "var_max_abs_diff": float(np.abs(var - ref["var"]).max()), "var_range": [float(ref["var"].min()), float(ref["var"].max())],
                      "cov_max_abs_diff": float(np.abs(cov - ref["cov"]).max()),
                      "cpu_seconds": r["seconds"], "cpu_threads": r["threads"], "cpu_peak_rss_gb": r["peak_rss_gb"]}), flush=True)


if __name__ == "__main__":
    main()

#!/usr/bin/env python3
"""GPU-box one-off: BASELINE configs[4] (C5: sparse VFE GP, Rbf, N = 10^6, M = 4096, D = 8) -- the collapsed bound evaluated ONCE
by the CPU oracle (oracle/gp_oracle.py VFEOracle = sparse_gpr.py:108-153 op for op; about 105 GB of host memory, a minute or
two on 64 threads) and by the streamed GPU path of models/sparse_gpr.py, on the same data.  The reference cannot hold the
M x N matrices in the build container (64 GB); the GPU boxes' hosts can.
    python tests/sweeps/c5_cpu_parity.py [threads]        -> one JSON line"""
import json
import os
import subprocess
import sys
import time
ROOT = os.environ.get("GRAFT_REPO_ROOT", os.path.abspath(os.path.join(os.path.dirname(__file__), "..", "..")))
sys.path.insert(0, ROOT)

N, M, D = int(os.environ.get("C5_N", 1000000)), 4096, 8
CHILD = r'''
import sys, time, json, resource, numpy as np, torch
sys.path.insert(0, %(root)r)
from oracle import gp_oracle as orc
from gptorch_amd import rng
n, m, d, th = %(n)d, %(m)d, %(d)d, %(th)d
torch.set_num_threads(th)
x, y = rng.make_regression(n, d, 1, seed=0)
z = rng.normal(99, (m, d))
o = orc.VFEOracle(x, y, z, kind="Rbf", variance=1.0, length_scales=float(np.sqrt(d)), noise=1e-2)
t0 = time.time()
with torch.no_grad():
    v = o.log_likelihood().item()
print("C5_CHILD " + json.dumps({"elbo": v, "seconds": time.time() - t0, "peak_rss_gb": resource.getrusage(resource.RUSAGE_SELF).ru_maxrss / 1e6, "threads": th}))
'''


def main():
    threads = int(sys.argv[1]) if len(sys.argv) > 1 else 64
    import bench
    need = 3.3 * 8.0 * M * N / 1e9 + 8
    avail = bench.host_mem_available_gb()
    if avail is not None and avail < need:
        sys.exit("host memory: %.0f GB available, %.0f GB needed" % (avail, need))
    import numpy as np
    import torch
    from gptorch_amd import kernels, likelihoods, mean_functions, rng
    from gptorch_amd.models import VFE
    xv, yv = rng.make_regression(N, D, 1, seed=0)
    z = rng.normal(99, (M, D))
    mod = VFE(xv, yv, kernels.Rbf(D, variance=1.0, length_scales=float(np.sqrt(D))), inducing_points=z,
              likelihood=likelihoods.Gaussian(variance=1e-2), mean_function=mean_functions.Zero(1))
    mod.cuda()
    with torch.no_grad():
        t0 = time.perf_counter()
        gpu = float(mod.log_likelihood().item())
        t_gpu = time.perf_counter() - t0
    del mod
    torch.cuda.empty_cache()
    env = dict(os.environ, HIP_VISIBLE_DEVICES="", ROCR_VISIBLE_DEVICES="")
    out = subprocess.run([sys.executable, "-c", CHILD % {"root": ROOT, "n": N, "m": M, "d": D, "th": threads}], capture_output=True, text=True,
                         timeout=2400, env=env)
    r = None
    for ln in out.stdout.splitlines():
        if ln.startswith("C5_CHILD "):
            r = json.loads(ln[len("C5_CHILD "):])
    if r is None:
        sys.exit("cpu child failed (%d): %s" % (out.returncode, out.stderr[-500:]))
    print(json.dumps({"workload": "C5: sparse VFE GP, Rbf, N=%d, M=%d, D=%d fp64: collapsed bound" % (N, M, D), "gpu_elbo": gpu,
                      "gpu_first_eval_s": t_gpu, "cpu_oracle_elbo": r["elbo"], "cpu_seconds": r["seconds"], "cpu_threads": r["threads"],
                      "cpu_peak_rss_gb": r["peak_rss_gb"], "abs_diff": abs(gpu - r["elbo"]), "rel_diff": abs(gpu - r["elbo"]) / abs(r["elbo"])}), flush=True)


if __name__ == "__main__":
    main()

#!/usr/bin/env python3
"""GPU-box one-off: BASELINE configs[2] (C3: N = 32768, D = 16, Matern52) -- d loss / d raw parameters by AUTOGRAD through the CPU
oracle's op chain (oracle/gp_oracle.py GPROracle.loss_and_grads: what the reference's CholeskyBackward0 + TriangularSolveBackward0
+ elementwise chain compute; about 115 GB of host memory, a few minutes on 64 threads) against the closed-form native backward
on the same data.  The reference cannot run autograd at this size in the build container (64 GB); the GPU boxes' hosts can.
    python tests/sweeps/c3_grad_cpu_parity.py [threads]        -> one JSON line"""
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
o = orc.GPROracle(x, y, kind=w["kind"], variance=w["variance"], length_scales=w["length_scales"], noise=w["noise"])
t0 = time.time()
loss, g = o.loss_and_grads()
print("C3G_CHILD " + json.dumps({"loss": loss.item(), "grad_loss": {"kernel.variance": g[0].tolist(), "kernel.length_scales": g[1].tolist(),
      "likelihood.variance": g[2].tolist()}, "seconds": time.time() - t0, "peak_rss_gb": resource.getrusage(resource.RUSAGE_SELF).ru_maxrss / 1e6,
      "threads": %(th)d}))
'''


def main():
    threads = int(sys.argv[1]) if len(sys.argv) > 1 else 64
    import bench
    w = bench.WORKLOADS[os.environ.get("WORKLOAD", "c3")]
    need = 14.0 * 8.0 * w["n"] ** 2 / 1e9
    avail = bench.host_mem_available_gb()
    if avail is not None and avail < need:
        sys.exit("host memory: %.0f GB available, %.0f GB needed" % (avail, need))
    import numpy as np
    import torch
    m, _, _ = bench.build_model(w, 0, torch.device("cuda:0"))
    t0 = time.perf_counter()
    loss = m.loss()
    loss.backward()
    torch.cuda.synchronize()
    t_gpu = time.perf_counter() - t0
    gpu = {"kernel.variance": m.kernel.variance.grad.tolist(), "kernel.length_scales": m.kernel.length_scales.grad.tolist(),
           "likelihood.variance": m.likelihood.variance.grad.tolist()}
    gl = float(loss.item())
    del m, loss
    torch.cuda.empty_cache()
    env = dict(os.environ, HIP_VISIBLE_DEVICES="", ROCR_VISIBLE_DEVICES="")
    spec = json.dumps({k: w[k] for k in ("n", "d", "dy", "kind", "variance", "length_scales", "noise")})
    out = subprocess.run([sys.executable, "-c", CHILD % {"root": ROOT, "w": spec, "th": threads}], capture_output=True, text=True, timeout=3000, env=env)
    r = None
    for ln in out.stdout.splitlines():
        if ln.startswith("C3G_CHILD "):
            r = json.loads(ln[len("C3G_CHILD "):])
    if r is None:
        sys.exit("cpu child failed (%d): %s" % (out.returncode, out.stderr[-500:]))
    rel = {k: float(np.max(np.abs(np.asarray(gpu[k]) - np.asarray(r["grad_loss"][k])) / np.maximum(1.0, np.abs(np.asarray(r["grad_loss"][k])))))
           for k in gpu}
    print(json.dumps({"workload": w["name"].replace("LML eval", "loss + backward"), "gpu_loss": gl, "gpu_grad_loss": gpu, "gpu_first_step_s": t_gpu,
                      "cpu_oracle_loss": r["loss"], "cpu_oracle_grad_loss": r["grad_loss"], "cpu_seconds": r["seconds"], "cpu_threads": r["threads"],
                      "cpu_peak_rss_gb": r["peak_rss_gb"], "loss_abs_diff": abs(gl - r["loss"]), "grad_rel_diff": rel}), flush=True)


if __name__ == "__main__":
    main()

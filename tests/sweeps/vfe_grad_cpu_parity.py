#!/usr/bin/env python3
"""GPU-box one-off: the streamed closed-form backward of the VFE bound (models/sparse_gpr.py: row chunks of 65536, two pipelines,
split-K accumulation, inducing-point gradients) at a size where those mechanisms are all in play -- N = 262144 (four chunks),
M = 2048, D = 8, ARD Rbf -- against AUTOGRAD through the CPU oracle's op chain (oracle/gp_oracle.py vfe_grads_autograd =
what loss().backward() gives the reference) on the box's host cores.  The reference-generated gradient goldens stop at
N = 3000, M = 200 (tests/golden/vfe_cases.json).
    python tests/sweeps/vfe_grad_cpu_parity.py [threads]        -> one JSON line"""
import json
import os
import subprocess
import sys
import time
ROOT = os.environ.get("GRAFT_REPO_ROOT", os.path.abspath(os.path.join(os.path.dirname(__file__), "..", "..")))
sys.path.insert(0, ROOT)
N, M, D = int(os.environ.get("VFE_N", 262144)), int(os.environ.get("VFE_M", 2048)), 8
LS = [1.2, 1.4, 1.6, 1.8, 2.0, 2.2, 2.4, 2.6]
VAR, NOISE = 1.3, 0.05

CHILD = r'''
import sys, time, json, resource, numpy as np, torch
sys.path.insert(0, %(root)r)
from oracle import gp_oracle as orc
from gptorch_amd import rng
n, m, d, th = %(n)d, %(m)d, %(d)d, %(th)d
torch.set_num_threads(th)
x, y = rng.make_regression(n, d, 1, seed=0)
z = rng.normal(99, (m, d))
o = orc.VFEOracle(x, y, z, kind="Rbf", variance=%(var)r, length_scales=np.asarray(%(ls)r), noise=%(noise)r)
t0 = time.time()
with torch.no_grad():
    elbo = o.log_likelihood().item()
g = orc.vfe_grads_autograd(o)
c = orc.vfe_closed_form_grads(o)          # the closed form the native backward implements, in fp64 on the CPU
np.savez(%(out)r, elbo=elbo, g_variance=g[0].numpy(), g_length_scales=g[1].numpy(), g_noise=g[2].numpy(), g_Z=g[3].numpy(),
         c_variance=c[0].numpy(), c_length_scales=c[1].numpy(), c_noise=c[2].numpy(), c_Z=c[3].numpy())
print("VFEG_CHILD " + json.dumps({"seconds": time.time() - t0, "peak_rss_gb": resource.getrusage(resource.RUSAGE_SELF).ru_maxrss / 1e6, "threads": th}))
'''


def main():
    threads = int(sys.argv[1]) if len(sys.argv) > 1 else 64
    import numpy as np
    import torch
    from gptorch_amd import kernels, likelihoods, mean_functions, rng
    from gptorch_amd.models import VFE
    xv, yv = rng.make_regression(N, D, 1, seed=0)
    z = rng.normal(99, (M, D))
    mod = VFE(xv, yv, kernels.Rbf(D, variance=VAR, length_scales=np.asarray(LS), ARD=True), inducing_points=z,
              likelihood=likelihoods.Gaussian(variance=NOISE), mean_function=mean_functions.Zero(1))
    mod.cuda()
    loss = mod.loss()
    loss.backward()
    torch.cuda.synchronize()
    got = {"elbo": -float(loss.item()), "g_variance": mod.kernel.variance.grad.cpu().numpy().ravel(),
           "g_length_scales": mod.kernel.length_scales.grad.cpu().numpy().ravel(), "g_noise": mod.likelihood.variance.grad.cpu().numpy().ravel(),
           "g_Z": mod.Z.grad.cpu().numpy()}
    del mod, loss
    torch.cuda.empty_cache()
    out_npz = os.environ.get("VFEG_OUT", "/tmp/vfe_grad_oracle.npz")
    env = dict(os.environ, HIP_VISIBLE_DEVICES="", ROCR_VISIBLE_DEVICES="")
    out = subprocess.run([sys.executable, "-c", CHILD % {"root": ROOT, "n": N, "m": M, "d": D, "th": threads, "var": VAR, "ls": LS, "noise": NOISE,
                                                          "out": out_npz}], capture_output=True, text=True, timeout=3000, env=env)
    r = None
    for ln in out.stdout.splitlines():
        if ln.startswith("VFEG_CHILD "):
            r = json.loads(ln[len("VFEG_CHILD "):])
    if r is None:
        sys.exit("cpu child failed (%d): %s" % (out.returncode, out.stderr[-600:]))
    ref = np.load(out_npz)
    # the oracle differentiates the bound w.r.t. the CONSTRAINED values; the model's .grad is d loss / d raw = -value * that
    want = {"g_variance": -VAR * ref["g_variance"].ravel(), "g_length_scales": -np.asarray(LS) * ref["g_length_scales"].ravel(),
            "g_noise": -NOISE * ref["g_noise"].ravel(), "g_Z": -ref["g_Z"]}
    rel = {k: float(np.abs(got[k] - want[k]).max() / np.abs(want[k]).max()) for k in want}
    wantc = {"g_variance": -VAR * ref["c_variance"].ravel(), "g_length_scales": -np.asarray(LS) * ref["c_length_scales"].ravel(),
             "g_noise": -NOISE * ref["c_noise"].ravel(), "g_Z": -ref["c_Z"]}
    relc = {k: float(np.abs(got[k] - wantc[k]).max() / np.abs(wantc[k]).max()) for k in wantc}      # native vs the CPU closed form
    relac = {k: float(np.abs(want[k] - wantc[k]).max() / np.abs(wantc[k]).max()) for k in wantc}    # CPU autograd vs CPU closed form
    print(json.dumps({"workload": "VFE, ARD Rbf, N=%d, M=%d, D=%d fp64: loss().backward()" % (N, M, D), "gpu_elbo": got["elbo"],
                      "cpu_oracle_elbo": float(ref["elbo"]), "elbo_rel_diff": abs(got["elbo"] - float(ref["elbo"])) / abs(float(ref["elbo"])),
                      "grad_rel_diff_max_norm": rel, "native_vs_cpu_closed_form": relc, "cpu_autograd_vs_cpu_closed_form": relac, "cpu_seconds": r["seconds"], "cpu_threads": r["threads"], "cpu_peak_rss_gb": r["peak_rss_gb"]}), flush=True)


if __name__ == "__main__":
    main()

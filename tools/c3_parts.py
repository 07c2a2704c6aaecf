#!/usr/bin/env python3
"""GPU-box tool: the two data-dependent terms of the C3 LML (sum log L_ii, |alpha|^2) for several
factorisation-driver variants, next to sampled entries of K -- to see which term carries the
difference to the reference's value (tests/golden/lml_c3.json) and how far it moves with the
summation order."""
import json, os, sys, time
import torch
ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, ROOT)
from gptorch_amd import _native, _ops, rng  # noqa: E402
import bench  # noqa: E402

w = bench.WORKLOADS[sys.argv[1] if len(sys.argv) > 1 else "c3"]
dev = torch.device("cuda:0")
x, y = rng.make_regression(w["n"], w["d"], 1, seed=0)
X, Y = torch.tensor(x, device=dev), torch.tensor(y, device=dev)
t = lambda v: torch.tensor([v], dtype=torch.float64, device=dev)
var, ls, nz = t(w["variance"]), t(w["length_scales"]), t(w["noise"])
lib = _native.lib()
idx = [(0, 0), (1, 0), (100, 7), (w["n"] - 1, w["n"] - 2), (20000 % w["n"], 123), (5, 4), (31000 % w["n"], 30999 % w["n"])]
f = None
for name, v in [("default", 0), ("right-looking aux", 1 << 5), ("left-looking aux", 1 << 3), ("pw1024", 8 << 8), ("pw1536", 12 << 8),
                ("pw4096", 32 << 8), ("recursive", 1)]:
    _native.debug_begin().gpn_debug_set_potrf_variant(v)
    f, terms = _ops.lml_forward(w["kind"], X, Y, var, ls, nz, factor=f)
    torch.cuda.synchronize()
    t0 = time.time()
    f, terms = _ops.lml_forward(w["kind"], X, Y, var, ls, nz, factor=f)
    torch.cuda.synchronize()
    dt = time.time() - t0
    d = f.A.diagonal()[:w["n"]].log()
    a = f.extra()[0]
    import math
    print(json.dumps({"variant": name, "ms": dt * 1e3, "logdet_half": terms[0].item(), "quad": terms[1].item(), "lml": terms[2].item(),
                      "ld_fsum": math.fsum(d.cpu().tolist()), "quad_fsum": math.fsum((a.cpu().numpy() ** 2).tolist())}), flush=True)
_native.debug_end()
K = _ops.kernel_matrix(w["kind"], X[:40000], None, var, ls, noise=nz) if w["n"] <= 8192 else None
from gptorch_amd import kernels
ks = []
for i, j in idx:
    kij = _ops.kernel_matrix(w["kind"], X[i:i + 1], X[j:j + 1], var, ls)[0, 0].item() + (w["noise"] if i == j else 0.0)
    ks.append(kij)
print(json.dumps({"k_samples": ks}))

#!/usr/bin/env python3
"""GPU-box tool: same-box A/B of the look-ahead over panels (trailing-update bulk on a CU-masked
stream underneath the next panel's chain; gpn_debug_set_potrf_variant bits 20..27 = CUs kept free,
bit 6 = CU-mask layout).    python tools/panel_la_ab.py c2|c3|c4 [reps]"""
import json, os, sys, time
import torch
ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, ROOT)
from gptorch_amd import _native, _ops, rng  # noqa: E402
import bench  # noqa: E402

w = bench.WORKLOADS[sys.argv[1] if len(sys.argv) > 1 else "c3"]
reps = int(sys.argv[2]) if len(sys.argv) > 2 else (20 if w["n"] <= 8192 else 5)
variants = sys.argv[3].split(",") if len(sys.argv) > 3 else None
dev = torch.device("cuda:0")
x, y = rng.make_regression(w["n"], w["d"], 1, seed=0)
X, Y = torch.tensor(x, device=dev), torch.tensor(y, device=dev)
t = lambda v: torch.tensor([v], dtype=torch.float64, device=dev)
var, ls, nz = t(w["variance"]), t(w["length_scales"]), t(w["noise"])
lib = _native.lib()
f = None
cases = [("default", 0)]
for R in (2, 4, 8, 16, 32):
    for lay in (0, 1):
        cases.append(("la R=%d layout=%d" % (R, lay), (R << 20) | (lay << 6)))
cases.append(("la unmasked low-priority", 255 << 20))
cases.append(("default again", 0))
base = None
for name, v in cases:
    if variants and not any(k in name for k in variants):
        continue
    lib.gpn_debug_set_potrf_variant(v)
    for _ in range(2):
        f, terms = _ops.lml_forward(w["kind"], X, Y, var, ls, nz, factor=f)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        f, terms = _ops.lml_forward(w["kind"], X, Y, var, ls, nz, factor=f)
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / reps * 1e3
    lml = terms[2].item()
    if base is None:
        base = lml
    print(json.dumps({"workload": w["name"], "variant": name, "ms": round(ms, 3), "lml": lml, "same_bits_as_default": lml == base}), flush=True)
lib.gpn_debug_set_potrf_variant(0)

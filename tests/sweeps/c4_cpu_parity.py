#!/usr/bin/env python3
"""GPU-box one-off: BASELINE configs[3] (C4: N = 65536, D = 32, Rbf, fp64) evaluated ONCE by the CPU oracle on the box's host
cores (about 140 GB of host memory, a few minutes) and by the single-GPU path -- so that C4's value is pinned against the
reference's op sequence AT FULL SIZE, not only by the build's own agreement between one GPU and the block-cyclic grid
(tests/test_gpu_parity.py::test_c4_full_size_block_cyclic_2x4_grid).  The reference itself cannot hold C4 in the build
container (64 GB); the GPU boxes' hosts can (300 GiB cgroup).
    python tests/sweeps/c4_cpu_parity.py [threads]        -> one JSON line"""
import json
import os
import sys
import time
ROOT = os.environ.get("GRAFT_REPO_ROOT", os.path.abspath(os.path.join(os.path.dirname(__file__), "..", "..")))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
import bench  # noqa: E402
from gptorch_amd import _ops  # noqa: E402

w = bench.WORKLOADS[os.environ.get("WORKLOAD", "c4")]
threads = int(sys.argv[1]) if len(sys.argv) > 1 else 64
need = 4.5 * 8.0 * w["n"] ** 2 / 1e9
avail = bench.host_mem_available_gb()
if avail is not None and avail < need:
    sys.exit("host memory: %.0f GB available, %.0f GB needed" % (avail, need))
dev = torch.device("cuda:0")
m, _, _ = bench.build_model(w, 0, dev)
k = m.kernel
out = {"workload": w["name"]}
with torch.no_grad():
    t0 = time.perf_counter()
    out["gpu_lml_refined"] = float(m.log_likelihood().item())
    out["gpu_first_eval_s"] = time.perf_counter() - t0
    resid = m.Y - m.mean_function(m.X)
    f, terms = _ops.lml_forward(k._kind, m.X, resid, k.variance.transform(), k.length_scales.transform(),
                                m.likelihood.variance.transform(), refine=False)
    out["gpu_lml_plain"] = float(terms[2].item())
    out["gpu_half_logdet"] = float(terms[0].item())
del m, f
torch.cuda.empty_cache()
t0 = time.perf_counter()
r = bench.cpu_child(w, w["n"], threads, 0, 1, 2400.0)
out.update({"cpu_oracle_lml": r["lml"], "cpu_seconds": float(r["times"][0]), "cpu_threads": threads, "cpu_peak_rss_gb": r.get("peak_rss_gb"),
            "abs_diff_refined": abs(out["gpu_lml_refined"] - r["lml"]), "abs_diff_plain": abs(out["gpu_lml_plain"] - r["lml"])})
print(json.dumps(out), flush=True)

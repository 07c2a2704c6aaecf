"""GPU-box tool: host time to ENQUEUE one C2 evaluation (no synchronisation) vs its GPU time."""
import os, sys, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch, bench
from gptorch_amd import _ops, _native
lib = _native.lib()
dev = torch.device("cuda:0")
m, _, _ = bench.build_model(bench.WORKLOADS["c2"], 0, dev)
k = m.kernel
resid = (m.Y - m.mean_function(m.X)).contiguous()
var, ls, nz = k.variance.transform().detach(), k.length_scales.transform().detach(), m.likelihood.variance.transform().detach()
f = _ops.kernel_factor_async(k._kind, m.X, var, ls, nz, R=resid)
torch.cuda.synchronize()
for variant in (0, 1):
    _native.debug_begin().gpn_debug_set_potrf_variant(variant)
    ts = []
    for _ in range(8):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        _ops.kernel_factor_async(k._kind, m.X, var, ls, nz, R=resid, factor=f)
        t1 = time.perf_counter()
        torch.cuda.synchronize()
        t2 = time.perf_counter()
        ts.append(((t1 - t0) * 1e3, (t2 - t0) * 1e3))
    print("variant %d: host enqueue %.2f ms, total %.2f ms" % (variant, min(a for a, b in ts), min(b for a, b in ts)))
_native.debug_end()

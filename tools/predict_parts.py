import os, sys, time
import torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import bench
from gptorch_amd import rng, _ops
import gptorch_amd.models.gpr as gpr_mod
w = bench.WORKLOADS["c2"]
m, _, _ = bench.build_model(w, 0, torch.device("cuda:0"))
xs = torch.tensor(rng.normal(77, (1024, w["d"])), device="cuda:0")
with torch.no_grad():
    for _ in range(4):
        m._predict(xs)
    torch.cuda.synchronize()
    f, var, ls = m._factor_for_predict(m.X)
    def tm(fn, reps=20):
        fn(); torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(reps): fn()
        torch.cuda.synchronize(); return (time.perf_counter() - t0) / reps * 1e3
    print("gpr_predict blocked (ops level): %.3f ms" % tm(lambda: _ops.gpr_predict("Rbf", m.X, xs, var, ls, f, blocked=True)))
    print("_factor_for_predict: %.3f ms" % tm(lambda: m._factor_for_predict(m.X)))
    print("_predict: %.3f ms" % tm(lambda: m._predict(xs)))
    print("predict_y: %.3f ms" % tm(lambda: m.predict_y(xs)))
    # events around the C call only
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    ts = []
    for _ in range(10):
        e0.record(); _ops.gpr_predict("Rbf", m.X, xs, var, ls, f, blocked=True); e1.record(); torch.cuda.synchronize(); ts.append(e0.elapsed_time(e1))
    print("gpr_predict blocked, GPU time between events: %.3f ms (min %.3f)" % (sorted(ts)[5], min(ts)))

import os, sys, time, cProfile, pstats
import torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import bench
from gptorch_amd import rng, _ops
import gptorch_amd.models.gpr as gpr_mod
gpr_mod.INVERSE_AFTER_CALLS = 10 ** 9
w = bench.WORKLOADS["c2"]
m, _, _ = bench.build_model(w, 0, torch.device("cuda:0"))
xs = torch.tensor(rng.normal(77, (1024, w["d"])), device="cuda:0")
with torch.no_grad():
    for _ in range(3):
        m._predict(xs)
    torch.cuda.synchronize()
    f, var, ls = m._factor_for_predict(m.X)
    def tm(fn, reps=5):
        fn(); torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(reps): fn()
        torch.cuda.synchronize(); return (time.perf_counter() - t0) / reps * 1e3
    print("gpr_predict chain: %.3f ms" % tm(lambda: _ops.gpr_predict("Rbf", m.X, xs, var, ls, f)))
    print("_factor_for_predict: %.3f ms" % tm(lambda: m._factor_for_predict(m.X)))
    print("_predict: %.3f ms" % tm(lambda: m._predict(xs)))
    print("predict_y: %.3f ms" % tm(lambda: m.predict_y(xs)))
    pr = cProfile.Profile(); pr.enable()
    for _ in range(3): m.predict_y(xs)
    torch.cuda.synchronize(); pr.disable()
    pstats.Stats(pr).sort_stats("cumulative").print_stats(14)

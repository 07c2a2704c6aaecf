import os, sys, time, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import bench
from gptorch_amd import dist as gdist, rng
dev = torch.device("cuda:0")
for wl in ("c3", "c4"):
    w = bench.WORKLOADS[wl]
    x, y = rng.make_regression(w["n"], w["d"], 1, seed=0)
    X, Y = torch.tensor(x, device=dev), torch.tensor(y, device=dev)
    t = lambda v: torch.tensor([v], dtype=torch.float64, device=dev)
    g = gdist.BlockCyclicGP(X, Y, w["kind"], tile=2048)
    for refine in (False, True, False, True):
        g.refine = refine
        g.log_likelihood(t(w["variance"]), t(w["length_scales"]), t(w["noise"]), Y); torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(3):
            lml = g.log_likelihood(t(w["variance"]), t(w["length_scales"]), t(w["noise"]), Y)
        torch.cuda.synchronize()
        print(wl, "world-1 block-cyclic engine, refine=%s: %.2f ms  lml %.10f  golden diff %s" % (refine, (time.perf_counter() - t0) / 3 * 1e3, lml.item(),
              "%.3e" % abs(lml.item() - bench.golden_lml(w)) if bench.golden_lml(w) else "-"), flush=True)
    del g
    torch.cuda.empty_cache()

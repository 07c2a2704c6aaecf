import os, sys, time, json, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import bench
from gptorch_amd import dist as gdist, rng
dev = torch.device("cuda:0")
for wl in ("c3", "c4"):
    w = bench.WORKLOADS[wl]
    x, y = rng.make_regression(w["n"], w["d"], 1, seed=0)
    X, Y = torch.tensor(x, device=dev), torch.tensor(y, device=dev)
    t = lambda v: torch.tensor([v], dtype=torch.float64, device=dev)
    c = gdist.NativeDistLML(X, Y, w["kind"], tile=2048)
    for refine in (False, True):
        c.refine = refine
        c.log_likelihood(t(w["variance"]), t(w["length_scales"]), t(w["noise"])); torch.cuda.synchronize()
        t0 = time.perf_counter()
        lml = c.log_likelihood(t(w["variance"]), t(w["length_scales"]), t(w["noise"]))
        torch.cuda.synchronize()
        print("%s C driver (1x1 grid) refine=%s: %.1f ms  lml %.10f  |lml - golden| %.3e" % (wl, refine, (time.perf_counter() - t0) * 1e3, lml.item(), abs(lml.item() - bench.golden_lml(w))), flush=True)
    del c
    torch.cuda.empty_cache()

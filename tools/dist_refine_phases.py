"""GPU-box tool: where BlockCyclicGP._refine spends its time on one GPU (world 1): host-timed phases with a device sync after each
(the serial sweep over tile rows is what more ranks do NOT shorten; the tile inversions and the residual pass divide by the ranks)."""
import os, sys, time, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import bench
from gptorch_amd import dist as gdist, rng, _ops
dev = torch.device("cuda:0")
wl = sys.argv[1] if len(sys.argv) > 1 else "c4"
w = bench.WORKLOADS[wl]
x, y = rng.make_regression(w["n"], w["d"], 1, seed=0)
X, Y = torch.tensor(x, device=dev), torch.tensor(y, device=dev)
t = lambda v: torch.tensor([v], dtype=torch.float64, device=dev)
g = gdist.BlockCyclicGP(X, Y, w["kind"], tile=2048)
g.refine = False
var, ls, nz = t(w["variance"]), t(w["length_scales"]), t(w["noise"])
g.log_likelihood(var, ls, nz, Y)
ops = g.ops
marks = {}
orig = {k: getattr(ops, k) for k in ("tile_inverse", "gemv_t_acc", "resid_part", "refine_finish")}
def wrap(name):
    def f(*a, **k):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        r = orig[name](*a, **k)
        torch.cuda.synchronize()
        marks[name] = marks.get(name, 0.0) + time.perf_counter() - t0
        return r
    return f
for rep in range(3):
    marks.clear()
    for k in orig: setattr(ops, k, wrap(k))
    torch.cuda.synchronize(); t0 = time.perf_counter()
    g._refine(var, ls, nz, Y)
    torch.cuda.synchronize(); tot = time.perf_counter() - t0
    for k in orig: setattr(ops, k, orig[k])
    torch.cuda.synchronize(); t0 = time.perf_counter()
    g._refine(var, ls, nz, Y)
    torch.cuda.synchronize(); free = time.perf_counter() - t0
    print("%s world-1 _refine: %.2f ms un-instrumented; with a sync around every op %.2f ms: %s; rest (python, slicing, small torch ops) %.2f ms" % (
        wl, free * 1e3, tot * 1e3, ", ".join("%s %.2f" % (k, v * 1e3) for k, v in marks.items()), (tot - sum(marks.values())) * 1e3), flush=True)

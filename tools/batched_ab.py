import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from gptorch_amd import _native, _ops, rng
dev = torch.device("cuda:0")
n, d, B = 8192, 8, 8
x, y = rng.make_regression(n, d, 1, seed=0)
X, Y = torch.as_tensor(x).to(dev), torch.as_tensor(y).to(dev)
var = torch.linspace(1.0, 1.1, B, dtype=torch.float64, device=dev)
ls = (float(np.sqrt(d)) * torch.linspace(1.0, 1.2, B, dtype=torch.float64, device=dev))[:, None]
nz = torch.full((B,), 1e-2, dtype=torch.float64, device=dev)
with _native.debug_library() as lib:
    for name, v in (("shipped", 0), ("PW 384", 3 << 8), ("PW 512", 4 << 8), ("PW 512 left", 8 | (4 << 8)), ("PW 768", 6 << 8), ("PW 768 left", 8 | (6 << 8)), ("PW 1024", 8 << 8), ("PW 1024 left", 8 | (8 << 8)), ("PW 1536 left", 8 | (12 << 8)), ("shipped", 0)):
        lib.gpn_debug_set_potrf_variant(v)
        fb = None
        for it in range(2):
            fb, t = _ops.lml_forward_batched("Rbf", X, Y, var, ls, nz, fb=fb)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for it in range(5):
            fb, t = _ops.lml_forward_batched("Rbf", X, Y, var, ls, nz, fb=fb)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / 5
        print("%-18s %.2f ms per batch of %d = %.1f evals/s   lml[0] %.10f" % (name, dt * 1e3, B, B / dt, t[0, 2].item()))

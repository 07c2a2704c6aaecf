#!/usr/bin/env python3
"""GPU-box tool: what the reference's op sequence costs on this GPU through PyTorch-ROCm's GENERIC
ops (torch.mm / exp / linalg.cholesky / solve_triangular = rocBLAS + rocSOLVER + elementwise
kernels) -- i.e. gptorch after model.cuda() -- next to the native path.  Context only."""
import os, sys, time, math
import torch
ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), "..", ".."))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from oracle import gp_oracle as orc  # noqa: E402  (checker/tool use only)

dev = torch.device("cuda:0")
for wl in (sys.argv[1:] or ["c2"]):
    w = bench.WORKLOADS[wl]
    m, x, y = bench.build_model(w, 0, dev)
    X, Y = torch.tensor(x, device=dev), torch.tensor(y, device=dev)
    var = torch.tensor([w["variance"]], dtype=torch.float64, device=dev)
    ls = torch.tensor([w["length_scales"]], dtype=torch.float64, device=dev)
    nz = torch.tensor([w["noise"]], dtype=torch.float64, device=dev)
    n = X.shape[0]

    def generic(backward=False):
        v, l, s = var.clone().requires_grad_(backward), ls.clone().requires_grad_(backward), nz.clone().requires_grad_(backward)
        K = orc.kernel_K(w["kind"], X, None, v, l) + s * torch.eye(n, dtype=torch.float64, device=dev)
        L = torch.linalg.cholesky(K)
        a = torch.linalg.solve_triangular(L, Y, upper=False)
        lml = -0.5 * a.pow(2).sum() - L.diagonal().log().sum() - 0.5 * n * math.log(2 * math.pi)
        if backward:
            (-lml).backward()
        return lml

    def native(backward=False):
        if backward:
            m.zero_grad(); loss = m.loss(); loss.backward(); return -loss
        with torch.no_grad():
            return m.log_likelihood()

    def t(fn, reps=3):
        fn(); torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(reps):
            r = fn()
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / reps * 1e3, float(r)
    for bw in (False, True):
        if bw and wl == "c3":
            continue
        tg, lg = t(lambda: generic(bw))
        tn, ln = t(lambda: native(bw))
        print("%s %-13s generic PyTorch-ROCm ops: %9.2f ms   native: %8.2f ms   (x%.1f)   lml %.6f / %.6f" % (
            wl, "loss+backward" if bw else "LML forward", tg, tn, tg / tn, lg, ln), flush=True)

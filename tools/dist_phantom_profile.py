#!/usr/bin/env python3
"""GPU-box tool: where the compute side of ONE rank of a larger grid goes (phantom rank: launches and sizes of
rank r of a w-rank grid, collectives skipped) -- library profiler classes + wall.
    python tools/dist_phantom_profile.py <rank> <world> [n d tile]"""
import ctypes, os, sys, time
import torch
ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, ROOT)
from gptorch_amd import _native, dist as gdist, rng  # noqa: E402
import bench  # noqa: E402

r, w = int(sys.argv[1]), int(sys.argv[2])
n, d, T = (int(a) for a in (sys.argv[3:6] + ["65536", "32", "2048"][len(sys.argv) - 3:]))
dev = torch.device("cuda:0")
x, y = rng.make_regression(n, d, 1, seed=0)
X, Y = torch.tensor(x, device=dev), torch.tensor(y, device=dev)
g = gdist.BlockCyclicGP(X, Y, "Rbf", tile=T, phantom=(r, w))
one = torch.ones(1, dtype=torch.float64, device=dev)
ls = one * float(d) ** 0.5
lib = _native.lib()
for it in range(3):
    if it == 2:
        lib.gpn_profile_enable(1)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    g.assemble(one, ls, 0.01 * one, Y)
    torch.cuda.synchronize()
    t1 = time.perf_counter()
    g.factor()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
cls = bench.collect_classes(lib)
lib.gpn_profile_enable(0)
names = ["gemm rect (updates, trsm)", "gemm lower (diag tiles, in-tile syrk)", "panel solves (in place)", "gemm tri", "kmat", "grad", "leaf"]
print("phantom rank %d/%d grid %dx%d N=%d T=%d: assemble %.1f ms, factor %.1f ms (with per-launch events)" % (r, w, g.pr, g.pc, n, T, (t1 - t0) * 1e3, (t2 - t1) * 1e3))
for nm, (cnt, ms, work) in zip(names, cls):
    if cnt:
        print("   %-40s %6d launches %9.2f ms  %7.2f %s" % (nm, cnt, ms, work / (ms * 1e-3) / 1e12 if "gemm" in nm or "solve" in nm else work / (ms * 1e-3) / 1e9, "TFLOP/s executed" if "gemm" in nm or "solve" in nm else "GB/s"))

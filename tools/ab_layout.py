import os, sys, subprocess
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch, time
import bench
from gptorch_amd import _native, _ops
lib = _native.lib()
dev = torch.device("cuda:0")
def gemm(M, N, K, lower, reps=5):
    A = torch.randn(M + 16, K, dtype=torch.float64, device=dev)
    B = A if lower else torch.randn(N + 16, K, dtype=torch.float64, device=dev)
    C = torch.zeros(M, N, dtype=torch.float64, device=dev)
    out = {}
    for rnd in range(3):
        for v in (0x40, 0):
            lib.gpn_debug_set_gemm_variant(v)
            _ops.gemm_nt(A, B, M, N, K, alpha=-1.0, beta=1.0, C=C, lower=lower)
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(reps):
                _ops.gemm_nt(A, B, M, N, K, alpha=-1.0, beta=1.0, C=C, lower=lower)
            e1.record(); torch.cuda.synchronize()
            ms = e0.elapsed_time(e1) / reps
            fl = (M * (M + 1) if lower else 2.0 * M * N) * K
            out.setdefault(v, []).append(fl / ms / 1e9)
    print("M=%d N=%d K=%d lower=%d: legacy %s | new %s" % (M, N, K, lower, ["%.2f" % x for x in out[0x40]], ["%.2f" % x for x in out[0]]), flush=True)
def sweep(M, N, K, lower, reps=4):
    A = torch.randn(M + 16, K, dtype=torch.float64, device=dev)
    B = A if lower else torch.randn(N + 16, K, dtype=torch.float64, device=dev)
    C = torch.zeros(M, N, dtype=torch.float64, device=dev)
    out = {}
    for rnd in range(2):
        for v in (4, 9, 3):       # 64 quads | 64 lines | 128 lines
            lib.gpn_debug_set_gemm_variant(v)
            _ops.gemm_nt(A, B, M, N, K, alpha=-1.0, beta=1.0, C=C, lower=lower)
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(reps):
                _ops.gemm_nt(A, B, M, N, K, alpha=-1.0, beta=1.0, C=C, lower=lower)
            e1.record(); torch.cuda.synchronize()
            ms = e0.elapsed_time(e1) / reps
            fl = (M * (M + 1) if lower else 2.0 * M * N) * K
            out.setdefault(v, []).append(fl / ms / 1e9)
    print("M=%6d N=%6d K=%5d lower=%d: 64 quads %s | 64 lines %s | 128 lines %s" % (M, N, K, lower, ["%.2f" % x for x in out[4]], ["%.2f" % x for x in out[9]], ["%.2f" % x for x in out[3]]), flush=True)
for sh in [(30720, 30720, 1024, 1), (16384, 16384, 4096, 1), (7168, 7168, 1024, 1), (4096, 4096, 1024, 1), (2048, 2048, 1024, 1), (61440, 61440, 1024, 1),
           (8192, 8192, 4096, 0), (16384, 16384, 2048, 1), (8192, 8192, 2048, 1), (65536, 4096, 4096, 0), (4096, 4096, 65536, 1), (8192, 128, 128, 0), (8192, 1024, 128, 0)]:
    sweep(*sh)
lib.gpn_debug_set_gemm_variant(0)

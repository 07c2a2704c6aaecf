#!/usr/bin/env python3
"""GPU-box experiment: can a factorisation (latency-bound chain on a HIGH-priority stream) run
concurrently with throughput work (short-workgroup contractions on a normal/low-priority
stream) without CU masks?  Prints the LML-evaluation latency alone / with background and the
background's throughput alone / with the evaluation in flight."""
import os, sys, time
import torch
ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from gptorch_amd import _ops  # noqa: E402

dev = torch.device("cuda:0")
w = bench.WORKLOADS["c2"]
m, _, _ = bench.build_model(w, 0, dev)
print("priority range", torch.cuda.Stream.priority_range() if hasattr(torch.cuda.Stream, "priority_range") else "n/a")
hi = torch.cuda.Stream(device=dev, priority=-1)
lo = torch.cuda.Stream(device=dev, priority=0)
KB = int(sys.argv[1]) if len(sys.argv) > 1 else 256
MB = 6144
A = torch.randn(MB + 16, KB, dtype=torch.float64, device=dev)
C = torch.zeros(MB, MB, dtype=torch.float64, device=dev)
bg_flops = MB * (MB + 1.0) * KB


def bg(n):
    with torch.cuda.stream(lo):
        for _ in range(n):
            _ops.gemm_nt(A, A, MB, MB, KB, alpha=-1.0, beta=1.0, C=C, lower=True)


def fwd(n):
    with torch.cuda.stream(hi), torch.no_grad():
        for _ in range(n):
            m.log_likelihood()


def wall(fn):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    fn()
    torch.cuda.synchronize()
    return time.perf_counter() - t0


fwd(3); bg(3)
t_f = wall(lambda: fwd(10)) / 10
nb = 400
t_b = wall(lambda: bg(nb)) / nb
print("alone: eval %.3f ms; background launch %.1f us = %.1f TFLOP/s" % (t_f * 1e3, t_b * 1e6, bg_flops / t_b / 1e12))
# together: enqueue background first (it keeps the GPU busy for ~nb*t_b), then the evaluations
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
t0 = time.perf_counter()
bg(nb)
with torch.cuda.stream(hi):
    e0.record()
fwd(10)
with torch.cuda.stream(hi):
    e1.record()
torch.cuda.synchronize()
tot = time.perf_counter() - t0
t_f2 = e0.elapsed_time(e1) / 10
print("together: eval %.3f ms (x%.2f); all done in %.1f ms vs %.1f ms serial" % (t_f2, t_f2 / (t_f * 1e3), tot * 1e3, (nb * t_b + 10 * t_f) * 1e3))

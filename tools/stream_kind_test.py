"""GPU-box experiment: cost of event waits between created streams, per stream kind / priority."""
import os, sys, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch, bench
from gptorch_amd import _native
lib = _native.lib()
dev = torch.device("cuda:0")
m, _, _ = bench.build_model(bench.WORKLOADS["c2"], 0, dev)
def run(stream, variant):
    _native.debug_begin().gpn_debug_set_potrf_variant(variant)
    ctx = torch.cuda.stream(stream) if stream is not None else torch.cuda.stream(torch.cuda.current_stream())
    with ctx, torch.no_grad():
        for _ in range(3): m.log_likelihood()
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(10): m.log_likelihood()
        torch.cuda.synchronize()
    _native.debug_end()
    return (time.perf_counter() - t0) / 10 * 1e3
s_norm = torch.cuda.Stream(device=dev)
s_hi = torch.cuda.Stream(device=dev, priority=-1)
for name, st in (("default", None), ("created", s_norm), ("created-hi", s_hi)):
    print("%-11s look-ahead %.2f ms   recursion %.2f ms" % (name, run(st, 0), run(st, 1)), flush=True)

"""GPU-box tool: LML evals/s with R = 1..8 independent C2 models in flight -- default two internal
lanes, back to back on one stream, one free stream per model (batched_log_likelihood)."""
import os, sys, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch, bench
from gptorch_amd.models import batched_log_likelihood
dev = torch.device("cuda:0")
w = bench.WORKLOADS["c2"]
for R in (1, 2, 3, 4, 8):
    models = [bench.build_model(w, seed=50 + r, device=dev)[0] for r in range(R)]
    for label, streams in (("default", None), ("b2b", [torch.cuda.current_stream(dev)] * R),
                           ("streams", [torch.cuda.Stream(device=dev) for _ in range(R)])):
        for _ in range(2): batched_log_likelihood(models, streams)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(6): batched_log_likelihood(models, streams)
        torch.cuda.synchronize()
        print("R=%d %-8s %.1f evals/s" % (R, label, R * 6 / (time.perf_counter() - t0)), flush=True)
    del models

#!/usr/bin/env python3
"""
bench.py -- GP log-marginal-likelihood evaluations / second (BASELINE.json metric).

A "step" is ONE evaluation of the exact-GP log marginal likelihood through the
gptorch-compatible shell (model.log_likelihood()): fused K(X)+sigma_n^2 I assembly
-> blocked fp64 MFMA Cholesky with the residual carried as extra rows (forward
substitution) -> log-det / ||alpha||^2 reduction -> info check (jitter ladder).
Inputs are resident in HBM before the timed region.

Workload (config.workload): BASELINE.json configs[1] = "C2": GPR + Rbf, N=8192,
D=8, fp64, synthetic X~N(0,1), y=sin(sum x)+0.1 eps (gptorch_amd.rng, seed 0),
sigma^2=1, ell=sqrt(D), sigma_n^2=1e-2.  `--workload c3` runs N=32768 D=16 Matern52.

N>1 GPUs: one process per GPU, each evaluates its own independent model
(hyper-parameter restarts: rank r uses seed r) -- "replicas", no data-path
collective (multi-GPU); value = all ranks' evaluations / max-over-ranks time.

Extra objects on the JSON line:
  roofline     -- the fp64 MFMA contraction kernel (gemm_nt_kernel): algorithmic
                  flops of the factorisation it carries / summed HIP-event time of its
                  launches (events on the launch stream, second pass over the same steps)
  cpu_baseline -- the CPU oracle (torch-CPU restatement of the reference path,
                  oracle/gp_oracle.py) on this box's host cores, rank 0, N=1 only
"""
import argparse
import ctypes
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

WORKLOADS = {
    "c2": dict(name="C2: GPR+Rbf N=8192 D=8 fp64 LML eval", kind="Rbf", n=8192, d=8, dy=1,
               variance=1.0, length_scales=float(np.sqrt(8.0)), noise=1e-2),
    "c3": dict(name="C3: GPR+Matern52 N=32768 D=16 fp64 LML eval", kind="Matern52", n=32768, d=16, dy=1,
               variance=1.0, length_scales=4.0, noise=1e-2),
    "c4": dict(name="C4: GPR+Rbf N=65536 D=32 fp64 LML eval (one GPU: the 34 GB factor fits in HBM)", kind="Rbf", n=65536,
               d=32, dy=1, variance=1.0, length_scales=float(np.sqrt(32.0)), noise=1e-2),
    "c1": dict(name="C1: GPR+Rbf N=512 D=2 fp64 LML eval", kind="Rbf", n=512, d=2, dy=1,
               variance=1.0, length_scales=1.0, noise=1e-2),
}
PEAK_FP64_MFMA_TFLOPS = 78.6   # MI355X fp64 matrix peak (AMD spec; = 256 CU x 4 SIMD x 32 flop/clk x 2.4 GHz)


def build_model(w, seed, device):
    from gptorch_amd import kernels, likelihoods, rng
    from gptorch_amd.models import GPR
    x, y = rng.make_regression(w["n"], w["d"], w["dy"], seed=seed)
    kern = getattr(kernels, w["kind"])(w["d"], variance=w["variance"], length_scales=w["length_scales"])
    m = GPR(x, y, kern, likelihood=likelihoods.Gaussian(variance=w["noise"]))
    m.cuda()
    return m, x, y


def algorithmic_gemm_flops(n, dy):
    """Cholesky = N^3/3 flops; everything but the 128x128 diagonal leaves
    (N/128 * 128^3/3) runs in the contraction kernel, plus the fused solve N^2*dy."""
    return n ** 3 / 3.0 - n * 128.0 ** 2 / 3.0 + float(n) ** 2 * dy


def cpu_baseline(w, x, y, budget_s=25.0):
    from oracle import gp_oracle as orc
    o = orc.GPROracle(x, y, kind=w["kind"], variance=w["variance"], length_scales=w["length_scales"], noise=w["noise"])
    with torch.no_grad():
        t0 = time.time()
        o.log_likelihood()          # warm-up
        first = time.time() - t0
        reps = int(max(1, min(5, budget_s // max(first, 1e-3) - 1)))
        times = []
        for _ in range(reps):
            t0 = time.time()
            o.log_likelihood()
            times.append(time.time() - t0)
    med = float(np.median(times))
    return {"value": 1.0 / med, "unit": "LML evals/s", "cores": torch.get_num_threads(),
            "host_cpus": os.cpu_count(), "kind": "port",
            "sample": "%d full evaluations of the same workload (N=%d, D=%d) after 1 warm-up, median" % (reps, w["n"], w["d"]),
            "seconds_per_eval": med}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--workload", default="c2", choices=sorted(WORKLOADS))
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extras", action="store_true", help="skip the backward / concurrent-restart legs (profiling runs)")
    ap.add_argument("--test-shared-gpu", action="store_true",
                    help="(testing the multi-rank control flow on a 1-GPU box) every rank uses cuda:0, gloo collectives")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    distributed = world > 1
    if args.test_shared_gpu:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    if distributed:
        import torch.distributed as dist
        if args.test_shared_gpu:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=device)

    from gptorch_amd import _native
    lib = _native.lib()   # raises if the HIP library is missing: no fallback

    w = WORKLOADS[args.workload]
    model, x, y = build_model(w, seed=rank, device=device)

    def step():
        with torch.no_grad():
            return model.log_likelihood()

    def barrier():
        torch.cuda.synchronize()
        if distributed:
            dist.barrier()
            torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        out = step()
    barrier()
    elapsed = time.perf_counter() - t0
    if distributed:
        t = torch.tensor([elapsed], dtype=torch.float64, device="cpu" if args.test_shared_gpu else device)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = t.item()
    lml = out.item()

    # roofline leg: same steps again with a HIP event pair around every launch of the
    # contraction kernel (recorded on the stream the kernel is launched on)
    lib.gpn_profile_enable(1)
    for _ in range(args.steps):
        step()
    torch.cuda.synchronize()
    res = (ctypes.c_double * 3)()
    lib.gpn_profile_collect(res)
    lib.gpn_profile_enable(0)
    launches, gemm_ms, exec_flops = res[0], res[1], res[2]
    alg = algorithmic_gemm_flops(w["n"], w["dy"]) * args.steps
    achieved = alg / (gemm_ms * 1e-3) / 1e12 if gemm_ms > 0 else 0.0
    traffic = None
    tpath = os.path.join(ROOT, "profiles", "traffic_%s.json" % args.workload)
    if os.path.exists(tpath):   # HBM bytes per launch from the committed rocprofv3 --pmc passes (tools/pmc_traffic.py)
        traffic = json.load(open(tpath)).get("gemm_bytes_per_launch")
    roofline = {"bound": "mfma", "kernel": "gemm_nt_kernel (fp64 MFMA NT contraction: SYRK/GEMM trailing updates + panel solves)",
                "achieved": achieved, "peak": PEAK_FP64_MFMA_TFLOPS, "unit": "TFLOP/s", "frac": achieved / PEAK_FP64_MFMA_TFLOPS,
                "traffic": traffic,
                "launches_per_step": launches / args.steps, "avg_launch_us": gemm_ms * 1e3 / max(launches, 1),
                "algorithmic_flops_per_launch": alg / max(launches, 1),
                "executed_tflops": exec_flops / (gemm_ms * 1e-3) / 1e12 if gemm_ms > 0 else 0.0,
                "kernel_ms_per_step": gemm_ms / args.steps}

    # second roofline object (north_star: "achieved HBM GB/s on the distance sweep"): the fused
    # K(X)+noise*I assembly alone, lower triangle straight into the factor buffer, HIP events on
    # the launch stream; algorithmic bytes = 8 (N(N+1)/2 + N D)  
    kmat = None
    notes = {}
    if world == 1:
      try:
        from gptorch_amd import _ops
        k = model.kernel
        with torch.no_grad():
            var, ls, nz = k.variance.transform(), k.length_scales.transform(), model.likelihood.variance.transform()
            f = model._holder["factor"]
            reps = 20 if w["n"] <= 8192 else 5
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            _ops.kernel_matrix(k._kind, model.X, None, var, ls, noise=nz, out=f.A, ldk=f.ld, lower=True)
            e0.record()
            for _ in range(reps):
                _ops.kernel_matrix(k._kind, model.X, None, var, ls, noise=nz, out=f.A, ldk=f.ld, lower=True)
            e1.record()
            torch.cuda.synchronize()
        us = e0.elapsed_time(e1) / reps * 1e3
        nbytes = 8.0 * (w["n"] * (w["n"] + 1) / 2.0 + w["n"] * w["d"])
        ktraffic = None
        if os.path.exists(tpath):                  # same committed PMC passes as the contraction kernel's traffic
            ktraffic = json.load(open(tpath)).get("kmat_bytes_per_launch")
        kmat = {"bound": "hbm", "kernel": "kmat_kernel (fused distance + %s + noise, lower tiles)" % w["kind"],
                "achieved": nbytes / us / 1e3, "peak": 8000.0, "unit": "GB/s", "frac": nbytes / us / 1e3 / 8000.0,
                "traffic": ktraffic, "avg_launch_us": us, "algorithmic_bytes_per_launch": nbytes,
                "vector_flops_per_entry": 3 * w["d"] + 30,
                # "report both GB/s and vector-flop fraction" -- (3D+30) flop per entry
                "vector_tflops": (3 * w["d"] + 30) * (w["n"] * (w["n"] + 1) / 2.0) / us / 1e6,
                "vector_frac_of_fp64_peak": (3 * w["d"] + 30) * (w["n"] * (w["n"] + 1) / 2.0) / us / 1e6 / PEAK_FP64_MFMA_TFLOPS}
      except Exception as exc:      # an auxiliary leg must never cost the headline line
        notes["roofline_k_assembly_error"] = repr(exc)

    # extra (not part of `value`): one loss()+backward() step -- what Adam (base.py:260-269) pays
    # per iteration -- and the throughput with 4 independent restarts in flight on 4 HIP streams
    extra = {}
    if world == 1 and args.workload in ("c1", "c2") and not args.no_extras:
      try:
        torch.cuda.synchronize()
        for _ in range(2):
            model.zero_grad()
            model.loss().backward()
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        nb = max(2, args.steps // 4)
        for _ in range(nb):
            model.zero_grad()
            model.loss().backward()
        torch.cuda.synchronize()
        extra["loss_backward_ms"] = (time.perf_counter() - t1) / nb * 1e3
        from gptorch_amd.models import batched_log_likelihood
        R = 4
        models = [model] + [build_model(w, seed=100 + r, device=device)[0] for r in range(R - 1)]
        res = {}
        for label, streams in (("two_lanes", None), ("back_to_back", [torch.cuda.current_stream(device)] * R)):
            for _ in range(2):
                batched_log_likelihood(models, streams)
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            rounds = max(2, args.steps // 4)
            for _ in range(rounds):
                batched_log_likelihood(models, streams)
            torch.cuda.synchronize()
            res[label] = R * rounds / (time.perf_counter() - t1)
        extra["concurrent_restarts"] = {"restarts": R, "evals_per_s": res["two_lanes"],
                                        "evals_per_s_back_to_back": res["back_to_back"],
                                        "note": "R independent models alternating between two HIP streams "
                                                "(batched_log_likelihood), info read once per round; not the headline value"}
        del models
      except Exception as exc:
        notes["extras_error"] = repr(exc)

    if rank == 0:
        ms = elapsed / args.steps * 1e3
        line = {
            "metric": "GP log-marginal-likelihood evals/sec (Cholesky+solve) at NxD fp64",
            "value": world * args.steps / elapsed, "unit": "LML evals/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": ms,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f64", "data": "synthetic",
            "config": {"workload": w["name"], "N": w["n"], "D": w["d"], "dy": w["dy"], "kernel": w["kind"],
                       "parallelism": "replicas x%d (independent models, no collective)" % world},
            "lml": lml,
            "cholesky_frac_of_fp64_peak": (w["n"] ** 3 / 3.0) / (elapsed / args.steps) / 1e12 / PEAK_FP64_MFMA_TFLOPS,
            "roofline": roofline,
        }
        if kmat is not None:
            line["roofline_k_assembly"] = kmat
        line.update(extra)
        if world == 1 and not args.no_cpu_baseline:
            try:
                line["cpu_baseline"] = cpu_baseline(w, x, y)
            except Exception as exc:
                notes["cpu_baseline_error"] = repr(exc)
        if notes:
            line["notes"] = notes
        print(json.dumps(line), flush=True)
    if distributed:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()

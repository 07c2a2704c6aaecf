#!/usr/bin/env python3
"""
bench.py -- GP log-marginal-likelihood evaluations / second (BASELINE.json metric).

A "step" is ONE evaluation of the exact-GP log marginal likelihood (GPR.log_likelihood,
gptorch/models/gpr.py:47-67): fused K(X)+sigma_n^2 I assembly -> blocked fp64 MFMA Cholesky with
the residual carried as extra rows (forward substitution) -> log-det / |alpha|^2 reduction -> info
read-back (jitter ladder).  Inputs are resident in HBM before the timed region.

--gpus 1 (default): the headline is BASELINE.json configs[2] = "C3", the largest single-GPU
    configuration and the north-star target size: GPR + Matern52, N=32768, D=16, fp64 (synthetic
    X~N(0,1), y=sin(sum x)+0.1 eps from gptorch_amd.rng seed 0; sigma^2=1, ell=4, sigma_n^2=1e-2),
    through the gptorch-compatible shell (`model.log_likelihood()`).  Keyed extras carry the other
    single-GPU configs, each with its own config string: `c2` (N=8192, D=8, Rbf), `c4_1gpu`
    (N=65536, D=32: the 34 GB factor fits one GPU's HBM), `c5_vfe` (sparse VFE, N=1e6, M=4096) and
    `loss_backward` (one Adam step's loss()+backward()) at C2 and C3.
--gpus N>1 (launched by torch.distributed.run, one rank per GPU): STRONG scaling of
    configs[3] = "C4" (GPR + Rbf, N=65536, D=32): ONE model, its Gram matrix 2-D block-cyclic over
    all ranks (gptorch_amd/dist.py), panels exchanged by RCCL broadcasts on row / column
    sub-communicators.  value = evaluations/s of that single sharded model.  Rank 0 then times the
    same matrix on its own GPU alone (`single_gpu_same_run`) and every rank runs independent C2
    replicas (`replicas_c2`, labelled, not the headline).

Extra objects on the JSON line (N=1):
  roofline            -- fp64 MFMA contraction kernel (gemm_nt_kernel), ALL its launches of an
                         evaluation: algorithmic flops of the factorisation they carry / summed
                         HIP-event time (events on the launch stream, second pass over the same steps)
  roofline_syrk       -- only the SYRK trailing updates (the lower-tile K = panel-width contraction
                         at the end of every panel): north_star's ">= 50 % of fp64 MFMA peak at N=32768"
  roofline_k_assembly -- fused distance + kernel + noise assembly, HBM roofline (+ vector-flop model)
  cpu_baseline        -- the CPU oracle (torch-CPU restatement of the reference path,
                         oracle/gp_oracle.py) on this box's host cores: a bounded sample, see `sample`
"""
import argparse
import ctypes
import datetime
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
    "c1": dict(name="C1: GPR+Rbf N=512 D=2 fp64 LML eval", kind="Rbf", n=512, d=2, dy=1,
               variance=1.0, length_scales=1.0, noise=1e-2),
    "c2": dict(name="C2: GPR+Rbf N=8192 D=8 fp64 LML eval", kind="Rbf", n=8192, d=8, dy=1,
               variance=1.0, length_scales=float(np.sqrt(8.0)), noise=1e-2, golden=("lml_cases.json", "C2_rbf_8192_8")),
    "c3": dict(name="C3: GPR+Matern52 N=32768 D=16 fp64 LML eval", kind="Matern52", n=32768, d=16, dy=1,
               variance=1.0, length_scales=4.0, noise=1e-2, golden=("lml_c3.json", None)),
    "c4": dict(name="C4: GPR+Rbf N=65536 D=32 fp64 LML eval", kind="Rbf", n=65536,
               d=32, dy=1, variance=1.0, length_scales=float(np.sqrt(32.0)), noise=1e-2),
}
PEAK_FP64_MFMA_TFLOPS = 78.6   # MI355X fp64 matrix peak (AMD spec; = 256 CU x 4 SIMD x 32 flop/clk x 2.4 GHz; 77.6 measured)
PEAK_HBM_GBS = 8000.0
# profile.hip launch classes (gpn_common.h PROF_*)
P_GEMM, P_SYRK, P_SOLVE, P_TRI, P_KMAT, P_GRAD, P_LEAF, P_N = 0, 1, 2, 3, 4, 5, 6, 7


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


def algorithmic_syrk_flops(n, dy, pw):
    """the SYRK trailing updates of the look-ahead driver: after each panel of `pw` columns one
    lower-tile contraction over the m rows below it (incl. the dy residual rows), K = pw:
    m (m + 1) pw flops each (SURVEY 8(d) row 3 restricted to the launches that are timed)."""
    total, p0 = 0.0, 0
    while p0 < n:
        w = min(pw, n - p0)
        m = n + dy - (p0 + w)
        if p0 + w < n:
            total += m * (m + 1.0) * w
        p0 += w
    return total


def golden_lml(w):
    """the reference's LML for this workload from the committed fixtures (tests/golden/, generated
    by importing the reference: tests/golden/make_golden.py), or None."""
    g = w.get("golden")
    if not g:
        return None
    try:
        data = json.load(open(os.path.join(ROOT, "tests", "golden", g[0])))
        if g[1] is None:
            return float(data["lml"])
        return float([c for c in data if c["name"] == g[1]][0]["lml"])
    except Exception:
        return None


def collect_classes(lib):
    buf = (ctypes.c_double * (3 * P_N))()
    lib.gpn_profile_collect_classes(buf, P_N)
    return [(buf[3 * c], buf[3 * c + 1], buf[3 * c + 2]) for c in range(P_N)]


def timed(fn, steps, warmup):
    for _ in range(warmup):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        out = fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / steps, out


def cpu_baseline(w, x, y, budget_s=25.0):
    """the oracle on the host cores.  C2-sized problems run whole; for larger N a bounded sample --
    the first 8192 rows of the SAME data, same kernel and hyper-parameters -- is timed and scaled by
    (N/8192)^3 (the evaluation is Cholesky-bound: N^3/3 flops), labelled as an extrapolation."""
    from oracle import gp_oracle as orc
    ns = min(w["n"], 8192)
    o = orc.GPROracle(x[:ns], y[:ns], kind=w["kind"], variance=w["variance"], length_scales=w["length_scales"], noise=w["noise"])
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
    scale = (w["n"] / float(ns)) ** 3
    out = {"value": 1.0 / (med * scale), "unit": "LML evals/s", "cores": torch.get_num_threads(),
           "host_cpus": os.cpu_count(), "kind": "port", "seconds_per_eval": med * scale,
           "sample": "%d evaluations of N=%d rows of the same workload (D=%d, %s) after 1 warm-up, median %.3f s"
                     % (reps, ns, w["d"], w["kind"], med)}
    if ns != w["n"]:
        out["sample"] += "; value EXTRAPOLATED to N=%d by (N/%d)^3 = %.0fx (Cholesky-bound, N^3/3 flops)" % (w["n"], ns, scale)
        out["extrapolated"] = True
        out["measured_sample_evals_per_s"] = 1.0 / med
    return out


# ------------------------------------------------------------------------------------------------
# one GPU
# ------------------------------------------------------------------------------------------------
def rooflines(lib, w, steps, fn):
    """second pass over the same steps with a HIP-event pair around every launch of the profiled
    kernel classes (recorded on the stream each launch goes to)."""
    n, d, dy = w["n"], w["d"], w["dy"]
    lib.gpn_profile_enable(1)
    for _ in range(steps):
        fn()
    torch.cuda.synchronize()
    cls = collect_classes(lib)
    lib.gpn_profile_enable(0)
    out = {}
    launches = sum(cls[c][0] for c in (P_GEMM, P_SYRK, P_SOLVE, P_TRI))
    ms = sum(cls[c][1] for c in (P_GEMM, P_SYRK, P_SOLVE, P_TRI))
    exec_flops = sum(cls[c][2] for c in (P_GEMM, P_SYRK, P_SOLVE, P_TRI))
    alg = algorithmic_gemm_flops(n, dy) * steps
    ach = alg / (ms * 1e-3) / 1e12 if ms > 0 else 0.0
    out["roofline"] = {
        "bound": "mfma", "kernel": "gemm_nt_kernel (fp64 MFMA NT contraction: SYRK/GEMM trailing updates + panel solves), all launches",
        "achieved": ach, "peak": PEAK_FP64_MFMA_TFLOPS, "unit": "TFLOP/s", "frac": ach / PEAK_FP64_MFMA_TFLOPS,
        "traffic": None,
        "launches_per_step": launches / steps, "avg_launch_us": ms * 1e3 / max(launches, 1),
        "algorithmic_flops_per_launch": alg / max(launches, 1), "algorithmic_flops_per_step": alg / steps,
        "executed_tflops": exec_flops / (ms * 1e-3) / 1e12 if ms > 0 else 0.0,
        "kernel_ms_per_step": ms / steps,
        "by_class_ms_per_step": {"in_panel_updates": cls[P_GEMM][1] / steps, "syrk_trailing_updates": cls[P_SYRK][1] / steps,
                                 "panel_solves": cls[P_SOLVE][1] / steps, "leaf_128x128": cls[P_LEAF][1] / steps,
                                 "k_assembly": cls[P_KMAT][1] / steps}}
    pw = int(lib.gpn_potrf_panel_width(n))
    if pw and cls[P_SYRK][0]:
        salg = algorithmic_syrk_flops(n, dy, pw) * steps
        sms = cls[P_SYRK][1]
        sach = salg / (sms * 1e-3) / 1e12
        out["roofline_syrk"] = {
            "bound": "mfma", "kernel": "gemm_nt_kernel, lower-tile launches only = the SYRK trailing update after each %d-column panel" % pw,
            "achieved": sach, "peak": PEAK_FP64_MFMA_TFLOPS, "unit": "TFLOP/s", "frac": sach / PEAK_FP64_MFMA_TFLOPS, "traffic": None,
            "launches_per_step": cls[P_SYRK][0] / steps, "avg_launch_us": sms * 1e3 / cls[P_SYRK][0],
            "algorithmic_flops_per_launch": salg / cls[P_SYRK][0], "algorithmic_flops_per_step": salg / steps,
            "share_of_cholesky_flops": salg / steps / (n ** 3 / 3.0), "kernel_ms_per_step": sms / steps}
    if cls[P_KMAT][0]:
        kb, kms, kl = cls[P_KMAT][2], cls[P_KMAT][1], cls[P_KMAT][0]
        gbs = kb / (kms * 1e-3) / 1e9
        vflop = (3 * d + 30) * (n * (n + 1) / 2.0) * kl
        out["roofline_k_assembly"] = {
            "bound": "hbm", "kernel": "kmat_kernel (fused distance + %s + noise, lower tiles straight into the factor buffer)" % w["kind"],
            "achieved": gbs, "peak": PEAK_HBM_GBS, "unit": "GB/s", "frac": gbs / PEAK_HBM_GBS, "traffic": None,
            "avg_launch_us": kms * 1e3 / kl, "algorithmic_bytes_per_launch": kb / kl,
            "vector_flops_per_entry_model": 3 * d + 30,      # "report both GB/s and vector-flop fraction"
            "vector_tflops": vflop / (kms * 1e-3) / 1e12, "vector_frac_of_fp64_peak": vflop / (kms * 1e-3) / 1e12 / PEAK_FP64_MFMA_TFLOPS}
    return out


def attach_traffic(roof, key, workload):
    """HBM bytes per launch from the committed rocprofv3 --pmc passes (tools/pmc_traffic.py), tagged
    with the file they come from (they are NOT measured in this run)."""
    tpath = os.path.join(ROOT, "profiles", "traffic_%s.json" % workload)
    if roof is not None and os.path.exists(tpath):
        try:
            t = json.load(open(tpath))
            if t.get(key) is not None:
                roof["traffic"] = t[key]
                roof["traffic_source"] = "profiles/traffic_%s.json (separate --pmc passes%s)" % (
                    workload, ", " + t["round"] if "round" in t else "")
        except Exception:
            pass


def backward_leg(lib, model, w, steps):
    """loss()+backward() (what one Adam step of base.py:260-269 pays) + rooflines of its kernels:
    the K-clipped contractions (triangular inversion U = L^-T and Kyy^-1 = U U^T: 2 N^3/3 flops) on
    the MFMA roof, the gradient sweep (reads the lower triangle of Kyy^-1 once) on the HBM roof."""
    n = w["n"]

    def fb():
        model.zero_grad()
        model.loss().backward()

    t, _ = timed(fb, steps, 2)
    lib.gpn_profile_enable(1)
    for _ in range(steps):
        fb()
    torch.cuda.synchronize()
    cls = collect_classes(lib)
    lib.gpn_profile_enable(0)
    out = {"config": w["name"].replace("LML eval", "loss()+backward()"), "ms_per_step": t * 1e3,
           "whole_step_tflops_on_N3": n ** 3 / t / 1e12, "whole_step_frac_of_fp64_peak": n ** 3 / t / 1e12 / PEAK_FP64_MFMA_TFLOPS}
    tri_ms = cls[P_TRI][1] / steps
    if tri_ms > 0:
        ach = (2.0 * n ** 3 / 3.0) / (tri_ms * 1e-3) / 1e12
        out["roofline_backward_mfma"] = {
            "bound": "mfma", "kernel": "gemm_nt_kernel, K-clipped launches (U = L^-T by level-parallel inversion, Kyy^-1 = U U^T)",
            "achieved": ach, "peak": PEAK_FP64_MFMA_TFLOPS, "unit": "TFLOP/s", "frac": ach / PEAK_FP64_MFMA_TFLOPS, "traffic": None,
            "algorithmic_flops_per_step": 2.0 * n ** 3 / 3.0, "kernel_ms_per_step": tri_ms,
            "launches_per_step": cls[P_TRI][0] / steps, "executed_tflops": cls[P_TRI][2] / steps / (tri_ms * 1e-3) / 1e12}
    if cls[P_GRAD][0]:
        gms, gb = cls[P_GRAD][1], cls[P_GRAD][2]
        gbs = gb / (gms * 1e-3) / 1e9
        out["roofline_grad_sweep"] = {
            "bound": "hbm", "kernel": "grad_sweep_kernel (sum G o dK/dtheta, G formed on the fly from Kyy^-1 and a)",
            "achieved": gbs, "peak": PEAK_HBM_GBS, "unit": "GB/s", "frac": gbs / PEAK_HBM_GBS, "traffic": None,
            "avg_launch_us": gms * 1e3 / cls[P_GRAD][0], "algorithmic_bytes_per_launch": gb / cls[P_GRAD][0]}
    return out


def run_single(args, device):
    from gptorch_amd import _native
    lib = _native.lib()   # raises if the HIP library is missing: no fallback
    w = WORKLOADS[args.workload]
    model, x, y = build_model(w, seed=0, device=device)
    held = {"model": model}            # released before the large extra legs allocate
    del model
    notes, extra = {}, {}

    def step():
        with torch.no_grad():
            return held["model"].log_likelihood()

    for _ in range(args.warmup):
        step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        out = step()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    lml = out.item()

    roofs = rooflines(lib, w, args.steps, step)
    attach_traffic(roofs.get("roofline"), "gemm_bytes_per_launch", args.workload)
    attach_traffic(roofs.get("roofline_k_assembly"), "kmat_bytes_per_launch", args.workload)
    attach_traffic(roofs.get("roofline_syrk"), "syrk_bytes_per_launch", args.workload)

    if not args.no_extras:
        def leg(name, fn):
            try:
                extra[name] = fn()
            except Exception as exc:          # an auxiliary leg must never cost the headline line
                notes[name + "_error"] = repr(exc)
            torch.cuda.synchronize()
            torch.cuda.empty_cache()

        def lml_leg(key, steps, warm, with_backward):
            def run():
                ww = WORKLOADS[key]
                if key == args.workload:
                    m = held["model"]
                    res = {}
                else:
                    m, _, _ = build_model(ww, 0, device)

                    def st():
                        with torch.no_grad():
                            return m.log_likelihood()
                    t, o = timed(st, steps, warm)
                    res = {"config": ww["name"], "value": 1.0 / t, "unit": "LML evals/s", "ms_per_step": t * 1e3, "lml": o.item(),
                           "cholesky_frac_of_fp64_peak": (ww["n"] ** 3 / 3.0) / t / 1e12 / PEAK_FP64_MFMA_TFLOPS}
                    gl = golden_lml(ww)
                    if gl is not None:
                        res["lml_abs_err_vs_reference_golden"] = abs(res["lml"] - gl)
                    r2 = rooflines(lib, ww, steps, st)
                    for k2 in ("roofline", "roofline_syrk", "roofline_k_assembly"):
                        if k2 in r2:
                            res[k2] = r2[k2]
                if with_backward:
                    res["loss_backward"] = backward_leg(lib, m, ww, max(2, steps // 4) if ww["n"] <= 8192 else 2)
                if key == "c2":
                    # configs[1] reads "kernel build + Cholesky + predict": GPR.predict_f at 1024 test points
                    # (gpr.py:88-117) -- with the factor cached between calls, and re-factorising every call as
                    # the reference does (gpr.py:104)
                    from gptorch_amd import rng
                    xs = torch.tensor(rng.normal(7, (1024, ww["d"])), device=device)

                    def pred():
                        with torch.no_grad():
                            return m._predict(xs)

                    def pred_refactor():
                        m._predict_cache = None
                        return pred()
                    tp, _ = timed(pred, 10, 3)
                    tr, _ = timed(pred_refactor, 5, 1)
                    res["predict"] = {"config": "C2: GPR._predict at 1024 test points, diag variance", "ms_cached_factor": tp * 1e3,
                                      "ms_with_refactorisation": tr * 1e3}
                return res
            return run

        if args.workload == "c3":
            leg("c2", lml_leg("c2", 20, 3, True))
            leg("c3_backward", lml_leg("c3", 0, 0, True))
            if "c3_backward" in extra:
                extra["loss_backward"] = extra.pop("c3_backward")["loss_backward"]

            def restarts():
                from gptorch_amd.models import batched_log_likelihood
                R = 4
                models = [build_model(WORKLOADS["c2"], seed=100 + r, device=device)[0] for r in range(R)]
                res = {}
                for label, streams in (("two_lanes", None), ("back_to_back", [torch.cuda.current_stream(device)] * R)):
                    t, _ = timed(lambda: batched_log_likelihood(models, streams), 5, 2)
                    res[label] = R / t
                return {"config": "C2 x %d independent restarts alternating between two HIP streams (batched_log_likelihood)" % R,
                        "evals_per_s": res["two_lanes"], "evals_per_s_back_to_back": res["back_to_back"]}
            leg("c2_concurrent_restarts", restarts)
            held.clear()
            torch.cuda.empty_cache()
            leg("c4_1gpu", lml_leg("c4", 2, 1, False))

            def vfe():
                from gptorch_amd import kernels, likelihoods, mean_functions, rng
                from gptorch_amd.models import VFE
                n, m, d = 1000000, 4096, 8
                xv, yv = rng.make_regression(n, d, 1, seed=0)
                z = rng.normal(99, (m, d))
                mod = VFE(xv, yv, kernels.Rbf(d, variance=1.0, length_scales=float(np.sqrt(d))), inducing_points=z,
                          likelihood=likelihoods.Gaussian(variance=1e-2), mean_function=mean_functions.Zero(1))
                mod.cuda()

                def st():
                    with torch.no_grad():
                        return mod.log_likelihood()
                t, o = timed(st, 2, 1)
                fl = 2.0 * n * m * m + 2.0 * m ** 3 / 3.0
                return {"config": "C5: sparse VFE GP, Rbf, N=1e6, M=4096 inducing, D=8 fp64: collapsed-bound evaluation (sparse_gpr.py:108-151)",
                        "value": 1.0 / t, "unit": "bound evals/s", "ms_per_step": t * 1e3, "elbo": o.item(),
                        "tflops_on_2NM2_plus_2M3_3": fl / t / 1e12, "frac_of_fp64_peak": fl / t / 1e12 / PEAK_FP64_MFMA_TFLOPS}
            leg("c5_vfe", vfe)
        else:
            leg("loss_backward", lambda: backward_leg(lib, held["model"], w, max(2, args.steps // 4) if w["n"] <= 8192 else 2))

    ms = elapsed / args.steps * 1e3
    line = {
        "metric": "GP log-marginal-likelihood evals/sec (Cholesky+solve) at NxD fp64",
        "value": args.steps / elapsed, "unit": "LML evals/s",
        "n_gpus": 1, "steps": args.steps, "warmup": args.warmup, "ms_per_step": ms,
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "f64", "data": "synthetic",
        "config": {"workload": w["name"], "N": w["n"], "D": w["d"], "dy": w["dy"], "kernel": w["kind"], "parallelism": "1 GPU"},
        "lml": lml,
        "cholesky_frac_of_fp64_peak": (w["n"] ** 3 / 3.0) / (elapsed / args.steps) / 1e12 / PEAK_FP64_MFMA_TFLOPS,
    }
    gl = golden_lml(w)
    if gl is not None:
        line["lml_reference_golden"] = gl
        line["lml_abs_err_vs_reference_golden"] = abs(lml - gl)
    if args.workload == "c3":
        try:     # extended-precision value of the same expression (tests/golden/make_c3_extended.py)
            ext = json.load(open(os.path.join(ROOT, "tests", "golden", "lml_c3_extended.json")))
            line["lml_extended_precision"] = ext["lml_extended"]
            line["lml_abs_err_vs_extended_precision"] = abs(lml - ext["lml_extended"])
            line["reference_abs_err_vs_extended_precision"] = ext["reference_abs_err_vs_extended"]
        except Exception:
            pass
    line.update(roofs)
    line.update(extra)
    if not args.no_cpu_baseline:
        try:
            line["cpu_baseline"] = cpu_baseline(w, x, y)
        except Exception as exc:
            notes["cpu_baseline_error"] = repr(exc)
    if notes:
        line["notes"] = notes
    print(json.dumps(line), flush=True)


# ------------------------------------------------------------------------------------------------
# N > 1 GPUs: one model, block-cyclic over all ranks
# ------------------------------------------------------------------------------------------------
def run_multi(args, rank, local_rank, world, device):
    import torch.distributed as dist
    from gptorch_amd import _native, rng
    from gptorch_amd import dist as gdist
    lib = _native.lib()
    shared = args.test_shared_gpu
    cpu = torch.device("cpu")
    wkey = args.workload if args.workload_given else "c4"
    w = WORKLOADS[wkey]
    notes = {}

    def barrier():
        torch.cuda.synchronize()
        dist.barrier()
        torch.cuda.synchronize()

    def max_over_ranks(v):
        t = torch.tensor([v], dtype=torch.float64, device=cpu if shared else device)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return float(t.item())

    x, y = rng.make_regression(w["n"], w["d"], w["dy"], seed=0)          # the same model on every rank
    X, Y = torch.tensor(x, device=device), torch.tensor(y, device=device)
    var = torch.tensor([w["variance"]], dtype=torch.float64, device=device)
    ls = torch.tensor([w["length_scales"]], dtype=torch.float64, device=device)
    nz = torch.tensor([w["noise"]], dtype=torch.float64, device=device)
    g = gdist.BlockCyclicGP(X, Y, w["kind"], tile=args.tile)

    def step():
        return g.log_likelihood(var, ls, nz, Y)

    for _ in range(args.warmup):
        step()
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        out = step()
    barrier()
    elapsed = max_over_ranks(time.perf_counter() - t0)
    lml = float(out.item())
    sec = elapsed / args.steps
    agg = (w["n"] ** 3 / 3.0) / sec / 1e12

    # rank 0's contraction time inside one distributed evaluation (compute vs exchange/wait split)
    lib.gpn_profile_enable(1)
    step()
    torch.cuda.synchronize()
    cls = collect_classes(lib)
    lib.gpn_profile_enable(0)
    gemm_ms = sum(cls[c][1] for c in (P_GEMM, P_SYRK, P_SOLVE, P_TRI))
    barrier()

    extra = {}
    if not args.no_extras:
        # the same matrix on ONE GPU (rank 0 alone; the others wait): the N = 1 point of the strong-scaling curve
        single = None
        if rank == 0:
            try:
                m1, _, _ = build_model(w, 0, device)

                def st():
                    with torch.no_grad():
                        return m1.log_likelihood()
                t1, o1 = timed(st, 2, 1)
                single = {"config": w["name"] + " on one GPU (rank 0 alone, same run)", "ms_per_step": t1 * 1e3, "value": 1.0 / t1,
                          "lml": o1.item(), "cholesky_frac_of_fp64_peak": (w["n"] ** 3 / 3.0) / t1 / 1e12 / PEAK_FP64_MFMA_TFLOPS}
                del m1
                torch.cuda.empty_cache()
            except Exception as exc:
                notes["single_gpu_same_run_error"] = repr(exc)
        barrier()
        if single is not None:
            extra["single_gpu_same_run"] = single
            extra["speedup_vs_single_gpu_same_run"] = single["ms_per_step"] * 1e-3 / sec
            extra["lml_abs_diff_vs_single_gpu"] = abs(lml - single["lml"])
        if args.dist_backward:
            # opt-in: one distributed loss + closed-form backward (U = L^-T carried on the grid, Kyy^-1 = U U^T,
            # per-rank sweeps) of the same model -- 2x the local matrix and ~3x the evaluation's time
            try:
                barrier()
                t0 = time.perf_counter()
                lml_b, grad = g.log_likelihood_and_grad(var, ls, nz, Y)
                barrier()
                tb = max_over_ranks(time.perf_counter() - t0)
                extra["dist_loss_backward"] = {"config": w["name"].replace("LML eval", "LML + closed-form gradients") + ", block-cyclic over %d GPUs" % world,
                                               "ms_per_step": tb * 1e3, "lml": float(lml_b.item()), "grads_constrained": [float(v) for v in grad.tolist()]}
            except Exception as exc:
                notes["dist_backward_error"] = repr(exc)
        # labelled extra: independent C2 replicas, one model per GPU, no collective (GP-fits/s at small N)
        try:
            mr, _, _ = build_model(WORKLOADS["c2"], seed=rank, device=device)

            def sr():
                with torch.no_grad():
                    return mr.log_likelihood()
            for _ in range(3):
                sr()
            barrier()
            t0 = time.perf_counter()
            for _ in range(20):
                sr()
            barrier()
            tr = max_over_ranks(time.perf_counter() - t0)
            extra["replicas_c2"] = {"config": "C2 x %d independent replicas (one model per GPU, rank r = seed r, no collective)" % world,
                                    "value": world * 20 / tr, "unit": "LML evals/s", "scaling": "weak"}
        except Exception as exc:
            notes["replicas_error"] = repr(exc)

    if rank == 0:
        peak = world * PEAK_FP64_MFMA_TFLOPS
        line = {
            "metric": "GP log-marginal-likelihood evals/sec (Cholesky+solve) at NxD fp64",
            "value": 1.0 / sec, "unit": "LML evals/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": sec * 1e3,
            "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
            "dtype": "f64", "data": "synthetic",
            "config": {"workload": w["name"] + ", ONE model 2-D block-cyclic over %d GPUs" % world, "N": w["n"], "D": w["d"], "dy": w["dy"],
                       "kernel": w["kind"], "parallelism": "block-cyclic %dx%d grid, tile %d, %s broadcasts on row/column sub-communicators"
                       % (g.pr, g.pc, g.T, dist.get_backend())},
            "backend": dist.get_backend(), "world_size_reported_by_backend": dist.get_world_size(),
            "single_factorisation_wall_s": sec, "lml": lml, "info": g.info,
            "roofline": {"bound": "mfma", "kernel": "whole evaluation, all ranks: N^3/3 flops / wall (the contraction kernel carries all but the leaves)",
                         "achieved": agg, "peak": peak, "unit": "TFLOP/s", "frac": agg / peak, "traffic": None,
                         "peak_note": "%d x %.1f TFLOP/s fp64 MFMA" % (world, PEAK_FP64_MFMA_TFLOPS)},
            "rank0_contraction_ms_per_step": gemm_ms, "rank0_local_matrix_gb": g.A.numel() * 8 / 1e9,
        }
        line.update(extra)
        if notes:
            line["notes"] = notes
        print(json.dumps(line), flush=True)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=None)
    ap.add_argument("--warmup", type=int, default=None)
    ap.add_argument("--workload", default=None, choices=sorted(WORKLOADS),
                    help="default: c3 on one GPU; c4 (block-cyclic) on several")
    ap.add_argument("--tile", type=int, default=2048, help="block-cyclic tile size (N > 1 GPUs)")
    ap.add_argument("--dist-backward", action="store_true", help="(N > 1 GPUs) also time one distributed loss + backward")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extras", action="store_true", help="headline + rooflines only (profiling runs)")
    ap.add_argument("--test-shared-gpu", action="store_true",
                    help="(testing the multi-rank control flow on a 1-GPU box) every rank uses cuda:0, gloo collectives")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if args.gpus != world:
        sys.exit("bench.py: --gpus %d but WORLD_SIZE=%d: launch N>1 with `python -m torch.distributed.run --nproc-per-node N "
                 "--master-addr 127.0.0.1 bench.py --gpus N`" % (args.gpus, world))
    args.workload_given = args.workload is not None
    if args.workload is None:
        args.workload = "c3" if world == 1 else "c4"
    big = WORKLOADS[args.workload]["n"] > 8192
    if args.steps is None:
        args.steps = (20 if world == 1 else 5) if big else 20
    if args.warmup is None:
        args.warmup = (3 if world == 1 else 2) if big else 3
    if args.test_shared_gpu:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    if world == 1:
        run_single(args, device)
        return
    import torch.distributed as dist
    if args.test_shared_gpu:
        dist.init_process_group("gloo", timeout=datetime.timedelta(minutes=10))
    else:
        dist.init_process_group("nccl", device_id=device, timeout=datetime.timedelta(minutes=10))
    try:
        run_multi(args, rank, local_rank, world, device)
    except Exception as exc:
        # never leave the driver without a line: an unmeasured run says so (value null) with the reason
        if rank == 0:
            print(json.dumps({"metric": "GP log-marginal-likelihood evals/sec (Cholesky+solve) at NxD fp64", "value": None,
                              "unit": "LML evals/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": None,
                              "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
                              "config": {"workload": WORKLOADS[args.workload]["name"] + ", ONE model 2-D block-cyclic over %d GPUs" % world},
                              "error": repr(exc)}), flush=True)
        raise
    finally:
        try:
            dist.destroy_process_group()
        except Exception:
            pass


if __name__ == "__main__":
    main()

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
    single-GPU configs, each with its own config string: `c1` (N=512, with its CPU timing), `c2` (N=8192, D=8, Rbf; + `cpu_baseline`), `c4_1gpu`
    (N=65536, D=32: the 34 GB factor fits one GPU's HBM), `c5_vfe` (sparse VFE, N=1e6, M=4096) and
    `loss_backward` (one Adam step's loss()+backward()) at C2 and C3.
--gpus N>1 (launched by torch.distributed.run, one rank per GPU): STRONG scaling of
    configs[3] = "C4" (GPR + Rbf, N=65536, D=32): ONE model, its Gram matrix 2-D block-cyclic over
    all ranks (gptorch_amd/dist.py), panels exchanged by RCCL broadcasts on row / column
    sub-communicators.  value = evaluations/s of that single sharded model.  Rank 0 then times the
    same matrix on its own GPU alone (`single_gpu_same_run`) and every rank runs independent C2
    replicas (`replicas_c2`, labelled, not the headline).

Extra objects on the JSON line (N=1):
  roofline            -- the dominant kernel on one stream: the fp64 MFMA contraction kernel's SYRK trailing updates (the
                         lower-tile K = panel-width launch at the end of every panel; north_star's ">= 50 % of fp64
                         MFMA peak at N=32768"): algorithmic flops of those launches / their summed HIP-event time
                         (events on the launch stream, second pass over the same steps).  = roofline_syrk
  roofline_all_contractions -- ALL launches of gemm_nt_kernel (two overlapping streams: a lower bound, see its note)
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
               d=32, dy=1, variance=1.0, length_scales=float(np.sqrt(32.0)), noise=1e-2, golden=("lml_c4_cpu_oracle.json", None)),
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


def golden_source(w):
    """who computed the golden LML of this workload: "reference" (the imported gptorch, tests/golden/make_golden.py) or
    "cpu_oracle" (oracle/ at full size on a GPU box's host, where the reference does not fit: C4)."""
    g = w.get("golden")
    return None if not g else ("cpu_oracle" if "cpu_oracle" in g[0] else "reference")


def golden_err_key(w):
    return "lml_abs_err_vs_%s_golden" % golden_source(w)


def golden_lml(w):
    """the reference's LML for this workload from the committed fixtures (tests/golden/, generated
    by importing the reference: tests/golden/make_golden.py), or None.  C4 (N = 65536) does not fit the container the
    reference runs in: its fixture is the CPU oracle's value measured at full size on a GPU box's host
    (tests/sweeps/c4_cpu_parity.py; provenance in the file)."""
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


CPU_CHILD = r"""
import json, os, sys, time
sys.path.insert(0, %(root)r)
import numpy as np
import torch
spec = json.loads(sys.argv[1])
if spec.get("threads"):
    torch.set_num_threads(int(spec["threads"]))
from gptorch_amd import rng
from oracle import gp_oracle as orc
w = spec["w"]
x, y = rng.make_regression(w["n"], w["d"], w["dy"], seed=0)
ns = int(spec["rows"])
o = orc.GPROracle(x[:ns], y[:ns], kind=w["kind"], variance=w["variance"], length_scales=w["length_scales"], noise=w["noise"])
times, val = [], None
warm, reps, budget = int(spec["warmup"]), int(spec["reps"]), float(spec.get("budget") or 0.0)
t_start = time.time()
with torch.no_grad():
    i = 0
    while i < warm + reps:
        t0 = time.time()
        val = float(o.log_likelihood().item())
        dt = time.time() - t0
        if i >= warm:
            times.append(dt)
        i += 1
        if budget > 0 and i >= warm + 1 and (time.time() - t_start) + dt > budget:
            break          # one more evaluation would overrun the budget: stop with what there is (never fewer than one)
try:
    import resource
    peak_gb = resource.getrusage(resource.RUSAGE_SELF).ru_maxrss / 1e6
except Exception:
    peak_gb = None
mkl = None
try:
    mkl = torch.backends.mkl.is_available()
except Exception:
    pass
print("CPU_CHILD_RESULT " + json.dumps({"times": times, "lml": val, "threads": torch.get_num_threads(), "interop_threads": torch.get_num_interop_threads(),
                                        "mkl_available": mkl, "peak_rss_gb": peak_gb,
                                        "parallel_info": [l.strip() for l in torch.__config__.parallel_info().splitlines() if "threads" in l.lower() or "MKL" in l][:8]}))
"""


def cpu_child(w, rows, threads, warmup, reps, timeout, budget=0.0):
    """the oracle in a CHILD process (CPU only; it never touches the GPU): a host OOM-kill or a time-out there costs this
    leg only, never the line.  -> dict or raises."""
    import subprocess
    spec = {"w": {k: w[k] for k in ("n", "d", "dy", "kind", "variance", "length_scales", "noise")}, "rows": rows, "threads": threads,
            "warmup": warmup, "reps": reps, "budget": budget}
    env = dict(os.environ, HIP_VISIBLE_DEVICES="", ROCR_VISIBLE_DEVICES="")
    out = subprocess.run([sys.executable, "-c", CPU_CHILD % {"root": ROOT}, json.dumps(spec)], capture_output=True, text=True, timeout=timeout, env=env)
    for ln in out.stdout.splitlines():
        if ln.startswith("CPU_CHILD_RESULT "):
            return json.loads(ln[len("CPU_CHILD_RESULT "):])
    raise RuntimeError("cpu oracle child exited with code %d: %s" % (out.returncode, out.stderr[-400:]))


def host_mem_available_gb():
    """what this process may still allocate: the host's MemAvailable, capped by the container's cgroup limit (the GPU
    boxes show 3 TB of host memory behind a 300 GiB memory.max: exceeding THAT kills the box's job)."""
    avail = None
    try:
        for ln in open("/proc/meminfo"):
            if ln.startswith("MemAvailable:"):
                avail = int(ln.split()[1]) / 1e6
    except Exception:
        pass
    for mx, cur in (("/sys/fs/cgroup/memory.max", "/sys/fs/cgroup/memory.current"),
                    ("/sys/fs/cgroup/memory/memory.limit_in_bytes", "/sys/fs/cgroup/memory/memory.usage_in_bytes")):
        try:
            limit = open(mx).read().strip()
            if limit == "max":
                continue
            room = (int(limit) - int(open(cur).read().strip())) / 1e9
            if room < 1e6:                       # (an unlimited v1 cgroup reports ~9e18)
                avail = room if avail is None else min(avail, room)
        except Exception:
            pass
    return avail


def cgroup_cpu_quota():
    """the container's CPU quota in cores (cgroup v2 cpu.max / v1 cfs quota), or None when unlimited / unreadable: a box that
    shows 256 host CPUs behind an 8-core quota runs MKL's 128 threads on 8 cores."""
    try:
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        return None if q == "max" else float(q) / float(per)
    except Exception:
        pass
    try:
        q = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
        per = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
        return None if q <= 0 else q / float(per)
    except Exception:
        return None


def usable_cpus():
    try:
        return len(os.sched_getaffinity(0))
    except Exception:
        return os.cpu_count() or 1


def cpu_baseline(w, x, y, full=True, full_timeout=900.0, budget=90.0):
    """the CPU oracle (oracle/gp_oracle.py: the reference's op sequence on the same ATen / MKL kernels, kind "port") on
    this box's host cores, SURVEY 8(d).
      1. an 8192-row sample of the SAME data / kernel / hyper-parameters at several thread counts (a few seconds each):
         MKL's dpotrf does not scale to 128+ threads at this size, so the thread count for step 2 is the fastest one
         measured, not the box's core count;
      2. the whole workload (C3: N = 32768, about 35-50 s per evaluation, about 4 live N x N = 35 GB of host memory) in ONE child
         process: one warm-up evaluation + up to 3 timed ones, `value` = 1 / MEDIAN ("median of >= 3 after one
         warm-up"), MEASURED (`extrapolated: false`).  The number of timed evaluations is cut (never below 1, and said so in
         `sample`) when the warm-up shows that 3 more would overrun `budget` seconds (--cpu-baseline-budget), so that the
         default run stays inside the driver's time-out.
    The thread count of step 2 is fixed by step 1 over candidates capped by the box's USABLE cores -- the cgroup CPU quota
    (cpu.max) and the affinity mask, both recorded beside `cores`: round 4's two boxes picked 16 threads (37 s) and 8 (50 s)
    from single un-warmed samples.  If the full-size run cannot be made (memory, time-out), the extrapolation is reported
    and flagged as before."""
    ns = min(w["n"], 8192)
    ncpu = os.cpu_count() or 1
    quota, aff = cgroup_cpu_quota(), usable_cpus()
    cap = int(min(ncpu, aff, max(1.0, np.ceil(quota)) if quota else ncpu))
    sweep = {}
    for th in sorted({t for t in (8, 16, 32, 64, 128, cap) if t <= cap} or {cap}):
        try:
            r = cpu_child(w, ns, th, 1, 3, 300.0)
            sweep[th] = float(np.median(r["times"]))
            info = r
        except Exception as exc:
            sweep[th] = None
    good = {k: v for k, v in sweep.items() if v}
    if not good:
        raise RuntimeError("no CPU sample could be timed")
    best_th = min(good, key=good.get)
    med = good[best_th]
    scale = (w["n"] / float(ns)) ** 3
    out = {"value": 1.0 / (med * scale), "unit": "LML evals/s", "cores": best_th, "host_cpus": ncpu, "cgroup_cpu_quota_cores": quota,
           "affinity_cpus": aff, "thread_candidates_capped_at": cap, "kind": "port",
           "seconds_per_eval": med * scale, "torch_default_threads": torch.get_num_threads(),
           "sample_seconds_by_threads": {str(k): v for k, v in sorted(sweep.items())},
           "parallel_info": info.get("parallel_info"), "mkl_available": info.get("mkl_available"),
           "sample": "N=%d rows of the same workload (D=%d, %s): 3 evaluations after 1 warm-up per thread count, median; fastest: %d threads, %.3f s"
                     % (ns, w["d"], w["kind"], best_th, med)}
    if ns == w["n"]:
        out["extrapolated"] = False
        return out
    out["extrapolated"] = True
    out["measured_sample_evals_per_s"] = 1.0 / med
    out["extrapolated_seconds_per_eval_N3"] = med * scale
    if full:
        avail = host_mem_available_gb()
        need = 4.5 * 8.0 * w["n"] ** 2 / 1e9
        if avail is not None and avail < need:
            out["full_size_skipped"] = "host MemAvailable %.0f GB < %.0f GB needed for ~4.5 live N x N fp64" % (avail, need)
        else:
            try:
                # 1 warm-up + 3 timed evaluations; the child stops earlier (never before one timed evaluation) when the next
                # one would overrun the budget -- decided from the evaluations it has timed itself, not from the sample's
                # (N/8192)^3 extrapolation, which over-estimates 3x (MKL runs the big factorisation far more efficiently)
                r = cpu_child(w, w["n"], best_th, 1, 3, full_timeout, budget=budget)
                ts = [float(v) for v in r["times"]]
                reps = len(ts)
                t = float(np.median(ts))
                out.update({"value": 1.0 / t, "seconds_per_eval": t, "extrapolated": False, "lml": r["lml"], "peak_rss_gb": r.get("peak_rss_gb"),
                            "full_size_seconds": ts, "full_size_warmups": 1, "full_size_reps": reps,
                            "fitted_exponent_8192_to_N": float(np.log(t / med) / np.log(w["n"] / float(ns))),
                            "sample": "the WHOLE workload (N=%d, D=%d, %s) on %d threads: median of %d evaluation(s) after 1 warm-up = %.1f s "
                                      "(measured, not extrapolated%s); beside it the %d-row sample (%.3f s on its fastest thread count, %d) whose "
                                      "(N/%d)^3 extrapolation would have said %.1f s"
                                      % (w["n"], w["d"], w["kind"], best_th, reps, t,
                                         "" if reps >= 3 else "; fewer than 3 to keep the run inside --cpu-baseline-budget %.0f s" % budget,
                                         ns, med, best_th, ns, med * scale)})
            except Exception as exc:
                out["full_size_error"] = repr(exc)[:300]
    if out["extrapolated"]:
        out["sample"] += "; value EXTRAPOLATED to N=%d by (N/%d)^3 = %.0fx (Cholesky-bound, N^3/3 flops)" % (w["n"], ns, scale)
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
            "launches_per_step": cls[P_TRI][0] / steps,
            # tiles launched x 2 BM BN K with the UNCLIPPED K: these launches skip the zero half of their triangular operands, so
            # this is an upper bound on what ran (it can exceed the peak), not an executed rate -- `achieved` is the figure
            "launched_tile_tflops_unclipped_k": cls[P_TRI][2] / steps / (tri_ms * 1e-3) / 1e12}
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
    if "roofline_syrk" in roofs:
        # `roofline` = the DOMINANT kernel on ONE stream: the lower-tile launches of the contraction kernel (the SYRK trailing
        # updates: ~78 % of an evaluation's wall time at C3, all on the caller's stream, so launches x average duration is a
        # real time).  The figure over ALL contraction launches sums launches that overlap on the aux stream -- its
        # kernel_ms_per_step can exceed ms_per_step -- and is kept as a conservative lower bound under its own key.
        roofs["roofline_all_contractions"] = roofs["roofline"]
        roofs["roofline_all_contractions"]["note"] = ("sum over launches of two overlapping streams: kernel_ms_per_step is not a "
                                                      "wall time and frac is a lower bound")
        roofs["roofline"] = dict(roofs["roofline_syrk"])

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
                        res[golden_err_key(ww)] = abs(res["lml"] - gl)
                        res["lml_golden_provenance"] = golden_source(ww)
                    r2 = rooflines(lib, ww, steps, st)
                    if "roofline_syrk" in r2:
                        r2["roofline_all_contractions"], r2["roofline"] = r2["roofline"], dict(r2["roofline_syrk"])
                    attach_traffic(r2.get("roofline_syrk"), "syrk_bytes_per_launch", key)
                    attach_traffic(r2.get("roofline"), "syrk_bytes_per_launch", key)
                    attach_traffic(r2.get("roofline_all_contractions"), "gemm_bytes_per_launch", key)
                    attach_traffic(r2.get("roofline_k_assembly"), "kmat_bytes_per_launch", key)
                    for k2 in ("roofline", "roofline_syrk", "roofline_all_contractions", "roofline_k_assembly"):
                        if k2 in r2:
                            res[k2] = r2[k2]
                if with_backward:
                    res["loss_backward"] = backward_leg(lib, m, ww, max(2, steps // 4) if ww["n"] <= 8192 else 2)
                if key in ("c2", "c3"):
                    # configs[1] reads "kernel build + Cholesky + predict": GPR._predict at 1024 test points (gpr.py:88-117) with
                    # the factor cached between calls (steady state: the right-solve through the inverted 1024 x 1024 diagonal
                    # blocks), and re-factorising every call as the reference does (gpr.py:104)
                    from gptorch_amd import rng
                    ns_ = 1024
                    xs = torch.tensor(rng.normal(7, (ns_, ww["d"])), device=device)

                    def pred():
                        with torch.no_grad():
                            return m._predict(xs)

                    def pred_refactor():
                        m._predict_cache = None
                        return pred()
                    tp, _ = timed(pred, 10, 4)
                    nn = float(ww["n"])
                    pflops = nn * nn * ns_                       # the TRSM of predict, N^2 N*
                    res["predict"] = {"config": "%s: GPR._predict at %d test points, diag variance" % (ww["name"][:2], ns_), "ms_cached_factor": tp * 1e3}
                    res["roofline_predict"] = {
                        "bound": "mfma", "kernel": "gemm_nt_kernel launches of gpn_predict_blocked (X_k = B_k W_k^T, B_rest -= X_k L^T; "
                                                   "N / 1024 steps), whole call timed with the factor and its block inverses cached",
                        "achieved": pflops / tp / 1e12, "peak": PEAK_FP64_MFMA_TFLOPS, "unit": "TFLOP/s",
                        "frac": pflops / tp / 1e12 / PEAK_FP64_MFMA_TFLOPS, "traffic": None, "algorithmic_flops_per_call": pflops,
                        "hbm_bytes_K_star_write": 8.0 * nn * ns_, "ms_per_call": tp * 1e3}
                    if key == "c2":
                        tr, _ = timed(pred_refactor, 5, 1)
                        res["predict"]["ms_with_refactorisation"] = tr * 1e3
                return res
            return run

        if args.workload == "c3":
            leg("c2", lml_leg("c2", 20, 3, True))
            leg("c3_backward", lml_leg("c3", 0, 0, True))
            if "c3_backward" in extra:
                extra["loss_backward"] = extra.pop("c3_backward")["loss_backward"]

            def adam50():
                # BASELINE configs[2] end to end: "50 Adam steps of hyperparameter optimisation" = ONE GP fit, through
                # GPModel.optimize (base.py:260-269: loss, backward, step and the loss.item() read-back every step)
                mm, _, _ = build_model(WORKLOADS["c3"], 0, device)
                mm.loss().backward()
                mm.zero_grad()                           # warm-up: allocations, side streams
                torch.cuda.synchronize()
                import contextlib
                t0 = time.perf_counter()
                with contextlib.redirect_stdout(sys.stderr):      # optimize() reports like the reference (base.py:255-296): not on the JSON's stream
                    losses, _ = mm.optimize(method="Adam", max_iter=50, verbose=False, learning_rate=0.01)
                torch.cuda.synchronize()
                dt = time.perf_counter() - t0
                return {"config": "C3: GPR+Matern52 N=32768 D=16 fp64, one fit = 50 Adam steps (lr 0.01) of GPModel.optimize",
                        "s_per_fit": dt, "fits_per_s": 1.0 / dt, "ms_per_adam_step": dt / 50 * 1e3, "loss_first": float(losses[0]),
                        "loss_last": float(losses[-1]), "frac_of_fp64_peak_on_N3_per_step": 50 * 32768.0 ** 3 / dt / 1e12 / PEAK_FP64_MFMA_TFLOPS}
            if not args.no_fit:
                held.clear()
                torch.cuda.empty_cache()
                leg("c3_adam50", adam50)
                if "c3_adam50" in extra:
                    extra["fits_per_s"] = extra["c3_adam50"]["fits_per_s"]
                    extra["s_per_fit"] = extra["c3_adam50"]["s_per_fit"]

            def restarts():
                from gptorch_amd.models import batched_log_likelihood
                from gptorch_amd.models.gpr import two_lane_streams
                R = 4
                models = [build_model(WORKLOADS["c2"], seed=100 + r, device=device)[0] for r in range(R)]
                res = {}
                for label, streams in (("two_lanes", two_lane_streams(models)), ("back_to_back", [torch.cuda.current_stream(device)] * R)):
                    t, _ = timed(lambda: batched_log_likelihood(models, streams), 5, 2)
                    res[label] = R / t
                return {"config": "C2 x %d independent restarts as WHOLE evaluations alternating between two HIP streams (round 3's placement; "
                                  "c2_batched is the lock-step form)" % R,
                        "evals_per_s": res["two_lanes"], "evals_per_s_back_to_back": res["back_to_back"]}

            def lockstep(key, B):
                # hyper-parameter restarts in LOCK STEP (gpn_lml_forward_batched: leaf grid = B, strided-batch column passes and
                # contractions): B models of one shape over shared data, each value bit-identical to its own log_likelihood()
                def run():
                    from gptorch_amd.models import batched_log_likelihood
                    ww = WORKLOADS[key]
                    models = []
                    for b in range(B):
                        mb = build_model(dict(ww, variance=ww["variance"] * (1.0 + 0.01 * b), length_scales=ww["length_scales"] * (1.0 + 0.02 * b)),
                                         0, device)[0]
                        if models:
                            mb.X, mb.Y = models[0].X, models[0].Y
                        models.append(mb)
                    with torch.no_grad():
                        t, out = timed(lambda: batched_log_likelihood(models), 5, 2)
                        t1, _ = timed(lambda: [m.log_likelihood() for m in models], 2, 1)
                        same = [o.item() for o in out] == [m.log_likelihood().item() for m in models]
                    n = ww["n"]
                    return {"config": "%s x %d restarts in lock step (batched_log_likelihood -> gpn_lml_forward_batched)" % (ww["name"], B),
                            "batch": B, "evals_per_s": B / t, "ms_per_batch": t * 1e3, "evals_per_s_one_after_the_other": B / t1,
                            "bit_identical_to_sequential": bool(same),
                            "cholesky_frac_of_fp64_peak": B * (n ** 3 / 3.0) / t / 1e12 / PEAK_FP64_MFMA_TFLOPS}
                return run
            def fit_lockstep(key, B, iters, seq_models, seq_iters, capture=False):
                # north_star's GP-fits/sec at small N: B restarts x `iters` Adam steps as ONE lock-step fit (multi_start_optimize:
                # gpn_lml_forward_batched + gpn_lml_backward_batched + one optimiser step on the stacked raw parameters per
                # iteration) against the reference's loop, one model and one step at a time (base.py:260-269: GPModel.optimize)
                def run():
                    import contextlib
                    from gptorch_amd.models import multi_start_optimize
                    ww = WORKLOADS[key]

                    def restarts_(count):
                        ms_ = []
                        for b in range(count):
                            mb = build_model(dict(ww, variance=ww["variance"] * (1.0 + 0.01 * b), length_scales=ww["length_scales"] * (1.0 + 0.02 * b)),
                                             0, device)[0]
                            if ms_:
                                mb.X, mb.Y = ms_[0].X, ms_[0].Y
                            ms_.append(mb)
                        return ms_
                    models = restarts_(B)
                    with contextlib.redirect_stdout(sys.stderr):
                        multi_start_optimize(models, method="Adam", max_iter=2, learning_rate=0.01)      # warm-up: buffers, first launches
                        torch.cuda.synchronize()
                        t0 = time.perf_counter()
                        losses, _ = multi_start_optimize(models, method="Adam", max_iter=iters, learning_rate=0.01, capture=capture)
                        torch.cuda.synchronize()
                        dt = time.perf_counter() - t0
                        del models
                        torch.cuda.empty_cache()
                        seq = restarts_(seq_models)
                        seq[0].optimize(method="Adam", max_iter=2, verbose=False, learning_rate=0.01)
                        torch.cuda.synchronize()
                        t0 = time.perf_counter()
                        for m_ in seq:
                            m_.optimize(method="Adam", max_iter=seq_iters, verbose=False, learning_rate=0.01)
                        torch.cuda.synchronize()
                        dseq = (time.perf_counter() - t0) / (seq_models * seq_iters)      # seconds per model and step
                    n = float(ww["n"])
                    return {"config": "%s -> %d restarts x %d Adam steps (lr 0.01) as ONE lock-step fit (multi_start_optimize -> "
                                      "gpn_lml_forward_batched + gpn_lml_backward_batched)%s" % (ww["name"].replace(" LML eval", ""), B, iters,
                                                                                                 ", the iteration as one hipGraph replay (capture=True)" if capture else ""),
                            "batch": B, "adam_steps": iters, "s_per_batched_fit": dt, "fits_per_s": B / dt, "ms_per_batched_step": dt / iters * 1e3,
                            "frac_of_fp64_peak_on_N3": B * iters * n ** 3 / dt / 1e12 / PEAK_FP64_MFMA_TFLOPS,
                            "sequential_ms_per_model_step": dseq * 1e3, "sequential_fits_per_s": 1.0 / (iters * dseq),
                            "sequential_sample": "%d model(s) x %d steps of GPModel.optimize, one after the other" % (seq_models, seq_iters),
                            "speedup_vs_sequential": (B / dt) * (iters * dseq),
                            "loss_first_restart0": float(losses[0, 0]), "loss_last_restart0": float(losses[0, -1]),
                            "gradients": "bit-identical per model to loss().backward() (tests/test_gpu_lockstep_fit.py)"}
                return run
            def fit_captured(key, iters):
                # round 6: ONE model, the reference's loop (base.py:260-269) with the optimiser step as one hipGraph replay
                # (GPModel.optimize(capture=True)) against the ordinary loop (one host read-back of the loss per iteration)
                def run():
                    import contextlib
                    ww = WORKLOADS[key]
                    res = {}
                    for cap in (False, True):
                        m_ = build_model(ww, 0, device)[0]
                        with contextlib.redirect_stdout(sys.stderr):
                            m_.optimize(method="Adam", max_iter=8, verbose=False, learning_rate=0.01, capture=cap)
                            torch.cuda.synchronize()
                            t0 = time.perf_counter()
                            losses, _ = m_.optimize(method="Adam", max_iter=iters, verbose=False, learning_rate=0.01, capture=cap)
                            torch.cuda.synchronize()
                            res[cap] = ((time.perf_counter() - t0) / iters * 1e3, float(losses[-1]))
                    return {"config": "%s -> one model, %d Adam steps (lr 0.01): the step as one hipGraph replay" % (ww["name"].replace(" LML eval", ""), iters),
                            "ms_per_step_captured": res[True][0], "ms_per_step_ordinary_loop": res[False][0], "speedup": res[False][0] / res[True][0],
                            "final_loss_captured": res[True][1], "final_loss_ordinary_loop": res[False][1]}
                return run

            def persistent(key):
                # round 6: gpn_potrf_lower as ONE persistent dataflow launch (csrc/ppotrf.hip) against the launch-based driver on the
                # same matrix: built as asked, bit-identical, SLOWER -- it is not what log_likelihood() runs
                def run():
                    from gptorch_amd import _ops
                    ww = WORKLOADS[key]
                    mdl = build_model(ww, 0, device)[0]
                    k = mdl.kernel
                    f = _ops.Factor(ww["n"], ww["dy"], device)
                    _ops.kernel_matrix(k._kind, mdl.X, None, k.variance.transform(), k.length_scales.transform(),
                                       noise=mdl.likelihood.variance.transform(), out=f.A, ldk=f.ld, lower=True)
                    f.pack_rhs(mdl.Y)
                    saved = f.A.clone()
                    st = _ops._stream(device)

                    def one(fn):
                        ts, snap = [], None
                        for rep in range(6):
                            f.A.copy_(saved)
                            f.info.zero_()
                            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                            e0.record()
                            rc = fn(st, _ops._ptr(f.A), f.n, f.e, f.ld, _ops._ptr(f.winv), _ops._ptr(f.info))
                            e1.record()
                            torch.cuda.synchronize()
                            if rc != 0:
                                raise RuntimeError("status %d" % rc)
                            if rep >= 2:
                                ts.append(e0.elapsed_time(e1))
                        snap = (torch.tril(f.A[:f.n, :f.n]).clone(), f.A[f.n:f.n + f.e, :f.n].clone(), f.winv.clone(), int(f.info.item()))
                        return float(np.median(ts)), snap
                    t_l, a = one(lib.gpn_potrf_lower)
                    t_p, b = one(lib.gpn_potrf_lower_persistent)
                    same = all(torch.equal(u, v) for u, v in zip(a[:3], b[:3])) and a[3] == b[3]
                    return {"config": "%s: the factorisation alone, launch-based driver vs ONE persistent dataflow launch" % ww["name"].replace(" LML eval", ""),
                            "ms_launch_based": t_l, "ms_persistent": t_p, "bit_identical": bool(same), "shipped_driver": "launch-based"}
                return run
            def vfe_lockstep(n, m, d, B):
                # round 6: B sparse (VFE) restarts of one shape, `loss(); backward()` in LOCK STEP (batched_loss_and_grad ->
                # models/_vfe_lockstep.py) against the reference's order, one model after the other (base.py:260-269 over
                # sparse_gpr.py:108-153); every bound and gradient bit-identical to the model's own
                def run():
                    from gptorch_amd import kernels, likelihoods, mean_functions, rng
                    from gptorch_amd.models import VFE, batched_loss_and_grad
                    g = np.random.default_rng(0)
                    xv, yv = rng.make_regression(n, d, 1, seed=0)
                    ms_ = []
                    for b in range(B):
                        v = VFE(xv, yv, kernels.Rbf(d, variance=1.0 + 0.01 * b, length_scales=0.5 * float(np.sqrt(d)) * (1.0 + 0.02 * b)),
                                inducing_points=xv[g.choice(n, m, replace=False)], likelihood=likelihoods.Gaussian(variance=0.05),
                                mean_function=mean_functions.Zero(1))
                        v.cuda()
                        if ms_:
                            v.X, v.Y = ms_[0].X, ms_[0].Y
                        ms_.append(v)

                    def seq():
                        out = []
                        for v in ms_:
                            v.zero_grad()
                            l_ = v.loss()
                            l_.backward()
                            out.append(l_.detach())
                        return out

                    def lock():
                        for v in ms_:
                            v.zero_grad()
                        return batched_loss_and_grad(ms_)
                    t_s, a = timed(seq, 5, 2)
                    ga = [[p_.grad.clone() for p_ in v.parameters() if p_.grad is not None] for v in ms_]
                    t_l, b_ = timed(lock, 5, 2)
                    same = all(torch.equal(u, w) for u, w in zip(a, b_)) and \
                        all(torch.equal(u, w) for gs, v in zip(ga, ms_) for u, w in zip(gs, [p_.grad for p_ in v.parameters() if p_.grad is not None]))
                    return {"config": "VFE + Rbf, N=%d M=%d D=%d: %d restarts, loss(); backward() in lock step" % (n, m, d, B),
                            "batch": B, "ms_lock_step": t_l * 1e3, "ms_one_after_the_other": t_s * 1e3, "speedup": t_s / t_l,
                            "bit_identical_to_sequential": bool(same)}
                return run
            def kfold_lockstep(key, folds, ragged=False):
                # k-fold cross-validation of one model (equal folds: every fold's training set has N (k-1)/k rows of its OWN): the k
                # training evaluations `loss(); backward()` in lock step against one after the other (base.py:260-269 per fold)
                def run():
                    from gptorch_amd import kernels, likelihoods
                    from gptorch_amd.models import GPR, batched_loss_and_grad
                    ww = WORKLOADS[key]
                    m0 = build_model(ww, 0, device)[0]
                    n = ww["n"]
                    per = n // folds
                    ms_ = []
                    for f_ in range(folds):
                        keep = torch.cat([torch.arange(0, f_ * per, device=device), torch.arange((f_ + 1) * per, per * folds, device=device)])
                        if ragged:                 # folds of unequal length: fold f_ trains on 3 f_ rows fewer (one ragged lock-step group)
                            keep = keep[:keep.numel() - 3 * f_]
                        k_ = type(m0.kernel)(ww["d"], variance=ww["variance"], length_scales=ww["length_scales"])
                        mf = GPR(m0.X[keep].contiguous(), m0.Y[keep].contiguous(), k_, likelihood=likelihoods.Gaussian(variance=ww["noise"]))
                        mf.cuda()
                        ms_.append(mf)

                    def seq():
                        out = []
                        for v in ms_:
                            v.zero_grad()
                            l_ = v.loss()
                            l_.backward()
                            out.append(l_.detach())
                        return out

                    def lock():
                        for v in ms_:
                            v.zero_grad()
                        return batched_loss_and_grad(ms_)
                    t_s, a = timed(seq, 3, 1)
                    ga = [[p_.grad.clone() for p_ in v.parameters() if p_.grad is not None] for v in ms_]
                    t_l, b_ = timed(lock, 3, 1)
                    same = all(torch.equal(u.reshape(-1), w.reshape(-1)) for u, w in zip(a, b_)) and \
                        all(torch.equal(u, w) for gs, v in zip(ga, ms_) for u, w in zip(gs, [p_.grad for p_ in v.parameters() if p_.grad is not None]))
                    nf = float(per * (folds - 1))
                    return {"config": "%s: %d-fold cross-validation, the %d training evaluations (N = %s rows, own data) loss(); backward() in lock step%s"
                                      % (ww["name"].replace(" LML eval", ""), folds, folds,
                                         "%d ... %d" % (int(nf) - 3 * (folds - 1), int(nf)) if ragged else "%d each" % int(nf),
                                         " as ONE ragged group (padded to the largest fold with identity rows)" if ragged else ""),
                            "folds": folds, "ms_lock_step": t_l * 1e3, "ms_one_after_the_other": t_s * 1e3, "speedup": t_s / t_l,
                            "frac_of_fp64_peak_on_N3": folds * nf ** 3 / t_l / 1e12 / PEAK_FP64_MFMA_TFLOPS, "bit_identical_to_sequential": bool(same)}
                return run
            leg("c2_kfold_batched", kfold_lockstep("c2", 8))
            leg("c2_kfold_ragged", kfold_lockstep("c2", 8, ragged=True))
            leg("vfe_lockstep_n512", vfe_lockstep(512, 64, 2, 64))
            leg("vfe_lockstep_n8192", vfe_lockstep(8192, 512, 8, 8))
            leg("c2_batched", lockstep("c2", 8))
            leg("c1_batched", lockstep("c1", 64))
            leg("c2_persistent_factorisation", persistent("c2"))
            if not args.no_fit:
                leg("c1_fit_captured", fit_captured("c1", 300))
                leg("c2_fit_batched", fit_lockstep("c2", 8, 50, 1, 10))
                leg("c1_fit_batched", fit_lockstep("c1", 64, 50, 4, 50))
                leg("c1_fit_batched_captured", fit_lockstep("c1", 64, 200, 2, 20, capture=True))
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
                res = {"config": "C5: sparse VFE GP, Rbf, N=1e6, M=4096 inducing, D=8 fp64: collapsed-bound evaluation (sparse_gpr.py:108-151)",
                       "value": 1.0 / t, "unit": "bound evals/s", "ms_per_step": t * 1e3, "elbo": o.item(),
                       "tflops_on_2NM2_plus_2M3_3": fl / t / 1e12, "frac_of_fp64_peak": fl / t / 1e12 / PEAK_FP64_MFMA_TFLOPS}
                try:     # the CPU oracle's value at full size (evaluated once on a GPU box's host: the reference does not fit the build container)
                    gold = json.load(open(os.path.join(ROOT, "tests", "golden", "vfe_c5_cpu_oracle.json")))["elbo"]
                    res["elbo_abs_err_vs_cpu_oracle_golden"] = abs(res["elbo"] - gold)
                    res["elbo_rel_err_vs_cpu_oracle_golden"] = abs(res["elbo"] - gold) / abs(gold)
                    res["elbo_golden_provenance"] = "cpu_oracle"
                    res["elbo_golden_note"] = ("K(Z) is singular up to the ladder's jitter at this shape: the oracle itself moves by 2e-11 relative with its thread "
                                               "count; the quarter-size case pinned in extended precision is tests/golden/vfe_extended_262144_2048.json")
                except Exception:
                    pass
                # predictions (sparse_gpr.py:155-195): the reference re-evaluates the bound inside every _predict; the state is kept here
                xs_ = torch.as_tensor(rng.normal(7, (1024, d))).to(device)
                mod._predict_cache = None
                t_first, _ = timed(lambda: mod.predict_y(xs_), 1, 0)
                t_next, _ = timed(lambda: mod.predict_y(xs_), 3, 0)
                res["predict_1024_ms_first"] = t_first * 1e3
                res["predict_1024_ms_state_kept"] = t_next * 1e3
                return res
            leg("c5_vfe", vfe)
        else:
            leg("loss_backward", lambda: backward_leg(lib, held["model"], w, max(2, args.steps // 4) if w["n"] <= 8192 else 2))

    ms = elapsed / args.steps * 1e3
    line = {
        "metric": "GP log-marginal-likelihood evals/sec (Cholesky+solve) at NxD fp64",
        "value": args.steps / elapsed, "unit": "LML evals/s",
        "n_gpus": 1, "steps": args.steps, "warmup": args.warmup, "ms_per_step": ms,
        "higher_is_better": True, "scaling": None, "vs_baseline": None,
        "scaling_note": "one GPU: no scaling claim.  --gpus N>1 is STRONG scaling of C4 (N=65536) and carries its own "
                        "single-GPU point (single_gpu_same_run / speedup_vs_single_gpu_same_run); c4_1gpu below is that point here",
        "dtype": "f64", "data": "synthetic",
        "config": {"workload": w["name"], "N": w["n"], "D": w["d"], "dy": w["dy"], "kernel": w["kind"], "parallelism": "1 GPU"},
        "lml": lml,
        "cholesky_frac_of_fp64_peak": (w["n"] ** 3 / 3.0) / (elapsed / args.steps) / 1e12 / PEAK_FP64_MFMA_TFLOPS,
    }
    gl = golden_lml(w)
    if gl is not None:
        line["lml_%s_golden" % golden_source(w)] = gl
        line[golden_err_key(w)] = abs(lml - gl)
        line["lml_golden_provenance"] = golden_source(w)
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
            line["cpu_baseline"] = cpu_baseline(w, x, y, full=not args.cpu_sample_only, budget=args.cpu_baseline_budget)
        except Exception as exc:
            notes["cpu_baseline_error"] = repr(exc)
        # the CPU path is timed for C1, C2 and C3-forward.  The smaller two beside their GPU legs: median of 3
        # evaluations after one warm-up on the thread count the sweep above found fastest.
        if args.workload == "c3" and not args.no_extras and "cpu_baseline" in line:
            th = int(line["cpu_baseline"].get("cores") or 8)
            for key in ("c1", "c2"):
                try:
                    ww = WORKLOADS[key]
                    r = cpu_child(ww, ww["n"], th, 1, 3, 300.0)
                    rec = {"seconds_per_eval": float(np.median(r["times"])), "cores": th, "kind": "port", "lml": r["lml"],
                           "sample": "the whole workload, median of 3 after 1 warm-up"}
                    if key == "c2" and "c2" in line:
                        line["c2"]["cpu_baseline"] = rec
                    else:
                        # C1 (BASELINE configs[0], "CPU parity"): the GPU evaluation of the same model beside it
                        mm, _, _ = build_model(ww, 0, device)
                        with torch.no_grad():
                            tg, og = timed(lambda: mm.log_likelihood(), 50, 5)
                        gl1 = golden_lml(dict(ww, golden=("lml_cases.json", "C1_rbf_512_2")))
                        line["c1"] = {"config": ww["name"], "ms_per_step": tg * 1e3, "value": 1.0 / tg, "unit": "LML evals/s", "lml": og.item(),
                                      "lml_abs_err_vs_reference_golden": None if gl1 is None else abs(og.item() - gl1),
                                      "lml_abs_diff_vs_cpu_oracle_this_box": abs(og.item() - r["lml"]), "cpu_baseline": rec}
                        del mm
                except Exception as exc:
                    notes["cpu_%s_error" % key] = repr(exc)
    if notes:
        line["notes"] = notes
    print(json.dumps(line), flush=True)


# ------------------------------------------------------------------------------------------------
# N > 1 GPUs: one model, block-cyclic over all ranks
# ------------------------------------------------------------------------------------------------
def run_multi(args, rank, local_rank, world, device):
    import threading
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

    def ranks_seen(group=None):
        """the number of ranks the BACKEND's collective sees in a group: an all-reduce (sum) of one 1 per rank on the device --
        on the nccl backend that is RCCL itself counting its members (torch does not expose ncclCommCount of its communicators)."""
        t = torch.ones(1, dtype=torch.float64, device=cpu if shared else device)
        dist.all_reduce(t, op=dist.ReduceOp.SUM, group=group)
        return int(round(float(t.item())))

    x, y = rng.make_regression(w["n"], w["d"], w["dy"], seed=0)          # the same model on every rank
    X, Y = torch.tensor(x, device=device), torch.tensor(y, device=device)
    var = torch.tensor([w["variance"]], dtype=torch.float64, device=device)
    ls = torch.tensor([w["length_scales"]], dtype=torch.float64, device=device)
    nz = torch.tensor([w["noise"]], dtype=torch.float64, device=device)

    # Two exchange schedules over the same engine layout (gptorch_amd/dist.py): "bcast" = the backend's broadcast on
    # the row / column sub-communicators (RCCL picks the route), "mesh" = grouped point-to-point sends over the direct
    # xGMI links.  Both are timed in this run with the same steps / warmup; the headline is the better one and both
    # are reported.  Order: first schedule -> the safe extras -> second schedule under a watchdog, so that a second
    # schedule that hangs on a fabric it has never seen costs its own numbers only, never the line.
    schedules = [args.schedule] if args.schedule != "both" else ["bcast", "mesh"]
    per_schedule, engines = {}, {}

    def measure(sched):
        if os.environ.get("GPN_BENCH_TEST_HANG_SCHEDULE") == sched and rank == world - 1:
            time.sleep(3600)                       # test hook: one rank never joins this schedule's collectives
        # (the second engine reuses the first one's row / column sub-communicators: same grid, other schedule)
        g = gdist.BlockCyclicGP(X, Y, w["kind"], tile=args.tile, schedule=sched, share=next(iter(engines.values()), None))
        engines[sched] = g

        def step():
            return g.log_likelihood(var, ls, nz, Y)

        for _ in range(args.warmup):
            step()
        barrier()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            out = step()
        barrier()
        el = max_over_ranks(time.perf_counter() - t0)
        res = {"ms_per_step": el / args.steps * 1e3, "lml": float(out.item()), "info": g.info}
        # one more evaluation with an event pair around every wait on a collective: how long this rank's compute
        # stream stood still for the exchange (exposed communication), what it sent to whom, and its contraction time
        g.reset_comm_stats()
        g.comm_timing = True
        lib.gpn_profile_enable(1)
        step()
        torch.cuda.synchronize()
        cls = collect_classes(lib)
        lib.gpn_profile_enable(0)
        g.comm_timing = False
        st = g.comm_stats()
        mine = torch.tensor([st["exposed_comm_ms"], sum(cls[c][1] for c in (P_GEMM, P_SYRK, P_SOLVE, P_TRI)),
                             float(sum(st["sent_bytes_per_peer"].values())), float(max(list(st["sent_bytes_per_peer"].values()) or [0])),
                             float(st["bcast_root_bytes"]), float(st["recv_bytes"]),
                             float(getattr(g, "last_refine_ms", 0.0)) if g.refined else 0.0], dtype=torch.float64,
                            device=cpu if shared else device)
        allr = [torch.zeros_like(mine) for _ in range(world)]
        dist.all_gather(allr, mine)
        allr = torch.stack(allr).cpu()
        # the modelled latency-bound tail: in the last third of the panels the updates are too short to cover the
        # three dependent exchanges per panel -- per-panel exposed waits of every rank there, as a histogram
        pp = torch.tensor(st["exposed_ms_per_panel"], dtype=torch.float64, device=cpu if shared else device)
        allp = [torch.zeros_like(pp) for _ in range(world)]
        dist.all_gather(allp, pp)
        allp = torch.stack(allp).cpu()
        third = allp[:, (2 * allp.shape[1]) // 3:]
        edges = [0.0, 0.05, 0.2, 1.0, 5.0, 1e30]
        hist = [int(((third >= lo) & (third < hi)).sum()) for lo, hi in zip(edges[:-1], edges[1:])]
        res.update({"exposed_comm_ms_per_rank": [float(v) for v in allr[:, 0]], "exposed_comm_ms_max": float(allr[:, 0].max()),
                    "contraction_ms_per_rank": [float(v) for v in allr[:, 1]],
                    "recv_gb_per_rank": [float(v) / 1e9 for v in allr[:, 5]],
                    # the refinement step of the quadratic form on the grid (part of every timed evaluation from 12288
                    # rows on): this rank's stream time in it, incl. its 2 small collectives per tile row
                    "refine_ms_per_rank": [float(v) for v in allr[:, 6]],
                    "last_third_wait_histogram_ms": {"panels": [int((2 * allp.shape[1]) // 3), int(allp.shape[1]) - 1],
                                                     "bucket_edges_ms": edges[:-1] + ["inf"], "counts_over_ranks_x_panels": hist,
                                                     "sum_ms_per_rank": [float(v) for v in third.sum(dim=1)],
                                                     "rank0_ms_per_panel": [round(float(v), 4) for v in third[0]]}})
        if sched == "mesh":
            res["p2p_sent_gb_per_rank"] = [float(v) / 1e9 for v in allr[:, 2]]
            res["p2p_sent_gb_busiest_link_per_rank"] = [float(v) / 1e9 for v in allr[:, 3]]
            res["rank0_sent_gb_per_peer"] = {str(k): v / 1e9 for k, v in sorted(st["sent_bytes_per_peer"].items())}
        else:
            res["bcast_root_payload_gb_per_rank"] = [float(v) / 1e9 for v in allr[:, 4]]
        barrier()
        per_schedule[sched] = res

    t_first = time.perf_counter()
    measure(schedules[0])
    t_first = time.perf_counter() - t_first
    # first real multi-GPU run: proof that the backend's communicators hold N, Pc and Pr members (world, process row, column)
    g0 = engines[schedules[0]]
    rank_counts = {"world": ranks_seen(), "expected_world": world}
    try:
        rank_counts.update({"process_row": ranks_seen(g0.row_group) if g0.row_group is not None else 1,
                            "process_col": ranks_seen(g0.col_group) if g0.col_group is not None else 1,
                            "expected_row": g0.pc, "expected_col": g0.pr})
    except Exception as exc:
        rank_counts["groups_error"] = repr(exc)
    try:
        rank_counts["rccl_version"] = ".".join(str(v) for v in torch.cuda.nccl.version())
    except Exception:
        pass
    first = schedules[0]
    g = engines[first]
    if rank == 0 and args.partial_line_path:       # on disk before anything else runs (read back by self_launch if the ranks die)
        with open(args.partial_line_path, "w") as fh:
            json.dump({"schedules": per_schedule, "workload": w["name"], "world": world}, fh)
    sec = per_schedule[first]["ms_per_step"] * 1e-3
    lml = per_schedule[first]["lml"]

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
                # both paths refine the quadratic form from 12288 rows on (BlockCyclicGP._refine), so the values
                # are compared as they are; the single-GPU value WITHOUT that step is kept beside them (what the step removes)
                old_env = os.environ.get("GPN_REFINE_MIN_N")
                os.environ["GPN_REFINE_MIN_N"] = "0"
                try:
                    single["lml_without_refinement"] = st().item()
                finally:
                    if old_env is None:
                        del os.environ["GPN_REFINE_MIN_N"]
                    else:
                        os.environ["GPN_REFINE_MIN_N"] = old_env
                del m1
                torch.cuda.empty_cache()
            except Exception as exc:
                notes["single_gpu_same_run_error"] = repr(exc)
        barrier()
        if single is not None:
            extra["single_gpu_same_run"] = single
            extra["lml_abs_diff_vs_single_gpu"] = abs(lml - single["lml"])
            extra["lml_abs_diff_vs_single_gpu_without_refinement"] = abs(lml - single["lml_without_refinement"])
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
            del mr
        except Exception as exc:
            notes["replicas_error"] = repr(exc)
        # ... and GP FITS / s the same way (north_star: "GP-fits/sec at 1, 2, 4 and 8 GPUs"): every GPU steps its own 8 restarts of
        # C2 in lock step (multi_start_optimize: gpn_lml_forward_batched + gpn_lml_backward_batched + one optimiser step per
        # iteration), no collective on the data path; a fit = 50 Adam steps, 10 are timed
        try:
            import contextlib
            from gptorch_amd.models import multi_start_optimize
            wf = WORKLOADS["c2"]
            Bf, iters = 8, 10
            ms_ = []
            for b in range(Bf):
                mb = build_model(dict(wf, variance=wf["variance"] * (1.0 + 0.01 * b), length_scales=wf["length_scales"] * (1.0 + 0.02 * b)),
                                 rank, device)[0]
                if ms_:
                    mb.X, mb.Y = ms_[0].X, ms_[0].Y
                ms_.append(mb)
            with contextlib.redirect_stdout(sys.stderr):
                multi_start_optimize(ms_, method="Adam", max_iter=2, learning_rate=0.01)
                barrier()
                t0 = time.perf_counter()
                multi_start_optimize(ms_, method="Adam", max_iter=iters, learning_rate=0.01)
                barrier()
            tf = max_over_ranks(time.perf_counter() - t0)
            extra["replicas_c2_fit_batched"] = {
                "config": "C2 x %d restarts per GPU in lock step x %d GPUs (independent replicas, rank r = seed r, no collective); a fit = 50 Adam steps, %d timed"
                          % (Bf, world, iters),
                "value": world * Bf / (tf / iters * 50.0), "unit": "GP fits/s", "scaling": "weak", "ms_per_lockstep_step": tf / iters * 1e3,
                "frac_of_fp64_peak_on_N3": world * Bf * iters * float(wf["n"]) ** 3 / tf / 1e12 / (world * PEAK_FP64_MFMA_TFLOPS)}
            del ms_
            torch.cuda.empty_cache()
        except Exception as exc:
            notes["replicas_fit_error"] = repr(exc)

    def line_text():
        best = min(per_schedule, key=lambda k: per_schedule[k]["ms_per_step"])
        r = per_schedule[best]
        sec_b = r["ms_per_step"] * 1e-3
        agg = (w["n"] ** 3 / 3.0) / sec_b / 1e12
        peak = world * PEAK_FP64_MFMA_TFLOPS
        line = {
            "metric": "GP log-marginal-likelihood evals/sec (Cholesky+solve) at NxD fp64",
            "value": 1.0 / sec_b, "unit": "LML evals/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": sec_b * 1e3,
            "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
            "dtype": "f64", "data": "synthetic",
            "config": {"workload": w["name"] + ", ONE model 2-D block-cyclic over %d GPUs" % world, "N": w["n"], "D": w["d"], "dy": w["dy"],
                       "kernel": w["kind"], "parallelism": "block-cyclic %dx%d grid, tile %d, %s, panel exchange schedule '%s' on row/column sub-communicators"
                       % (g.pr, g.pc, g.T, dist.get_backend(), best)},
            "backend": dist.get_backend(), "world_size_reported_by_backend": dist.get_world_size(),
            "ranks_counted_by_collectives": rank_counts,
            "single_factorisation_wall_s": sec_b, "lml": r["lml"], "info": r["info"], "lml_refined": bool(g.refined),
            "exchange_schedule": best, "exchange_schedules": per_schedule,
            "exposed_comm_ms_per_rank": r["exposed_comm_ms_per_rank"],
            "roofline": {"bound": "mfma", "kernel": "whole evaluation, all ranks: N^3/3 flops / wall (the contraction kernel carries all but the leaves)",
                         "achieved": agg, "peak": peak, "unit": "TFLOP/s", "frac": agg / peak, "traffic": None,
                         "peak_note": "%d x %.1f TFLOP/s fp64 MFMA" % (world, PEAK_FP64_MFMA_TFLOPS)},
            "rank0_contraction_ms_per_step": r["contraction_ms_per_rank"][0], "rank0_local_matrix_gb": g.A.numel() * 8 / 1e9,
        }
        gl = golden_lml(w)
        if gl is not None:                         # C2: the reference's value; C4: the CPU oracle's, measured at full size on a GPU box's host
            line[golden_err_key(w)] = abs(r["lml"] - gl)
            line["lml_golden_provenance"] = golden_source(w)
        line.update(extra)
        if "single_gpu_same_run" in extra:         # the strong-scaling number of THIS run: same matrix, same box, same binary
            t1 = extra["single_gpu_same_run"]["ms_per_step"] * 1e-3
            line["speedup_vs_single_gpu_same_run"] = t1 / sec_b
            line["parallel_efficiency_vs_single_gpu_same_run"] = t1 / sec_b / world
        if notes:
            line["notes"] = dict(notes)
        return json.dumps(line)

    # Everything below runs under a watchdog: if a leg does not come back, rank 0 prints the line it has and every rank
    # leaves (a blocked collective cannot be cancelled from Python).  (An exception on one rank only would leave the others
    # in a collective: the watchdog covers that case too.)
    def guarded(label, deadline, fn):
        done = threading.Event()

        def watchdog():
            if not done.wait(deadline):
                notes[label + "_error"] = "no result within %.0f s (watchdog): leg abandoned" % deadline
                # A blocked collective cannot be cancelled from Python: every rank leaves.  Rank 0 first prints (and saves) the
                # line it has; the others wait for that, so the launcher's SIGTERM on the first exit cannot pre-empt the print.
                # The exit code is NON-ZERO (3): a run that gave up on a GPU process is not a success, whatever the line holds.
                if rank == 0:
                    txt = line_text()
                    print(txt, flush=True)
                    if args.partial_line_path:
                        try:
                            with open(args.partial_line_path, "w") as fh:
                                fh.write(txt)
                        except OSError:
                            pass
                else:
                    time.sleep(3.0)
                os._exit(3)
        th = threading.Thread(target=watchdog, daemon=True)
        th.start()
        try:
            fn()
        except Exception as exc:
            notes[label + "_error"] = repr(exc)
        done.set()
        # wait for the watchdog thread to END before going on: its closure references this run's engines and process
        # groups, and if run_multi returned first the thread would drop the last references -- tearing the groups down
        # from a daemon thread that the interpreter kills at exit (SIGABRT seen at 8 ranks; tools/dbg/abort_trace.c)
        th.join()

    wd = os.environ.get("GPN_BENCH_WATCHDOG_S")
    # (1) the remaining exchange schedule(s): the same work took t_first
    for sched in schedules[1:]:
        def second(sched=sched):
            try:
                measure(sched)
            except Exception:
                per_schedule.pop(sched, None)
                raise
        guarded("schedule_%s" % sched, float(wd) if wd else 3.0 * t_first + 45.0, second)
    # (2) GP fits / s at this GPU count (north_star): ONE distributed loss + closed-form backward of the same model (U = L^-T
    # carried on the grid, Kyy^-1 = U U^T with panels of U travelling like factorisation panels, per-rank sweeps, D + 2
    # scalars all-reduced) -- the step of GPModel.optimize's Adam loop (base.py:260-269); a fit of BASELINE configs[2]'s
    # length is 50 of them.  3x the local matrix and about 3x the evaluation's time; --no-dist-backward skips it.
    if not args.no_extras and not args.no_dist_backward:
        def dist_backward():
            best_engine = engines[min(per_schedule, key=lambda k: per_schedule[k]["ms_per_step"])]
            barrier()
            t0 = time.perf_counter()
            lml_b, grad = best_engine.log_likelihood_and_grad(var, ls, nz, Y)
            barrier()
            tb = max_over_ranks(time.perf_counter() - t0)
            extra["dist_loss_backward"] = {
                "config": w["name"].replace("LML eval", "LML + closed-form gradients") + ", block-cyclic over %d GPUs, exchange schedule '%s'" % (world, best_engine.schedule),
                "ms_per_step": tb * 1e3, "lml": float(lml_b.item()), "grads_constrained": [float(v) for v in grad.tolist()],
                "note": "first call: includes the allocation of the identity rows and of the Kyy^-1 accumulator"}
            extra["fits_per_s_estimate"] = 1.0 / (50.0 * tb)
            extra["fits_per_s_estimate_note"] = "1 / (50 x one distributed loss+backward): a fit = 50 Adam steps (BASELINE configs[2]); measured end to end only at N = 1 (c3_adam50)"
        guarded("dist_backward", float(wd) if wd else 12.0 * sec * (args.steps + args.warmup + 1) + 60.0, dist_backward)
    if rank == 0:
        print(line_text(), flush=True)


def failure_line(args, world, error, extra=None):
    """never leave the driver without a line: an unmeasured run says so (value null) with the reason."""
    wname = WORKLOADS[args.workload or ("c3" if world == 1 else "c4")]["name"]
    line = {"metric": "GP log-marginal-likelihood evals/sec (Cholesky+solve) at NxD fp64", "value": None,
            "unit": "LML evals/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": None,
            "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": wname + (", ONE model 2-D block-cyclic over %d GPUs" % world if world > 1 else "")},
            "error": error}
    if extra:
        line.update(extra)
    return json.dumps(line)


def self_launch(args):
    """`python bench.py --gpus N` without a launcher: run `python -m torch.distributed.run --nproc-per-node N bench.py ...`
    as a child, pass its output through, return its exit code.  If the child dies without a line (a crash or a hang in
    the second exchange schedule, say), print one from what rank 0 had already saved."""
    import socket
    import subprocess
    import tempfile
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    partial = os.path.join(tempfile.gettempdir(), "gpn_bench_partial_%d.json" % os.getpid())
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus),
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:] + \
          ["--partial-line-path", partial]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    proc = subprocess.Popen(cmd, stdout=subprocess.PIPE, env=env, text=True, cwd=ROOT)
    got_line = False
    for ln in proc.stdout:
        sys.stdout.write(ln)
        sys.stdout.flush()
        if ln.lstrip().startswith("{") and '"metric"' in ln:
            got_line = True
    rc = proc.wait()
    if not got_line:
        saved = None
        try:
            saved = json.load(open(partial))
        except Exception:
            pass
        print(failure_line(args, args.gpus, "the %d-rank child (torch.distributed.run) exited with code %d without a result line" % (args.gpus, rc),
                           {"partial": saved} if saved else None), flush=True)
    try:
        os.remove(partial)
    except OSError:
        pass
    return rc


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=None)
    ap.add_argument("--warmup", type=int, default=None)
    ap.add_argument("--workload", default=None, choices=sorted(WORKLOADS),
                    help="default: c3 on one GPU; c4 (block-cyclic) on several")
    ap.add_argument("--tile", type=int, default=2048, help="block-cyclic tile size (N > 1 GPUs)")
    ap.add_argument("--dist-backward", action="store_true", help=argparse.SUPPRESS)      # (now the default; kept for old command lines)
    ap.add_argument("--no-dist-backward", action="store_true", help="(N > 1 GPUs) skip the distributed loss + backward leg")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-fit", action="store_true", help="skip the 50-Adam-step fit leg (c3_adam50: about 30 s)")
    ap.add_argument("--cpu-sample-only", action="store_true", help="cpu_baseline from the 8192-row sample only (extrapolated), "
                    "skipping the full-size CPU evaluation (about 2-4 minutes at C3)")
    ap.add_argument("--cpu-baseline-budget", type=float, default=90.0,
                    help="seconds the full-size CPU leg may take (1 warm-up + up to 3 timed evaluations; fewer timed ones if "
                         "3 would overrun it; 0 = always 3)")
    ap.add_argument("--no-extras", action="store_true", help="headline + rooflines only (profiling runs)")
    ap.add_argument("--test-shared-gpu", action="store_true",
                    help="(testing the multi-rank control flow on a 1-GPU box) every rank uses cuda:0, gloo collectives")
    ap.add_argument("--schedule", default="both", choices=["both", "bcast", "mesh"],
                    help="(N > 1 GPUs) panel exchange: the backend's broadcast, grouped point-to-point over the direct "
                         "links, or both timed in this run (headline = the better one)")
    ap.add_argument("--partial-line-path", default=None, help=argparse.SUPPRESS)
    args = ap.parse_args()

    if args.gpus > 1 and "RANK" not in os.environ and "WORLD_SIZE" not in os.environ:
        # started the way the N = 1 run is started (`python bench.py --gpus N`): launch the ranks ourselves, as a CHILD
        # process, before this process has made any GPU call (a process that has initialised the GPU must not exec),
        # relay its JSON line and exit code
        sys.exit(self_launch(args))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if args.gpus != world:
        sys.exit("bench.py: --gpus %d but WORLD_SIZE=%d: launch N>1 with `python -m torch.distributed.run --nproc-per-node N "
                 "--master-addr 127.0.0.1 bench.py --gpus N` (or plain `python bench.py --gpus N`, which does that itself)"
                 % (args.gpus, world))
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
            print(failure_line(args, world, repr(exc)), flush=True)
        raise
    finally:
        try:
            dist.destroy_process_group()
        except Exception:
            pass


if __name__ == "__main__":
    main()

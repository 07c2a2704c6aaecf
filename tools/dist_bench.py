#!/usr/bin/env python3
"""2-D block-cyclic LML (gptorch_amd/dist.py) timing.
Single GPU:   python tools/dist_bench.py 16384 16 2048
8 GPUs:       python -m torch.distributed.run --nproc-per-node 8 --master-addr 127.0.0.1 tools/dist_bench.py 65536 32 2048
Env: GPN_SHARED_GPU=1   every rank on cuda:0 with gloo collectives (multi-rank orchestration on a 1-GPU box)
     GPN_DIST_GRAD=1    also the distributed closed-form backward
     GPN_FORCE_COMM=1   issue the row/column collectives even in single-member groups (drives the RCCL calls at world 1)
     GPN_PHANTOM=r/w    do the work of rank r of a w-rank grid with the collectives skipped (timing only)
     GPN_NATIVE=1       also time the single-GPU native factorisation of the same matrix
     GPN_DIST_PREDICT=1 (with GPN_CDRIVER=1) also gpn_dist_predict against the single-GPU prediction
     GPN_CDRIVER=1      also the C-ABI driver (gpn_dist_lml_forward) over torch.distributed callbacks,
                        or with GPN_RCCL=1 over its own RCCL communicators (libgpnative_rccl.so) -- the form for real multi-GPU runs"""
import os, sys, time
import torch
import torch.distributed as dist
ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, ROOT)
from gptorch_amd import dist as gdist, rng  # noqa: E402

n, d, T = (int(a) for a in (sys.argv[1:4] + ["16384", "16", "2048"][len(sys.argv) - 1:]))
world = int(os.environ.get("WORLD_SIZE", "1"))
local = int(os.environ.get("LOCAL_RANK", "0"))
shared = os.environ.get("GPN_SHARED_GPU") == "1"
if shared:
    local = 0
torch.cuda.set_device(local)
dev = torch.device("cuda", local)
if "RANK" in os.environ:                      # under torch.distributed.run, also with one process
    if shared:
        dist.init_process_group("gloo")
    else:
        dist.init_process_group("nccl", device_id=dev)
x, y = rng.make_regression(n, d, 1, seed=0)
X, Y = torch.tensor(x, device=dev), torch.tensor(y, device=dev)
phantom = None
if os.environ.get("GPN_PHANTOM"):
    r, w = os.environ["GPN_PHANTOM"].split("/")
    phantom = (int(r), int(w))
g = gdist.BlockCyclicGP(X, Y, "Rbf", tile=T, phantom=phantom, force_comm=os.environ.get("GPN_FORCE_COMM") == "1")
one = torch.ones(1, dtype=torch.float64, device=dev)
ls = one * float(d) ** 0.5
for it in range(3):
    torch.cuda.synchronize()
    t0 = time.time()
    if phantom:
        g.assemble(one, ls, 0.01 * one, Y)
        g.factor()
        lml = torch.zeros(())
    else:
        lml = g.log_likelihood(one, ls, 0.01 * one, Y)
    torch.cuda.synchronize()
    dt = time.time() - t0
    if g.rank == 0 or phantom:
        print("N=%d D=%d T=%d world=%d grid=%dx%d backend=%s%s: lml=%.8f  %.1f ms  (%.1f TFLOP/s aggregate on N^3/3)" % (
            n, d, T, g.world, g.pr, g.pc, dist.get_backend() if dist.is_initialized() else "none",
            " PHANTOM rank %d" % g.rank if phantom else "", lml.item(), dt * 1e3, n ** 3 / 3 / dt / 1e12), flush=True)
if os.environ.get("GPN_DIST_GRAD") == "1":      # distributed closed-form backward on the same grid
    torch.cuda.synchronize()
    t0 = time.time()
    lml, grad = g.log_likelihood_and_grad(one, ls, 0.01 * one, Y)
    torch.cuda.synchronize()
    if g.rank == 0:
        print("grad: lml=%.8f  %s  %.1f ms (factorisation carrying L^-T + K^-1 = U U^T + sweeps)" % (
            lml.item(), " ".join("%.10e" % v for v in grad.tolist()), (time.time() - t0) * 1e3), flush=True)
if os.environ.get("GPN_NATIVE") == "1" and g.rank == 0:
    from gptorch_amd import _ops
    f = None
    for it in range(3):
        torch.cuda.synchronize()
        t0 = time.time()
        f, terms = _ops.lml_forward("Rbf", X, Y, one, ls, 0.01 * one, factor=f)
        torch.cuda.synchronize()
        print("native single-GPU: lml=%.8f  %.1f ms" % (terms[2].item(), (time.time() - t0) * 1e3), flush=True)
if os.environ.get("GPN_CDRIVER") == "1":
    c = gdist.NativeDistLML(X, Y, "Rbf", tile=T, comm="rccl" if os.environ.get("GPN_RCCL") == "1" else "torch",
                            force_comm=os.environ.get("GPN_FORCE_COMM") == "1")
    for it in range(2):
        torch.cuda.synchronize()
        t0 = time.time()
        lml = c.log_likelihood(one, ls, 0.01 * one)
        torch.cuda.synchronize()
        if c.rank == 0:
            print("cdriver: lml=%.8f  %.1f ms  grid=%dx%d" % (lml.item(), (time.time() - t0) * 1e3, c.pr, c.pc), flush=True)
    if os.environ.get("GPN_DIST_GRAD") == "1":      # forward + closed-form backward in one C call (gpn_dist_lml_grad)
        torch.cuda.synchronize()
        t0 = time.time()
        lml, grad, g_resid = c.log_likelihood_and_grad(one, ls, 0.01 * one)
        torch.cuda.synchronize()
        if c.rank == 0:
            print("cdriver grad: lml=%.8f  %s  resid_grad_norm=%.10e  %.1f ms" % (
                lml.item(), " ".join("%.10e" % v for v in grad.tolist()), g_resid.norm().item(), (time.time() - t0) * 1e3), flush=True)
    if os.environ.get("GPN_DIST_PREDICT") == "1":   # gpn_dist_predict vs the single-GPU predict of the same model (rank 0 prints)
        xs = torch.tensor(rng.normal(9, (40, d)), device=dev)
        ms = torch.tensor(rng.normal(10, (40, 1)), device=dev)
        mu, var = c.predict(one, ls, 0.01 * one, xs, mean_new=ms, diag=True)
        _, cov = c.predict(one, ls, 0.01 * one, xs, mean_new=ms, diag=False)
        if c.rank == 0:
            from gptorch_amd import _ops
            f = _ops.kernel_factor("Rbf", X, one, ls, 0.01 * one, R=Y)
            mu1, var1 = _ops.gpr_predict("Rbf", X, xs, one, ls, f, diag=True, mean_new=ms)
            _, cov1 = _ops.gpr_predict("Rbf", X, xs, one, ls, f, diag=False, mean_new=ms)
            print("cdriver predict: mean_err=%.3e var_err=%.3e cov_err=%.3e" % (
                (mu - mu1).abs().max().item(), (var - var1).abs().max().item(), (cov - cov1).abs().max().item()), flush=True)
if dist.is_initialized():
    dist.destroy_process_group()

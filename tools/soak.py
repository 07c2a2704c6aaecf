#!/usr/bin/env python3
"""GPU-box tool: determinism / race soak.  The same evaluation repeated many times must give the
bitwise-identical LML and gradients every time (the leaf kernel hands blocks between its pivot
wave and its tile waves through LDS with one hardware and one software barrier per panel, and
the factorisation forks onto a second stream: a race would show up as a flipped last bit).
usage: soak.py [evaluations at N=8192 = 1500] [evaluations at N=1000 = 4000] [evaluations at N=16384 = 0]
(the third leg is the one that reaches the 8-wave 128x128 contraction tile: 7260 tiles in its first trailing update)"""
import os
import sys
import time

import torch

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from gptorch_amd import kernels, likelihoods, rng  # noqa: E402
from gptorch_amd.models import GPR, batched_log_likelihood  # noqa: E402

dev = torch.device("cuda:0")
n_big = int(sys.argv[1]) if len(sys.argv) > 1 else 1500
n_small = int(sys.argv[2]) if len(sys.argv) > 2 else 4000
n_large = int(sys.argv[3]) if len(sys.argv) > 3 else 0


def soak(model, reps, label, grads_every=0):
    ref = None
    t0 = time.perf_counter()
    bad = 0
    for i in range(reps):
        if grads_every and i % grads_every == 0:
            model.zero_grad()
            loss = model.loss()
            loss.backward()
            cur = torch.cat([-loss.detach().reshape(1)] + [p.grad.reshape(-1) for p in model.parameters() if p.grad is not None])
            key = "g"
        else:
            with torch.no_grad():
                cur = model.log_likelihood().reshape(1)
            key = "f"
        cur = cur.cpu().numpy().tobytes()
        if ref is None:
            ref = {}
        if key not in ref:
            ref[key] = cur
        elif ref[key] != cur:
            bad += 1
    print("%s: %d evaluations, %d differ from the first, %.1f s" % (label, reps, bad, time.perf_counter() - t0), flush=True)
    return bad


bad = 0
m2, _, _ = bench.build_model(bench.WORKLOADS["c2"], 0, dev)
bad += soak(m2, n_big, "N=8192 D=8 Rbf", grads_every=10)
x, y = rng.make_regression(1000, 5, 2, seed=4)
m1 = GPR(x, y, kernels.Matern52(5, variance=1.1, length_scales=1.5), likelihood=likelihoods.Gaussian(variance=0.02))
m1.cuda()
bad += soak(m1, n_small, "N=1000 D=5 dy=2 Matern52 (ragged)", grads_every=7)
if n_large:
    w = dict(name="n16384", kind="Matern52", n=16384, d=8, dy=1, variance=1.0, length_scales=8.0 ** 0.5, noise=1e-2)
    ml, _, _ = bench.build_model(w, 3, dev)
    bad += soak(ml, n_large, "N=16384 D=8 Matern52 (8-wave tile)", grads_every=10)
    del ml
    torch.cuda.empty_cache()
if n_large:
    # the block-cyclic engine (one rank) with the refinement step on the grid: tile inversions, the serial sweep's chunked
    # column sums, a share of the double-double residual pass -- and the same step inside the library (gpn_dist_lml_refine)
    from gptorch_amd import dist as gdist
    w = dict(n=16384, d=8)
    xg, yg = rng.make_regression(w["n"], w["d"], 1, seed=3)
    X, Y = torch.tensor(xg, device=dev), torch.tensor(yg, device=dev)
    tt = lambda v: torch.tensor([v], dtype=torch.float64, device=dev)
    for label, eng in (("block-cyclic engine + _refine", gdist.BlockCyclicGP(X, Y, "Matern52", tile=2048)),
                       ("C driver + gpn_dist_lml_refine", gdist.NativeDistLML(X, Y, "Matern52", tile=2048))):
        eng.refine = True
        ref, nb, t0 = None, 0, time.perf_counter()
        reps = max(1, n_large // 3)
        for i in range(reps):
            args = (tt(1.0), tt(8.0 ** 0.5), tt(1e-2)) + ((Y,) if isinstance(eng, gdist.BlockCyclicGP) else ())
            cur = eng.log_likelihood(*args).reshape(1).cpu().numpy().tobytes()
            if ref is None:
                ref = cur
            elif cur != ref:
                nb += 1
        print("N=16384 %s: %d evaluations, %d differ from the first, %.1f s" % (label, reps, nb, time.perf_counter() - t0), flush=True)
        bad += nb
        del eng
        torch.cuda.empty_cache()
# two evaluations in flight on two streams
models = [bench.build_model(bench.WORKLOADS["c2"], 50 + r, dev)[0] for r in range(4)]
ref = None
t0 = time.perf_counter()
nb = 0
for i in range(max(1, n_big // 8)):
    out = torch.cat(batched_log_likelihood(models)).cpu().numpy().tobytes()
    if ref is None:
        ref = out
    elif out != ref:
        nb += 1
print("4 restarts on two lanes: %d rounds, %d differ, %.1f s" % (max(1, n_big // 8), nb, time.perf_counter() - t0), flush=True)
bad += nb
# lock-step loss + backward (round 5): gpn_lml_forward_batched + gpn_lml_backward_batched over 4 restarts sharing the data
from gptorch_amd.models import batched_loss_and_grad  # noqa: E402
x2, y2 = rng.make_regression(4096, 6, 1, seed=8)
X2, Y2 = torch.tensor(x2, device=dev), torch.tensor(y2, device=dev)
restarts = []
for r in range(4):
    mr = GPR(X2, Y2, kernels.Matern52(6, variance=1.0 + 0.1 * r, length_scales=2.0 + 0.3 * r), likelihood=likelihoods.Gaussian(variance=0.02))
    mr.cuda()
    mr.X, mr.Y = X2, Y2
    restarts.append(mr)
ref, nb, t0 = None, 0, time.perf_counter()
rounds = max(1, n_big // 5)
for i in range(rounds):
    for mr in restarts:
        mr.zero_grad()
    losses = batched_loss_and_grad(restarts)
    cur = torch.cat([l.reshape(-1) for l in losses] + [p.grad.reshape(-1) for mr in restarts for p in mr.parameters() if p.grad is not None]).cpu().numpy().tobytes()
    if ref is None:
        ref = cur
    elif cur != ref:
        nb += 1
print("N=4096 x 4 restarts, lock-step loss + backward: %d rounds, %d differ, %.1f s" % (rounds, nb, time.perf_counter() - t0), flush=True)
bad += nb
# round 6: sparse models in lock step (the batched kernel matrix / right-solve / inversion / sweeps, the fixed-order scalar sums) and
# a ragged group (identity-padded assembly, masked sweeps)
def soak_group(models_, rounds_, label):
    ref_, nb_, t0_ = None, 0, time.perf_counter()
    for _ in range(rounds_):
        for mm in models_:
            mm.zero_grad()
        ls_ = batched_loss_and_grad(models_)
        cur_ = torch.cat([l.reshape(-1) for l in ls_] + [p.grad.reshape(-1) for mm in models_ for p in mm.parameters() if p.grad is not None])
        cur_ = cur_.cpu().numpy().tobytes()
        if ref_ is None:
            ref_ = cur_
        elif cur_ != ref_:
            nb_ += 1
    print("%s: %d rounds, %d differ, %.1f s" % (label, rounds_, nb_, time.perf_counter() - t0_), flush=True)
    return nb_


from gptorch_amd import mean_functions  # noqa: E402
from gptorch_amd.models import VFE  # noqa: E402
import numpy as np  # noqa: E402
xv, yv = rng.make_regression(2048, 4, 1, seed=12)
gz = np.random.default_rng(1)
sparse = []
for r in range(8):
    v = VFE(xv, yv, kernels.Rbf(4, variance=1.0 + 0.05 * r, length_scales=1.0 + 0.05 * r), inducing_points=xv[gz.choice(2048, 256, replace=False)],
            likelihood=likelihoods.Gaussian(variance=0.05), mean_function=mean_functions.Zero(1))
    v.cuda()
    sparse.append(v)
for v in sparse[1:]:
    v.X, v.Y = sparse[0].X, sparse[0].Y
bad += soak_group(sparse, max(1, n_big // 5), "N=2048 M=256 x 8 sparse restarts, lock-step loss + backward")
folds = []
for r, nn in enumerate((3000, 2999, 2950, 2800, 2501)):
    xf, yf = rng.make_regression(nn, 5, 1, seed=30 + r)
    mf = GPR(xf, yf, kernels.Matern52(5, variance=1.0, length_scales=2.0 + 0.1 * r), likelihood=likelihoods.Gaussian(variance=0.03))
    mf.cuda()
    folds.append(mf)
bad += soak_group(folds, max(1, n_big // 5), "N=2501..3000 x 5 as one ragged group, lock-step loss + backward")
print("SOAK", "OK" if bad == 0 else "MISMATCHES %d" % bad)
sys.exit(0 if bad == 0 else 1)

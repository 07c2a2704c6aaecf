"""Lock-step loss + backward + optimiser step over several independent GPR restarts (round 5): the reference's training
loop is `loss(); backward(); step()` ONE model at a time (gptorch/models/base.py:260-269, gptorch/models/gpr.py:47-67);
gpn_lml_backward_batched / batched_loss_and_grad / multi_start_optimize run B models of one shape through every launch
together.  The bar: every model's numbers are BIT-IDENTICAL to its own sequential evaluation, and the goldens generated
from the reference hold through the batched path at the same tolerances."""
import contextlib
import io

import numpy as np
import pytest
import torch

from tests._util import load_json
from gptorch_amd import kernels, likelihoods, mean_functions, rng
from gptorch_amd.models import GPR, batched_log_likelihood, batched_loss_and_grad, multi_start_optimize

pytestmark = pytest.mark.gpu

KERN = {"Rbf": kernels.Rbf, "Matern52": kernels.Matern52, "Matern32": kernels.Matern32, "Exp": kernels.Exp}


def _quiet():
    return contextlib.redirect_stdout(io.StringIO())


@pytest.mark.parametrize("n,d,dy,batch,kind,ard,shared", [
    (1500, 5, 1, 5, "Matern52", True, True),       # ragged last block, uneven inversion tree (single nodes per level)
    (1024, 8, 2, 3, "Rbf", False, False),          # power-of-two tree: equal nodes x models as two-level batches; dy = 2
    (200, 2, 1, 7, "Rbf", False, True),            # n <= 256: the recursive inversion, model by model inside the call
    (128, 3, 1, 4, "Matern32", False, False),      # one leaf
    (2176, 4, 1, 2, "Rbf", False, True),           # 17 leaves
    (5000, 3, 2, 3, "Matern52", False, True),      # ragged, two right-hand sides
    (4096, 20, 1, 3, "Exp", True, False),          # d > 16: two coordinate chunks in the sweep, per-dimension sums
])
def test_lockstep_backward_is_bit_identical_to_sequential(device, n, d, dy, batch, kind, ard, shared):
    """gpn_lml_backward_batched on the factors of gpn_lml_forward_batched: every model's constrained gradients and
    dLML/d(y - m) are BIT-IDENTICAL to gpn_lml_backward on that model's own factor (closed form of SURVEY 8(a) a9)."""
    from gptorch_amd import _backward, _ops
    g = torch.Generator().manual_seed(n + batch)
    xs, ys = [], []
    for b in range(batch):
        x, y = rng.make_regression(n, d, dy, seed=3 if shared else 3 + b)
        xs.append(torch.as_tensor(x).to(device))
        ys.append(torch.as_tensor(y).to(device))
    var = (0.5 + torch.rand(batch, generator=g, dtype=torch.float64)).to(device)
    ls = (0.7 + torch.rand(batch, d if ard else 1, generator=g, dtype=torch.float64)).to(device) * float(np.sqrt(d))
    nz = (0.01 + 0.05 * torch.rand(batch, generator=g, dtype=torch.float64)).to(device)
    X = xs[0] if shared else torch.stack(xs)
    R = ys[0] if shared else torch.stack(ys)
    fb, terms = _ops.lml_forward_batched(kind, X, R, var, ls, nz)
    assert int(fb.info.cpu().abs().max()) == 0
    grads, g_R = _ops.lml_backward_batched(kind, X, var, ls, fb, need_resid=True)
    nls = ls.shape[1]
    for b in range(batch):
        f, t = _ops.lml_forward(kind, xs[b], ys[b], var[b:b + 1], ls[b], nz[b:b + 1], refine=False)
        gv, gl, gn, gr = _backward.lml_backward(kind, xs[b], var[b:b + 1], ls[b], nz[b:b + 1], f)
        assert torch.equal(grads[b, 0:1], gv), (b, grads[b], gv)
        assert torch.equal(grads[b, 1:1 + nls], gl), (b, grads[b], gl)
        assert torch.equal(grads[b, 1 + nls:], gn), (b, grads[b], gn)
        assert torch.equal(g_R[b], gr), b
    # and the same call once more into the same workspace (a fit loop reuses it): same bits
    grads2, _ = _ops.lml_backward_batched(kind, X, var, ls, fb, need_resid=False)
    assert torch.equal(grads, grads2)


def _restarts(device, n, d, specs, seed=5, dy=1, mean=None):
    x, y = rng.make_regression(n, d, dy, seed=seed)
    X, Y = torch.as_tensor(x).to(device), torch.as_tensor(y).to(device)
    ms = []
    for kind, ard, var, ell, nzv in specs:
        ls = np.full(d, ell) * (1.0 + 0.1 * np.arange(d)) if ard else ell
        mf = None if mean is None else mean_functions.Constant(dy, val=torch.full((dy,), mean, dtype=torch.float64))
        m = GPR(X, Y, KERN[kind](d, variance=var, length_scales=ls, ARD=ard), likelihood=likelihoods.Gaussian(variance=nzv), mean_function=mf)
        m.cuda()
        m.X, m.Y = X, Y                            # restarts over ONE data set: the same device tensors
        ms.append(m)
    return ms


def _grads(m):
    return [None if p.grad is None else p.grad.clone() for p in m.parameters()]


MIXED = [("Rbf", False, 1.0, 1.5, 0.02), ("Matern52", False, 0.8, 2.0, 0.03), ("Rbf", False, 1.3, 1.1, 0.05),
         ("Matern52", False, 1.1, 1.7, 0.01), ("Rbf", True, 0.9, 1.4, 0.02), ("Rbf", True, 1.2, 1.9, 0.04),
         ("Exp", False, 1.0, 2.5, 0.02)]                     # two Rbf, two Matern52, two ARD Rbf: three groups; one singleton


def test_batched_loss_and_grad_is_bit_identical_with_several_groups_in_one_call(device):
    """three lock-step groups of EQUAL count, N and dy in one call (different kernel kind / ARD: the case in which round 4's
    buffer cache handed two groups the same buffers) plus a singleton: every loss and every .grad equals the model's own
    loss(); backward() bit for bit (base.py:260-269)."""
    ms = _restarts(device, 900, 3, MIXED)
    seq_loss, seq_grads = [], []
    for m in ms:
        m.zero_grad()
        loss = m.loss()
        loss.backward()
        seq_loss.append(loss.detach().clone())
        seq_grads.append(_grads(m))
        m.zero_grad()
    out = batched_loss_and_grad(ms)
    for i, m in enumerate(ms):
        assert out[i].shape == (1,) and torch.equal(out[i], seq_loss[i]), (i, out[i], seq_loss[i])
        for ga, gb in zip(_grads(m), seq_grads[i]):
            assert (ga is None) == (gb is None)
            if ga is not None:
                assert torch.equal(ga, gb), (i, ga, gb)
    # gradients ACCUMULATE like backward(): a second call doubles them
    batched_loss_and_grad(ms)
    for i, m in enumerate(ms):
        for ga, gb in zip(_grads(m), seq_grads[i]):
            if ga is not None:
                assert torch.equal(ga, gb + gb)
    # the forward-only entry point with the same three groups (round-4 advice, high): values of the right models
    vals = batched_log_likelihood(ms)
    for i, m in enumerate(ms):
        assert torch.equal(vals[i], -seq_loss[i]), i


def test_batched_loss_and_grad_trainable_mean_and_fixed_parameters(device):
    """a trainable Constant mean function (dLML/d(y - m) flows back through the stacked residuals) and a parameter frozen
    in every model: same gradients as the sequential path, nothing for the frozen one."""
    ms = _restarts(device, 700, 2, [("Rbf", False, 1.0, 1.2, 0.02), ("Rbf", False, 0.7, 0.9, 0.04), ("Rbf", False, 1.4, 1.6, 0.03)],
                   dy=2, mean=0.3)
    for m in ms:
        m.mean_function.val.requires_grad_(True)
        m.kernel.variance.requires_grad_(False)
    seq = []
    for m in ms:
        m.loss().backward()
        seq.append(_grads(m))
        m.zero_grad()
    batched_loss_and_grad(ms)
    for m, ref in zip(ms, seq):
        assert m.kernel.variance.grad is None
        for (name, p), gb in zip(m.named_parameters(), ref):
            if gb is None:
                continue
            if name.startswith("mean_function"):
                # the mean's gradient sums n entries of -a in a different reduction (stack + sum vs per-model sum)
                assert torch.allclose(p.grad, gb, rtol=1e-12, atol=1e-12), (name, p.grad, gb)
            else:
                assert torch.equal(p.grad, gb), (name, p.grad, gb)


def test_lockstep_backward_replays_the_ladder_per_failing_model(device):
    """one model of the batch is singular at its own noise level: the forward replays THAT model through the jitter ladder
    (functions.py:20-43) into a private factor, the backward runs on it; all models match their sequential gradients."""
    n, d = 600, 2
    x, y = rng.make_regression(n, d, 1, seed=21)
    xdup = np.array(x)
    xdup[300:] = xdup[:300]
    ms = []
    for b in range(4):
        m = GPR(xdup if b == 2 else x, y, kernels.Rbf(d, variance=1.0 + 0.1 * b, length_scales=1.3), likelihood=likelihoods.Gaussian(variance=0.03))
        m.cuda()
        if b == 2:
            m.likelihood.variance.data.fill_(-80.0)
        ms.append(m)
    seq_loss, seq = [], []
    for m in ms:
        loss = m.loss()
        loss.backward()
        seq_loss.append(loss.detach().clone())
        seq.append(_grads(m))
        m.zero_grad()
    assert ms[2]._holder["factor"].jitter_rung >= 0
    out = batched_loss_and_grad(ms)
    for i, m in enumerate(ms):
        assert torch.equal(out[i], seq_loss[i]), i
        for ga, gb in zip(_grads(m), seq[i]):
            assert (ga is None) == (gb is None)
            if ga is not None:
                assert torch.equal(ga, gb), (i, ga, gb)


def test_c2_gradient_golden_through_the_lockstep_path(device):
    """BASELINE config 2 at FULL size (N = 8192, D = 8, Rbf) as one member of a batch of three restarts: the reference's
    LML (1e-8 absolute) and d loss / d raw parameters (1e-8, tests/golden/lml_c2_grad.json) through gpn_lml_forward_batched
    + gpn_lml_backward_batched."""
    case = load_json("lml_c2_grad.json")
    specs = [("Rbf", False, 0.7, 3.5, 0.02), ("Rbf", False, case["variance"], case["length_scales"], case["noise"]), ("Rbf", False, 1.4, 2.2, 0.05)]
    ms = _restarts(device, case["n"], case["d"], specs, seed=0)
    assert rng.checksum(ms[1].X.cpu().numpy()) == case["x_checksum"] and rng.checksum(ms[1].Y.cpu().numpy()) == case["y_checksum"]
    out = batched_loss_and_grad(ms)
    m = ms[1]
    assert abs(-out[1].item() - case["lml"]) < 1e-8
    for name, g in [("kernel.variance", m.kernel.variance.grad), ("kernel.length_scales", m.kernel.length_scales.grad),
                    ("likelihood.variance", m.likelihood.variance.grad)]:
        ref = np.asarray(case["grad_loss"][name])
        err = np.max(np.abs(g.cpu().numpy() - ref) / np.maximum(1.0, np.abs(ref)))
        assert err < 1e-8, (name, g, ref)
    # and bit-identical to the model alone
    ref_grads = _grads(m)
    m.zero_grad()
    loss = m.loss()
    loss.backward()
    assert torch.equal(loss.detach(), out[1])
    for ga, gb in zip(_grads(m), ref_grads):
        assert (ga is None) == (gb is None)
        if ga is not None:
            assert torch.equal(ga, gb)


@pytest.mark.parametrize("method", ["Adam", "SGD", "RMSprop"])
def test_multi_start_optimize_follows_each_models_own_trajectory(device, method):
    """multi_start_optimize (one lock-step loss + backward + ONE optimiser step on the stacked raw parameters per
    iteration) against GPModel.optimize restart by restart (base.py:111-296).  Loss and gradients are bit-identical given
    equal parameters (tests above); PyTorch's multi-tensor optimiser kernels round the update itself with or without an FMA
    depending on a tensor's size / alignment (1 ulp after 2 Adam steps, [4, 1] vs [1]), so the trajectories are held to
    1e-10 relative, two orders inside the goldens' 1e-8."""
    specs = [("Matern52", True, 1.0, 1.5, 0.02), ("Matern52", True, 0.6, 2.5, 0.05), ("Matern52", True, 1.5, 1.0, 0.01),
             ("Matern52", True, 1.2, 3.0, 0.03)]
    a = _restarts(device, 640, 4, specs)
    b = _restarts(device, 640, 4, specs)
    steps = 12
    with _quiet():
        losses, _ = multi_start_optimize(a, method=method, max_iter=steps)
    assert losses.shape == (4, steps)
    for i, m in enumerate(b):
        with _quiet():
            ref, _ = m.optimize(method=method, max_iter=steps, verbose=False)
        assert np.max(np.abs(losses[i] - ref) / np.maximum(1.0, np.abs(ref))) < 1e-10, (i, losses[i] - ref)
        assert losses[i][0] == ref[0]                      # before the first optimiser step: bit for bit
        for pa, pb in zip(a[i].parameters(), m.parameters()):
            assert torch.allclose(pa.data, pb.data, rtol=1e-10, atol=1e-12), (i, pa, pb)


def test_multi_start_with_per_model_optimisers_is_bitwise(device):
    """multi_start_optimize(stacked=False): every restart keeps its own optimiser over its own parameter tensors (the layouts of
    its own optimize()), only the evaluation is shared -- the 12-step Adam trajectories that differ by one ulp through the
    stacked path are bit-identical here."""
    specs = [("Matern52", True, 1.0, 1.5, 0.02), ("Matern52", True, 0.6, 2.5, 0.05), ("Matern52", True, 1.5, 1.0, 0.01),
             ("Matern52", True, 1.2, 3.0, 0.03)]
    a = _restarts(device, 640, 4, specs)
    b = _restarts(device, 640, 4, specs)
    with _quiet():
        losses, _ = multi_start_optimize(a, method="Adam", max_iter=12, stacked=False)
    for i, m in enumerate(b):
        with _quiet():
            ref, _ = m.optimize(method="Adam", max_iter=12, verbose=False)
        assert np.array_equal(losses[i], ref), (i, losses[i] - ref)
        for pa, pb in zip(a[i].parameters(), m.parameters()):
            assert torch.equal(pa.data, pb.data), i


def test_adam_trajectory_golden_through_multi_start(device):
    """the reference's 50-step Adam trajectories (tests/golden/adam_cases.json; base.py:149-151, 260-269), each run as one
    of three restarts stepped in lock step: the golden restart's losses and final parameters at the sequential tolerances."""
    for case in load_json("adam_cases.json"):
        d = case["d"]
        x, y = rng.make_regression(case["n"], d, case["dy"], seed=0)
        X, Y = torch.as_tensor(x).to(device), torch.as_tensor(y).to(device)
        ms = []
        for scale in (0.7, 1.0, 1.6):
            ls = np.asarray(case["length_scales"], dtype=np.float64) * scale * (np.ones(d) if case["ARD"] else 1.0)
            m = GPR(X, Y, KERN[case["kind"]](d, variance=case["variance"] * scale, length_scales=ls, ARD=case["ARD"]),
                    likelihood=likelihoods.Gaussian(variance=case["noise"]))
            m.cuda()
            m.X, m.Y = X, Y
            ms.append(m)
        with _quiet():
            losses, _ = multi_start_optimize(ms, method="Adam", max_iter=50)
        ref = np.asarray(case["losses"])
        assert np.max(np.abs(losses[1] - ref) / np.maximum(1.0, np.abs(ref))) < 1e-8, case["name"]
        m = ms[1]
        for name, p in [("kernel.variance", m.kernel.variance), ("kernel.length_scales", m.kernel.length_scales),
                        ("likelihood.variance", m.likelihood.variance)]:
            assert np.max(np.abs(p.detach().cpu().numpy() - np.asarray(case["final"][name]))) < 1e-8, (case["name"], name)


def test_multi_start_falls_back_for_what_cannot_run_in_lock_step(device):
    """LBFGS (a global line search per model), composite kernels and singletons are optimised by their own optimize()."""
    ms = _restarts(device, 300, 2, [("Rbf", False, 1.0, 1.0, 0.05), ("Rbf", False, 0.8, 1.4, 0.05)])
    ref = _restarts(device, 300, 2, [("Rbf", False, 1.0, 1.0, 0.05), ("Rbf", False, 0.8, 1.4, 0.05)])
    with _quiet():
        losses, _ = multi_start_optimize(ms, method="LBFGS", max_iter=3)
        for i, m in enumerate(ref):
            r, _ = m.optimize(method="LBFGS", max_iter=3, verbose=False)
            assert np.array_equal(losses[i, :len(r)], r)
    with pytest.raises(ValueError):
        with _quiet():
            multi_start_optimize(ms, method="NoSuchOptimiser", max_iter=1)


def test_batch_buffers_are_bounded_and_releasable(device):
    """round-4 advice: the lock-step buffer cache is keyed by the full group key, evicts least recently used entries and
    can be released."""
    from gptorch_amd.models import gpr as G
    G.release_batch_buffers()
    for n in (256, 384, 512, 640, 768, 896):
        ms = _restarts(device, n, 2, [("Rbf", False, 1.0, 1.0, 0.05), ("Rbf", False, 0.8, 1.4, 0.05)])
        batched_log_likelihood(ms)
    assert len(G._BATCH_BUFFERS) <= G.BATCH_BUFFER_MAX_ENTRIES
    G.release_batch_buffers()
    assert len(G._BATCH_BUFFERS) == 0


@pytest.mark.parametrize("n,dy,batch", [(384, 1, 3), (512, 2, 2), (1536, 1, 2), (2176, 1, 2), (4096, 2, 2), (5120, 1, 2), (8192, 1, 2), (1000, 1, 2)])
def test_backward_on_poisoned_workspaces(device, n, dy, batch):
    """gpn_lml_backward / gpn_lml_backward_batched clear their U / scratch matrices only when a ragged block or K padding
    makes a contraction read what no launch wrote (n not a multiple of 128): with the workspace filled with NaN beforehand the
    gradients are finite and BIT-IDENTICAL to a run on a zeroed workspace (n = 1000: the cleared path, same check)."""
    from gptorch_amd import _native, _ops
    d = 3
    x, y = rng.make_regression(n, d, dy, seed=2)
    X, Y = torch.as_tensor(x).to(device), torch.as_tensor(y).to(device)
    var = torch.linspace(0.9, 1.3, batch, dtype=torch.float64, device=device)
    ls = torch.linspace(1.2, 2.0, batch, dtype=torch.float64, device=device)[:, None]
    nz = torch.full((batch,), 0.03, dtype=torch.float64, device=device)
    fb, _ = _ops.lml_forward_batched("Matern52", X, Y, var, ls, nz)
    assert int(fb.info.cpu().abs().max()) == 0
    lib = _native.lib()
    words = int(lib.gpn_lml_backward_batched_work_bytes(n, dy, 1, batch)) // 8
    outs = []
    for fill in (0.0, float("nan")):
        fb._backward_work = torch.full((words,), fill, dtype=torch.float64, device=device)
        g, gr = _ops.lml_backward_batched("Matern52", X, var, ls, fb, need_resid=True)
        assert bool(torch.isfinite(g).all()) and bool(torch.isfinite(gr).all()), fill
        outs.append((g.clone(), gr.clone()))
    assert torch.equal(outs[0][0], outs[1][0]) and torch.equal(outs[0][1], outs[1][1])
    # the single-model entry point on model 0's factor
    f = fb.factor(0)
    one = int(lib.gpn_lml_backward_work_bytes(n, dy, 1)) // 8
    res = []
    for fill in (0.0, float("nan")):
        work = torch.full((one,), fill, dtype=torch.float64, device=device)
        out = torch.empty(3, dtype=torch.float64, device=device)
        g_R = torch.empty(n, dy, dtype=torch.float64, device=device)
        st = lib.gpn_lml_backward(_ops._stream(device), _ops.KINDS["Matern52"], _ops._ptr(X), n, d, _ops._ptr(var[0:1]), _ops._ptr(ls[0]), 1,
                                  _ops._ptr(f.A), f.ld, _ops._ptr(f.winv), dy, _ops._ptr(work), _ops._ptr(out), _ops._ptr(g_R))
        assert st == 0 and bool(torch.isfinite(out).all())
        res.append((out, g_R))
    assert torch.equal(res[0][0], res[1][0]) and torch.equal(res[0][1], res[1][1])
    assert torch.equal(res[0][0], outs[0][0][0])


@pytest.mark.parametrize("method", ["L-BFGS-B", "CG"])
def test_multi_start_scipy_is_bit_identical_to_each_models_own_run(device, method):
    """multi_start_optimize with a scipy method (base.py:298-320; L-BFGS-B is what examples/regression_1d.py:53 runs): every
    restart's scipy.optimize.minimize runs at once and each round of evaluations is one lock-step loss + backward -- every
    restart sees bit for bit what Model._loss_and_grad (model.py:123-133) would have given it, so iterates, result and the
    number of evaluations are those of its own optimize(); restarts that finish early leave the rounds."""
    specs = [("Rbf", True, 1.0, 1.5, 0.05), ("Rbf", True, 0.5, 3.0, 0.1), ("Rbf", True, 1.8, 0.8, 0.02), ("Matern52", False, 1.0, 2.0, 0.05)]
    a = _restarts(device, 400, 3, specs)
    b = _restarts(device, 400, 3, specs)
    out = io.StringIO()
    with contextlib.redirect_stdout(out):
        res, _ = multi_start_optimize(a, method=method, max_iter=15)
    assert len(res) == 4
    evals = 0
    for i, m in enumerate(b):
        buf = io.StringIO()
        with contextlib.redirect_stdout(buf):
            ref = m.optimize(method=method, max_iter=15)
        assert np.array_equal(res[i].x, ref.x) and res[i].fun == ref.fun and res[i].nit == ref.nit and res[i].nfev == ref.nfev, (i, res[i], ref)
        for pa, pb in zip(a[i].parameters(), m.parameters()):
            assert torch.equal(pa.data, pb.data)
        evals += ref.nfev
    # every evaluation was printed exactly once ("loss: ..." as model.py:129), whichever round it ran in
    assert len([ln for ln in out.getvalue().splitlines() if ln.startswith("loss:")]) == evals
    assert len({r.nfev for r in res}) > 1              # the restarts really needed different numbers of evaluations


def test_lbfgs_golden_through_multi_start(device):
    """the reference's L-BFGS-B run of its example model (Linear + Rbf + Constant, n = 100; tests/golden/lbfgs_case.json) as ONE of
    three restarts optimised at once: final parameters and loss at the tolerances of the sequential golden test.  (A composite
    kernel: its requests take batched_loss_and_grad's sequential path inside the shared rounds.)"""
    g = load_json("lbfgs_case.json")
    x, y = np.asarray(g["x"]).reshape(-1, 1), np.asarray(g["y"]).reshape(-1, 1)
    ms = []
    for scale in (1.0, 0.6, 1.7):
        k = kernels.Linear(1) + kernels.Rbf(1, length_scales=scale) + kernels.Constant(1)
        m = GPR(x, y, k)
        m.cuda()
        ms.append(m)
    with _quiet():
        res, _ = multi_start_optimize(ms, method="L-BFGS-B", max_iter=g["max_iter"])
    assert np.max(np.abs(res[0].x - np.asarray(g["final_params"]))) < 1e-5
    assert abs(ms[0].loss().item() - g["final_loss"]) < 1e-6


def test_batched_calls_accept_any_gp_model(device):
    """models that are not plain GPR (a sparse VFE model; a GPR over a composite kernel) simply take their own loss();
    backward() inside batched_loss_and_grad / multi_start_optimize -- same numbers as calling them directly."""
    from gptorch_amd.models import VFE
    x, y = rng.make_regression(500, 2, 1, seed=3)
    def build():
        ms = _restarts(device, 500, 2, [("Rbf", False, 1.0, 1.2, 0.05), ("Rbf", False, 0.7, 0.8, 0.05)], seed=3)
        v = VFE(x, y, kernels.Matern52(2), num_inducing_points=20)
        v.cuda()
        c = GPR(x, y, kernels.Rbf(2) + kernels.Linear(2), likelihood=likelihoods.Gaussian(variance=0.05))
        c.cuda()
        return ms + [v, c]
    torch.manual_seed(0); np.random.seed(0)
    a = build()
    torch.manual_seed(0); np.random.seed(0)
    b = build()
    out = batched_loss_and_grad(a)
    for i, m in enumerate(b):
        loss = m.loss()
        loss.backward()
        assert torch.equal(out[i].reshape(-1), loss.detach().reshape(-1)), i
        for pa, pb in zip(a[i].parameters(), m.parameters()):
            assert (pa.grad is None) == (pb.grad is None)
            if pa.grad is not None:
                assert torch.equal(pa.grad, pb.grad), i


def test_batched_loss_and_grad_with_priors(device):
    """parameters with priors (model.py:158-197: loss = -(LML + log prior)): the lock-step path adds each model's own
    log_prior() to its entry -- loss and gradients bit-identical to loss(); backward(), also through a scipy multi-start."""
    def build():
        ms = _restarts(device, 600, 2, [("Rbf", False, 1.0, 1.2, 0.05), ("Rbf", False, 0.7, 0.8, 0.03), ("Rbf", False, 1.5, 2.0, 0.08)])
        g = lambda a, b: torch.distributions.Gamma(torch.tensor(a, dtype=torch.float64, device=device), torch.tensor(b, dtype=torch.float64, device=device))
        ms[0].kernel.variance.prior = g(2.0, 1.0)
        ms[0].kernel.length_scales.prior = g(3.0, 2.0)
        ms[2].likelihood.variance.prior = g(1.5, 10.0)            # model 1 has none
        return ms
    a, b = build(), build()
    out = batched_loss_and_grad(a)
    for i, m in enumerate(b):
        loss = m.loss()
        loss.backward()
        assert torch.equal(out[i], loss.detach()), (i, out[i], loss)
        for pa, pb in zip(a[i].parameters(), m.parameters()):
            assert (pa.grad is None) == (pb.grad is None)
            if pa.grad is not None:
                assert torch.equal(pa.grad, pb.grad), (i, pa.grad, pb.grad)
    assert out[0].item() != (-(a[0].log_likelihood())).item()      # the prior really is in the loss
    a, b = build(), build()
    with _quiet():
        res, _ = multi_start_optimize(a, method="L-BFGS-B", max_iter=8)
        for i, m in enumerate(b):
            ref = m.optimize(method="L-BFGS-B", max_iter=8)
            assert np.array_equal(res[i].x, ref.x) and res[i].nfev == ref.nfev


def test_lockstep_above_the_refinement_threshold(device):
    """from refine_min_n() rows on (12288) log_likelihood() refines the quadratic form (gpn_lml_refine); the lock-step path
    refines every model's factor the same way: values, losses and gradients of two restarts at N = 12288 bit-identical to
    their sequential ones, and the values carry the refinement."""
    from gptorch_amd import _ops
    n = _ops.refine_min_n()
    ms = _restarts(device, n, 6, [("Matern52", False, 1.0, 2.5, 0.01), ("Matern52", False, 0.8, 2.0, 0.02)])
    seq_v = [m.log_likelihood().detach().clone() for m in ms]
    assert all(m._holder["factor"].refined for m in ms)
    vals = batched_log_likelihood(ms)
    for a, b in zip(vals, seq_v):
        assert torch.equal(a, b)
    seq = []
    for m in ms:
        loss = m.loss()
        loss.backward()
        seq.append((loss.detach().clone(), _grads(m)))
        m.zero_grad()
    out = batched_loss_and_grad(ms)
    for i, m in enumerate(ms):
        assert torch.equal(out[i], seq[i][0])
        for ga, gb in zip(_grads(m), seq[i][1]):
            assert (ga is None) == (gb is None) and (ga is None or torch.equal(ga, gb))
    fb, terms = _ops.lml_forward_batched("Matern52", ms[0].X, ms[0].Y, torch.stack([m.kernel.variance.transform().reshape(()) for m in ms]),
                                         torch.stack([m.kernel.length_scales.transform().reshape(-1) for m in ms]),
                                         torch.stack([m.likelihood.variance.transform().reshape(()) for m in ms]), refine=False)
    assert not torch.equal(terms[0, 2:3], seq_v[0])            # unrefined differs: the refinement really ran in the batch


def _composites(device, n, d, dy, builders, seed=4):
    x, y = rng.make_regression(n, d, dy, seed=seed)
    X, Y = torch.as_tensor(x).to(device), torch.as_tensor(y).to(device)
    ms = []
    for mk, nzv in builders:
        m = GPR(X, Y, mk(), likelihood=likelihoods.Gaussian(variance=nzv))
        m.cuda()
        m.X, m.Y = X, Y
        ms.append(m)
    return ms


@pytest.mark.parametrize("n,d,dy", [(100, 1, 1), (700, 3, 2), (2176, 2, 1), (6144, 4, 1)])
def test_lockstep_composite_kernels_are_bit_identical_to_sequential(device, n, d, dy):
    """composite kernels of one structure in lock step (_expr.BatchedExprLogLik; kernels.py:286-306 Sum / Product): the
    reference's example model Linear + Rbf + Constant (examples/regression_1d.py:34-53) as three restarts, a second group of two
    Matern52 * Rbf(ARD) products, and a singleton -- the expression's assembly and sweeps per model, the factorisation, the
    reductions and Kyy^-1 / a once over each group.  Losses and gradients bit-identical to loss(); backward(); n = 6144 is
    the size from which composite kernels refine the quadratic form."""
    lin = lambda v, ell, c: (lambda: kernels.Linear(d, variance=v) + kernels.Rbf(d, length_scales=ell) + kernels.Constant(d, variance=c))
    prod = lambda v, ell: (lambda: kernels.Matern52(d, variance=v, length_scales=ell) * kernels.Rbf(d, length_scales=np.full(d, 1.5 * ell), ARD=True))
    builders = [(lin(0.5, 1.0, 0.3), 0.05), (prod(0.8, 1.2), 0.04), (lin(0.9, 0.6, 0.5), 0.03), (lin(0.2, 2.0, 1.0), 0.08), (prod(1.3, 0.7), 0.06),
                (lambda: kernels.Rbf(d) + kernels.White(d, variance=0.01), 0.05)]
    a = _composites(device, n, d, dy, builders)
    b = _composites(device, n, d, dy, builders)
    from gptorch_amd.models import gpr as G
    groups = G._expression_groups(a)
    assert sorted(len(g) for _, g, _ in groups) == [2, 3]
    out = batched_loss_and_grad(a)
    for i, m in enumerate(b):
        loss = m.loss()
        loss.backward()
        assert torch.equal(out[i], loss.detach()), (i, out[i], loss)
        for (name, pa), pb in zip(a[i].named_parameters(), m.parameters()):
            assert (pa.grad is None) == (pb.grad is None), name
            if pa.grad is not None:
                assert torch.equal(pa.grad, pb.grad), (i, name, pa.grad, pb.grad)
    vals = batched_log_likelihood(a)
    for i, m in enumerate(b):
        with torch.no_grad():
            assert torch.equal(vals[i], m.log_likelihood()), i
    if n >= 6144:
        assert b[0]._holder["factor"].refined


def test_lockstep_composite_replays_the_ladder_and_fits_with_scipy(device):
    """(1) one composite model of the group is singular at its noise level (duplicated points, noise ~ 0): it is replayed alone
    through the jitter ladder, values and gradients equal its sequential ones; (2) the reference's example model as three
    restarts of an L-BFGS-B multi-start (base.py:298-320): results bit-identical to each restart's own optimize()."""
    n, d = 300, 2
    x, y = rng.make_regression(n, d, 1, seed=9)
    xdup = np.array(x)
    xdup[150:] = xdup[:150]
    def build():
        ms = []
        for b in range(3):
            m = GPR(xdup if b == 1 else x, y, kernels.Linear(d, variance=0.3 + 0.1 * b) + kernels.Rbf(d, length_scales=1.0 + 0.2 * b) + kernels.Constant(d),
                    likelihood=likelihoods.Gaussian(variance=0.03))
            m.cuda()
            if b == 1:
                m.likelihood.variance.data.fill_(-80.0)
            ms.append(m)
        return ms
    a, b = build(), build()
    out = batched_loss_and_grad(a)
    for i, m in enumerate(b):
        loss = m.loss()
        loss.backward()
        assert torch.equal(out[i], loss.detach()), i
        for pa, pb in zip(a[i].parameters(), m.parameters()):
            assert (pa.grad is None) == (pb.grad is None) and (pa.grad is None or torch.equal(pa.grad, pb.grad)), i
    assert b[1]._holder["factor"].jitter_rung >= 0
    g = load_json("lbfgs_case.json")
    xs, ys = np.asarray(g["x"]).reshape(-1, 1), np.asarray(g["y"]).reshape(-1, 1)
    def restarts():
        ms = []
        for scale in (1.0, 0.6, 1.7):
            m = GPR(xs, ys, kernels.Linear(1) + kernels.Rbf(1, length_scales=scale) + kernels.Constant(1))
            m.cuda()
            ms.append(m)
        return ms
    a, b = restarts(), restarts()
    with _quiet():
        res, _ = multi_start_optimize(a, method="L-BFGS-B", max_iter=25)
        for i, m in enumerate(b):
            ref = m.optimize(method="L-BFGS-B", max_iter=25)
            assert np.array_equal(res[i].x, ref.x) and res[i].nfev == ref.nfev and res[i].fun == ref.fun, i


def test_multi_start_shares_the_evaluation_of_models_that_cannot_be_stacked(device):
    """composite kernels of one structure and stationary models with priors cannot share a stacked parameter tensor;
    multi_start_optimize still evaluates them together every iteration (one batched_loss_and_grad) while each keeps its own
    optimiser over its own parameters -- the very tensors of its own optimize(), so losses and final parameters are
    BIT-IDENTICAL to it (base.py:260-269)."""
    d = 2
    def build():
        lin = lambda v, ell: (lambda: kernels.Linear(d, variance=v) + kernels.Rbf(d, length_scales=ell) + kernels.Constant(d))
        ms = _composites(device, 500, d, 1, [(lin(0.5, 1.0), 0.05), (lin(0.9, 0.6), 0.03), (lin(0.2, 2.0), 0.08)])
        ps = _restarts(device, 500, d, [("Rbf", False, 1.0, 1.2, 0.05), ("Rbf", False, 0.7, 0.8, 0.03)], seed=4)
        for m in ps:
            m.kernel.variance.prior = torch.distributions.Gamma(torch.tensor(2.0, dtype=torch.float64, device=device),
                                                                 torch.tensor(1.0, dtype=torch.float64, device=device))
        return ms + ps
    a, b = build(), build()
    with _quiet():
        losses, _ = multi_start_optimize(a, method="Adam", max_iter=8)
    for i, m in enumerate(b):
        with _quiet():
            ref, _ = m.optimize(method="Adam", max_iter=8, verbose=False)
        assert np.array_equal(losses[i], ref), (i, losses[i] - ref)
        for pa, pb in zip(a[i].parameters(), m.parameters()):
            assert torch.equal(pa.data, pb.data), i

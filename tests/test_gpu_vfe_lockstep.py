"""Sparse (VFE) models in lock step (gptorch_amd/models/_vfe_lockstep.py; gptorch/models/sparse_gpr.py:108-153 for B restarts
of one shape where the reference runs one model per optimiser step, models/base.py:260-269): every model's bound and
gradients BIT-IDENTICAL to its own log_likelihood() / loss(); backward(), the reference's known answer through the lock-step
path, the jitter ladder replayed per failing model, and the lock-step forms of the single-purpose entry points against
their single-model forms."""
import contextlib
import io

import numpy as np
import pytest
import torch

from tests._util import load_json, load_npz
from gptorch_amd import _native, _ops, kernels, likelihoods, mean_functions, rng
from gptorch_amd.models import GPR, VFE, batched_log_likelihood, batched_loss_and_grad, multi_start_optimize
from gptorch_amd.models import _vfe_lockstep, gpr as gpr_mod

pytestmark = pytest.mark.gpu

KERN = {"Rbf": kernels.Rbf, "Matern52": kernels.Matern52, "Matern32": kernels.Matern32, "Exp": kernels.Exp}


def _models(B, n, m, d, dy, kind="Rbf", ard=False, shared=True, seed=0, noise=0.05):
    g = np.random.default_rng(seed)
    x, y = rng.make_regression(n, d, dy, seed=seed)
    out = []
    for b in range(B):
        if not shared:
            x, y = rng.make_regression(n, d, dy, seed=seed + 100 + b)
        z = x[g.choice(n, m, replace=False)] + 0.01 * g.standard_normal((m, d))
        ls = (0.8 + 0.4 * g.random(d)) * np.sqrt(d) if ard else float((0.8 + 0.4 * g.random()) * np.sqrt(d))
        k = KERN[kind](d, variance=float(0.7 + 0.6 * g.random()), length_scales=ls, ARD=ard)
        mdl = VFE(x, y, k, inducing_points=z, likelihood=likelihoods.Gaussian(variance=float(noise * (0.5 + g.random()))),
                  mean_function=mean_functions.Zero(dy))
        mdl.cuda()
        out.append(mdl)
    if shared:
        for mdl in out[1:]:
            mdl.X, mdl.Y = out[0].X, out[0].Y
    return out


def _grads(m):
    return [p.grad.clone() for p in (m.kernel.variance, m.kernel.length_scales, m.likelihood.variance, m.Z)]


SHAPES = [
    # B, n, m, d, dy, kind, ard, shared
    (4, 300, 20, 1, 1, "Rbf", False, True),             # recursive drivers everywhere (M <= 256), ragged n
    (5, 512, 64, 2, 1, "Matern52", False, True),
    (3, 1000, 300, 3, 2, "Matern32", True, True),       # level-parallel inversion, ragged M, ARD, two outputs
    (3, 2048, 512, 4, 1, "Rbf", False, False),          # every model its own data
    (2, 9000, 384, 2, 1, "Exp", False, True),           # two K slices of the accumulation
]


@pytest.mark.parametrize("shape", SHAPES, ids=lambda s: "B%d_n%d_m%d_%s" % (s[0], s[1], s[2], s[5]))
def test_lockstep_vfe_is_bit_identical_to_sequential(device, shape):
    B, n, m, d, dy, kind, ard, shared = shape
    ms = _models(B, n, m, d, dy, kind, ard, shared)
    assert len(gpr_mod._vfe_groups(ms)) == 1
    seq_ll = [mdl.log_likelihood().detach().clone() for mdl in ms]
    got_ll = batched_log_likelihood(ms)
    for a, b in zip(seq_ll, got_ll):
        assert a.shape == b.shape and torch.equal(a, b), (a.item(), b.item())
    seq = []
    for mdl in ms:
        mdl.zero_grad()
        loss = mdl.loss()
        loss.backward()
        seq.append((loss.detach().clone(), _grads(mdl)))
        mdl.zero_grad()
    losses = batched_loss_and_grad(ms)
    for mdl, (l0, g0), l1 in zip(ms, seq, losses):
        assert l0.shape == l1.shape and torch.equal(l0, l1), (l0.item(), l1.item())
        for a, b in zip(g0, _grads(mdl)):
            assert a.shape == b.shape and torch.equal(a, b), (a - b).abs().max().item()
        assert all(torch.isfinite(g).all() for g in g0)


def test_reference_known_answer_through_the_lockstep_path(device):
    """test/test_models/test_sparse_gpr.py:81-142's pinned loss, as one of three restarts of a lock-step group"""
    z = load_npz("ref_sparse_gpr_fixtures.npz")
    ms = []
    for i in range(3):
        mdl = VFE(z["x"], z["y"], kernels.Matern32(1, variance=1.0 + 0.3 * i), inducing_points=z["z"] + 0.02 * i,
                  likelihood=likelihoods.Gaussian(variance=1.0), mean_function=mean_functions.Zero(1))
        mdl.cuda()
        ms.append(mdl)
    assert len(gpr_mod._vfe_groups(ms)) == 1
    losses = batched_loss_and_grad(ms)
    assert losses[0].ndimension() == 0 and losses[0].is_cuda
    assert losses[0].item() == pytest.approx(8.842242323920674)
    assert abs(losses[0].item() - float(z["vfe_loss_reference_run"][0])) < 1e-9
    case = load_json("vfe_cases.json")[0]
    # and the reference's gradients (autograd through sparse_gpr.py:108-153) for a medium case, second of two restarts
    from tests.test_gpu_parity import _vfe_case_model
    a, b = _vfe_case_model(case), _vfe_case_model(case)
    a.kernel.variance.data += 0.1
    for mdl in (a, b):
        mdl.zero_grad()
    out = batched_loss_and_grad([a, b])
    assert abs(-out[1].item() - case["elbo"]) < 1e-8 * abs(case["elbo"])
    got = [b.kernel.variance.grad.cpu().numpy().ravel(), b.kernel.length_scales.grad.cpu().numpy().ravel(),
           b.likelihood.variance.grad.cpu().numpy().ravel(), b.Z.grad.cpu().numpy()]
    for g, r in zip(got, [np.asarray(case[k]) for k in ("g_variance", "g_length_scales", "g_noise", "g_Z")]):
        assert np.abs(g.reshape(r.shape) - r).max() < 1e-7 * np.abs(r).max()


def test_lockstep_vfe_climbs_the_ladder_per_failing_model(device):
    """one model of the group has a K(Z) that is numerically singular (a length scale far beyond the data's extent:
    functions.py:20-43's ladder adds jitter): the ladder runs inside the lock-step evaluation (a sub-batch of the failing
    factorisations, rung by rung) -- every model bit-identical to its sequential evaluation"""
    ms = _models(4, 600, 48, 2, 1, "Rbf")
    bad = VFE(ms[0].X, ms[0].Y, kernels.Rbf(2, variance=1.0, length_scales=400.0), inducing_points=ms[2].Z.data.cpu().numpy(),
              likelihood=likelihoods.Gaussian(variance=0.05), mean_function=mean_functions.Zero(1))
    bad.cuda()
    bad.X, bad.Y = ms[0].X, ms[0].Y
    ms[2] = bad
    assert len(gpr_mod._vfe_groups(ms)) == 1 and len(gpr_mod._vfe_groups(ms)[0][1]) == 4
    seq = []
    for mdl in ms:
        mdl.zero_grad()
        loss = mdl.loss()
        loss.backward()
        seq.append((loss.detach().clone(), _grads(mdl)))
        mdl.zero_grad()
    assert ms[2]._bound(ms[2].X)[1].f_uu.jitter_rung >= 0          # the sequential evaluation did climb the ladder
    c0 = _vfe_lockstep.LADDER_CLIMBS
    losses = batched_loss_and_grad(ms)
    assert _vfe_lockstep.LADDER_CLIMBS > c0
    for mdl, (l0, g0), l1 in zip(ms, seq, losses):
        assert torch.equal(l0, l1)
        for a, b in zip(g0, _grads(mdl)):
            assert torch.equal(a, b)


def test_lockstep_vfe_with_priors_and_frozen_parameters(device):
    """parameters with priors (model.py:158-197: loss = -(bound + log prior), each model's own), frozen inducing points in one model
    and a frozen kernel variance in another: loss and every gradient that exists bit-identical to loss(); backward(); a scipy
    multi-start over the same group returns each model's own result"""
    g = lambda a, b: torch.distributions.Gamma(torch.tensor(a, dtype=torch.float64, device=device), torch.tensor(b, dtype=torch.float64, device=device))

    def fresh():
        ms = _models(4, 450, 36, 2, 1, "Matern52", seed=11)
        ms[0].kernel.variance.prior = g(2.0, 1.0)
        ms[2].likelihood.variance.prior = g(1.5, 10.0)
        ms[1].Z.requires_grad_(False)
        ms[3].kernel.variance.requires_grad_(False)
        return ms
    a, b = fresh(), fresh()
    assert len(gpr_mod._vfe_groups(b)) == 1
    for mdl in a:
        mdl.zero_grad()
        mdl.loss().backward()
    out = batched_loss_and_grad(b)
    for ma, mb, l1 in zip(a, b, out):
        assert torch.equal(ma.loss().detach(), l1)
        for p, q in zip(ma.parameters(), mb.parameters()):
            assert (p.grad is None) == (q.grad is None)
            if p.grad is not None:
                assert torch.equal(p.grad, q.grad)
    assert b[1].Z.grad is None and b[3].kernel.variance.grad is None
    assert out[0].item() != (-(b[0].log_likelihood())).item()        # the prior really is in the loss
    c, d = fresh(), fresh()
    with contextlib.redirect_stdout(io.StringIO()):
        own = [mdl.optimize(method="L-BFGS-B", max_iter=6) for mdl in c]
        res, _ = multi_start_optimize(d, method="L-BFGS-B", max_iter=6)
    for r0, r1 in zip(own, res):
        assert np.array_equal(r0.x, r1.x) and r0.fun == r1.fun and r0.nfev == r1.nfev


def test_mixed_models_take_their_own_paths(device):
    """VFE groups, a GPR group and a singleton in one call: every loss and gradient as from the model's own loss(); backward()"""
    v1 = _models(2, 400, 32, 2, 1, "Rbf", seed=1)
    v2 = _models(2, 400, 40, 2, 1, "Rbf", seed=2)                  # another M: another group
    x, y = rng.make_regression(256, 2, 1, seed=3)
    g = [GPR(x, y, kernels.Rbf(2, variance=1.0 + 0.1 * i)) for i in range(2)]
    for mdl in g:
        mdl.cuda()
    lone = _models(1, 300, 24, 2, 1, "Matern52", seed=4)
    ms = [v1[0], g[0], v2[0], lone[0], v1[1], g[1], v2[1]]
    assert len(gpr_mod._vfe_groups(ms)) == 2
    seq = []
    for mdl in ms:
        mdl.zero_grad()
        loss = mdl.loss()
        loss.backward()
        seq.append((loss.detach().clone(), [p.grad.clone() for p in mdl.parameters() if p.grad is not None]))
        mdl.zero_grad()
    losses = batched_loss_and_grad(ms)
    for mdl, (l0, g0), l1 in zip(ms, seq, losses):
        assert torch.equal(l0.reshape(-1), l1.reshape(-1))
        for a, b in zip(g0, [p.grad for p in mdl.parameters() if p.grad is not None]):
            assert torch.equal(a, b)


def test_multi_start_fit_of_sparse_models_is_bit_identical(device):
    """multi_start_optimize over VFE restarts: one optimiser per model, one lock-step evaluation per iteration -- the
    trajectory of every restart equals its own optimize() (base.py:260-269) bit for bit"""
    def fresh():
        return _models(3, 500, 40, 2, 1, "Matern52", seed=7)
    a, b = fresh(), fresh()
    with contextlib.redirect_stdout(io.StringIO()):
        own = [mdl.optimize(method="Adam", max_iter=12, learning_rate=0.05)[0] for mdl in a]
        losses, _ = multi_start_optimize(b, method="Adam", max_iter=12, learning_rate=0.05)
    for i in range(3):
        assert np.array_equal(np.asarray(own[i]), losses[i]), (own[i], losses[i])
        for p, q in zip(a[i].parameters(), b[i].parameters()):
            assert torch.equal(p.data, q.data)


# ---- the lock-step forms of the single-purpose entry points ------------------------------------------------------------
def _dev(a, device):
    return torch.as_tensor(np.ascontiguousarray(a), dtype=torch.float64).to(device)


@pytest.mark.parametrize("n,m,d,nls", [(200, 70, 3, 1), (130, 300, 5, 5), (64, 64, 70, 1)])
def test_kernel_matrix_and_sweeps_batched_equal_single(device, n, m, d, nls):
    lib = _native.lib()
    g = np.random.default_rng(0)
    B = 3
    X, Xs = _dev(g.standard_normal((n, d)), device), _dev(g.standard_normal((B, n, d)), device)
    Z = _dev(g.standard_normal((B, m, d)), device)
    var, ls = _dev(0.5 + g.random(B), device), _dev(1.0 + g.random((B, nls)), device)
    G = _dev(g.standard_normal((B, n, m + 6)), device)
    s = _ops._stream(device)
    for kind in ("Rbf", "Matern52"):
        k = _ops.KINDS[kind]
        for Xp, sX in ((X, 0), (Xs, n * d)):
            K = torch.full((B, n + 2, m + 4), float("nan"), dtype=torch.float64, device=device)
            rc = lib.gpn_kernel_matrix_batched(s, k, B, _ops._ptr(Xp), sX, n, _ops._ptr(Z), m * d, m, d, _ops._ptr(var), _ops._ptr(ls), nls,
                                               None, 0, _ops._ptr(K), m + 4, (n + 2) * (m + 4))
            assert rc == 0, rc
            w1 = torch.empty(B * int(lib.gpn_grad_work_bytes(n, m, nls, 0)) // 8, dtype=torch.float64, device=device)
            o1 = torch.empty(B, 1 + nls, dtype=torch.float64, device=device)
            rc = lib.gpn_kernel_grad_batched(s, k, B, _ops._ptr(Xp), sX, n, _ops._ptr(Z), m * d, m, d, _ops._ptr(var), _ops._ptr(ls), nls,
                                             _ops._ptr(G), m + 6, n * (m + 6), _ops._ptr(w1), _ops._ptr(o1))
            assert rc == 0, rc
            w2 = torch.empty(max(1, B * int(lib.gpn_grad_x2_work_bytes(n, m, d)) // 8), dtype=torch.float64, device=device)
            o2 = torch.ones(B, m, d, dtype=torch.float64, device=device)
            rc = lib.gpn_kernel_grad_x2_batched(s, k, B, _ops._ptr(Xp), sX, n, _ops._ptr(Z), m * d, m, d, _ops._ptr(var), _ops._ptr(ls), nls,
                                                _ops._ptr(G), m + 6, n * (m + 6), 0.5, 1, _ops._ptr(w2), _ops._ptr(o2))
            assert rc == 0, rc
            for b in range(B):
                xb = Xp if sX == 0 else Xp[b]
                one = _ops.kernel_matrix(kind, xb, Z[b], var[b:b + 1], ls[b])
                assert torch.equal(K[b, :n, :m], one)
                from gptorch_amd import _backward
                gv, gl = _backward.kernel_backward(kind, xb, Z[b], var[b:b + 1], ls[b], G[b, :, :m])
                assert torch.equal(o1[b, 0:1], gv) and torch.equal(o1[b, 1:], gl)
                acc = torch.ones(m, d, dtype=torch.float64, device=device)
                _backward.kernel_backward_x2(kind, xb, Z[b], var[b:b + 1], ls[b], G[b, :, :m], scale=0.5, out=acc)
                assert torch.equal(o2[b], acc)
        # symmetric, lower tiles, with a diagonal term
        nz = _dev(0.1 + g.random(B), device)
        K = torch.zeros(B, m, m, dtype=torch.float64, device=device)
        rc = lib.gpn_kernel_matrix_batched(s, k, B, _ops._ptr(Z), m * d, m, None, 0, m, d, _ops._ptr(var), _ops._ptr(ls), nls, _ops._ptr(nz), 1,
                                           _ops._ptr(K), m, m * m)
        assert rc == 0, rc
        for b in range(B):
            one = torch.zeros(m, m, dtype=torch.float64, device=device)
            _ops.kernel_matrix(kind, Z[b], None, var[b:b + 1], ls[b], noise=nz[b:b + 1], out=one, ldk=m, lower=True)
            assert torch.equal(torch.tril(K[b]), torch.tril(one))


@pytest.mark.parametrize("n,rows", [(100, 37), (256, 300), (700, 129), (1536, 2000)])
def test_right_solve_inverse_and_scaled_contraction_batched_equal_single(device, n, rows):
    lib = _native.lib()
    B = 3
    g = np.random.default_rng(1)
    s = _ops._stream(device)
    fb = _ops.FactorBatch(B, n, 0, device)
    A3 = fb.A.view(B, fb.rows, fb.ld)
    for b in range(B):
        x = _dev(g.standard_normal((n, 3)), device)
        _ops.kernel_matrix("Rbf", x, None, _dev([1.0 + b], device), _dev([1.5], device), noise=_dev([0.1], device), out=A3[b], ldk=fb.ld,
                           lower=True)
    assert lib.gpn_potrf_lower_batched(s, _ops._ptr(fb.A), n, 0, fb.ld, fb.sA, _ops._ptr(fb.winv), fb.sW, _ops._ptr(fb.info), B) == 0
    assert fb.info.tolist() == [0] * B
    rp = _ops.round_up(rows, 128) + 16
    Bm = torch.zeros(B, rp, fb.ld, dtype=torch.float64, device=device)
    Bm[:, :rows, :n] = _dev(g.standard_normal((B, rows, n)), device)
    ref = Bm.clone()
    rc = lib.gpn_trsm_right_lt_batched(s, _ops._ptr(fb.A), n, fb.ld, fb.sA, _ops._ptr(fb.winv), fb.sW, _ops._ptr(Bm), rows, fb.ld, rp * fb.ld, B)
    assert rc == 0, rc
    U = torch.zeros(B, fb.rows, fb.ld, dtype=torch.float64, device=device)
    S = torch.zeros_like(U)
    rc = lib.gpn_trtri_upper_batched(s, _ops._ptr(fb.A), n, fb.ld, fb.sA, _ops._ptr(fb.winv), fb.sW, _ops._ptr(U), fb.ld, fb.rows * fb.ld,
                                     _ops._ptr(S), fb.ld, fb.rows * fb.ld, B)
    assert rc == 0, rc
    from gptorch_amd import _backward
    for b in range(B):
        f = fb.factor(b)
        f.solve_right_lt(ref[b], rows)
        assert torch.equal(ref[b], Bm[b])
        assert torch.equal(_backward._upper_inverse(f), U[b])
    # one scale per problem, from device memory
    kp = _ops.round_up(n, 16)
    al = _dev(0.3 + g.random(B), device)
    C = torch.full((B, rows, rows), float("nan"), dtype=torch.float64, device=device)
    rc = lib.gpn_gemm_nt_batched_scaled(s, rows, rows, kp, _ops._ptr(al), _ops._ptr(Bm), fb.ld, rp * fb.ld, _ops._ptr(Bm), fb.ld, rp * fb.ld,
                                        0.0, _ops._ptr(C), rows, rows * rows, 1, 0, B)
    assert rc == 0, rc
    for b in range(B):
        one = _ops.gemm_nt(Bm[b], Bm[b], rows, rows, kp, alpha=float(al[b].item()), lower=True)
        assert torch.equal(torch.tril(C[b]), torch.tril(one))


@pytest.mark.parametrize("seed", [101, 102])
def test_fuzz_lockstep_vfe(device, seed):
    """random shapes on and around the blocking edges (16-row K padding, the 128 leaf, the 256-row recursion switch, the 8192-column
    accumulation slice), all stationary kinds, ARD / isotropic, dy 1..3, shared / own data: bounds and gradients bit for bit"""
    g = np.random.default_rng(seed)
    for _rep in range(6):
        n = int(g.choice([65, 128, 300, 511, 1024, 1500, 2049, 8200]))
        m = int(min(n - 1, g.choice([1, 15, 16, 64, 127, 128, 129, 256, 257, 400])))
        d, dy = int(g.choice([1, 2, 5, 9])), int(g.choice([1, 2, 3]))
        kind = str(g.choice(["Rbf", "Matern52", "Matern32", "Exp"]))
        ard, shared, B = bool(g.integers(2)) and d > 1, bool(g.integers(2)), int(g.choice([2, 3, 5]))
        ms = _models(B, n, m, d, dy, kind, ard, shared, seed=int(g.integers(1 << 30)), noise=float(g.choice([0.01, 0.1])))
        what = (n, m, d, dy, kind, ard, shared, B)
        assert len(gpr_mod._vfe_groups(ms)) == 1, what
        seq = []
        for mdl in ms:
            mdl.zero_grad()
            loss = mdl.loss()
            loss.backward()
            seq.append((loss.detach().clone(), _grads(mdl)))
            mdl.zero_grad()
        for mdl, (l0, g0), l1 in zip(ms, seq, batched_loss_and_grad(ms)):
            assert torch.equal(l0, l1), what
            for a, b in zip(g0, _grads(mdl)):
                assert torch.equal(a, b), what


def test_vfe_prediction_state_is_kept_between_predictions(device):
    """the reference re-evaluates the bound inside every _predict (sparse_gpr.py:155-170); the state it needs is kept here while data
    and parameters are unchanged: same predictions bit for bit, the bound evaluated once; any parameter or data edit rebuilds it"""
    from gptorch_amd.models import sparse_gpr
    mdl = _models(1, 700, 48, 2, 1, "Matern52", seed=21)[0]
    xs = torch.as_tensor(rng.normal(5, (33, 2))).to(device)
    calls = []
    orig = sparse_gpr._vfe_forward
    sparse_gpr._vfe_forward = lambda *a, **k: (calls.append(1), orig(*a, **k))[1]
    try:
        m0, v0 = mdl.predict_y(xs)
        m1, c1 = mdl.predict_y(xs, diag=False)
        m2, v2 = mdl.predict_f(xs)
        assert len(calls) == 1
        assert torch.equal(m0, m1) and torch.allclose(torch.diagonal(c1).reshape(-1), v0.reshape(-1), rtol=1e-9, atol=1e-12)
        with torch.no_grad():
            mdl.Z.data[3] += 0.05                                  # an inducing point moves: the state is rebuilt
        m3, v3 = mdl.predict_y(xs)
        assert len(calls) == 2 and not torch.equal(m3, m0)
        with torch.no_grad():
            mdl.Z.data[3] -= 0.05
        mdl._predict_cache = None
        m4, v4 = mdl.predict_y(xs)
        assert len(calls) == 3
        fresh = _models(1, 700, 48, 2, 1, "Matern52", seed=21)[0]
        m5, v5 = fresh.predict_y(xs)
        assert torch.equal(m5, m0) and torch.equal(v5, v0)         # (what a model that never cached anything computes)
        mdl.Y = mdl.Y.clone()                                      # other data: rebuilt
        mdl.predict_y(xs)
        assert len(calls) == 5
    finally:
        sparse_gpr._vfe_forward = orig

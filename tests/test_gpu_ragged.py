"""Lock-step batches of models of DIFFERENT sizes (gpn_lml_forward_ragged / gpn_lml_backward_ragged; the reference evaluates one
model at a time, gptorch/models/base.py:260-269; cross-validation folds of unequal length, learning curves): every model padded to
the group's largest N with identity rows -- factor, alpha, LML terms and gradients BIT-IDENTICAL to the model's own sequential
evaluation (gpr.py:47-67 and its closed-form backward)."""
import numpy as np
import pytest
import torch

from gptorch_amd import _backward, _ops, kernels, likelihoods, rng
from gptorch_amd.models import GPR, batched_log_likelihood, batched_loss_and_grad

pytestmark = pytest.mark.gpu


def _case(device, sizes, d, dy, kind, ard, seed=0):
    g = np.random.default_rng(seed)
    B, nmax = len(sizes), max(sizes)
    X = torch.full((B, nmax, d), float("nan"), dtype=torch.float64, device=device)      # the padding must never reach a result
    Y = torch.full((B, nmax, dy), float("nan"), dtype=torch.float64, device=device)
    for b, n in enumerate(sizes):
        x, y = rng.make_regression(n, d, dy, seed=seed + b)
        X[b, :n], Y[b, :n] = torch.as_tensor(x).to(device), torch.as_tensor(y).to(device)
    nls = d if ard else 1
    var = torch.as_tensor(0.7 + 0.6 * g.random(B)).to(device)
    ls = torch.as_tensor((0.8 + 0.4 * g.random((B, nls))) * np.sqrt(d)).to(device)
    nz = torch.as_tensor(0.02 + 0.05 * g.random(B)).to(device)
    n_of = torch.tensor(sizes, dtype=torch.int32, device=device)
    return X, Y, var, ls, nz, n_of


CASES = [
    ((700, 1000, 1024, 257), 2, 1, "Rbf", False),                 # one panel level (N <= 2048), ragged blocks, a full one
    ((2049, 2300, 3000, 4096, 3333), 3, 1, "Matern52", False),     # two panel levels; the largest a power of two
    ((5000, 6144, 5121), 4, 2, "Rbf", True),                       # ARD, two right-hand sides, N_max not a power of two
    ((8192, 7168, 8000), 8, 1, "Matern32", False),
]


@pytest.mark.parametrize("case", CASES, ids=lambda c: "n%s_%s" % ("-".join(map(str, c[0])), c[3]))
def test_ragged_batch_is_bit_identical_to_each_models_own_evaluation(device, case):
    sizes, d, dy, kind, ard = case
    X, Y, var, ls, nz, n_of = _case(device, sizes, d, dy, kind, ard)
    fb, terms = _ops.lml_forward_batched(kind, X, Y, var, ls, nz, n_of=n_of)
    assert fb.info.tolist() == [0] * len(sizes)
    grads, _ = _ops.lml_backward_batched(kind, X, var, ls, fb, n_of=n_of)
    nmax = max(sizes)
    A3 = fb.A.view(len(sizes), fb.rows, fb.ld)
    for b, n in enumerate(sizes):
        f, t = _ops.lml_forward(kind, X[b, :n], Y[b, :n], var[b:b + 1], ls[b], nz[b:b + 1], refine=False)
        assert torch.equal(t, terms[b]), (b, n, t.tolist(), terms[b].tolist())
        assert torch.equal(torch.tril(f.A[:n, :n]), torch.tril(A3[b, :n, :n]))
        assert torch.equal(f.A[n:n + dy, :n], A3[b, nmax:nmax + dy, :n])
        # the identity block: factors to itself, zero coupling, zero right-hand sides
        if n < nmax:
            assert torch.equal(torch.tril(A3[b, n:nmax, n:nmax]), torch.eye(nmax - n, dtype=torch.float64, device=device))
            assert not A3[b, n:nmax, :n].any() and not A3[b, nmax:nmax + dy, n:nmax].any()
        gv, gl, gn, _ = _backward.lml_backward(kind, X[b, :n], var[b:b + 1], ls[b], nz[b:b + 1], f)
        assert torch.equal(grads[b], torch.cat([gv, gl, gn])), (b, n, grads[b].tolist(), torch.cat([gv, gl, gn]).tolist())


# ---- through the shell -------------------------------------------------------------------------------------------------
import contextlib
import io

from gptorch_amd.models import gpr as gpr_mod, multi_start_optimize


def _gprs(sizes, d=3, kind=kernels.Matern52, seed=0, noise=0.05, ls=None):
    ms = []
    for b, n in enumerate(sizes):
        x, y = rng.make_regression(n, d, 1, seed=seed + b)
        m = GPR(x, y, kind(d, variance=0.8 + 0.1 * b, length_scales=(1.0 + 0.1 * b) * np.sqrt(d) if ls is None else ls[b]),
                likelihood=likelihoods.Gaussian(variance=noise if np.isscalar(noise) else noise[b]))
        m.cuda()
        ms.append(m)
    return ms


def _own(ms):
    out = []
    for m in ms:
        m.zero_grad()
        loss = m.loss()
        loss.backward()
        out.append((loss.detach().clone(), [p.grad.clone() for p in m.parameters() if p.grad is not None]))
        m.zero_grad()
    return out


def test_models_of_unequal_size_share_one_lockstep_evaluation(device):
    """cross-validation folds of unequal length / a learning curve: GPR models of one kernel and different N in ONE ragged group --
    log_likelihood(), loss() and every gradient bit-identical to the model's own"""
    sizes = (1500, 1400, 1333, 1201, 1500 - 1)
    ms = _gprs(sizes)
    groups = gpr_mod._lockstep_groups(ms)
    assert len(groups) == 1 and len(groups[0][0]) == 6 and groups[0][0][5] == sizes and groups[0][0][1][0] == 1500
    seq_ll = [m.log_likelihood().detach().clone() for m in ms]
    for a, b in zip(seq_ll, batched_log_likelihood(ms)):
        assert torch.equal(a.reshape(-1), b.reshape(-1))
    own = _own(ms)
    losses = batched_loss_and_grad(ms)
    for m, (l0, g0), l1 in zip(ms, own, losses):
        assert torch.equal(l0.reshape(-1), l1.reshape(-1))
        for a, b in zip(g0, [p.grad for p in m.parameters() if p.grad is not None]):
            assert torch.equal(a, b)


def test_ragged_groups_next_to_equal_groups_and_outliers(device):
    """eight models of one size (an ordinary lock-step group), two of another size with three of nearby sizes (small equal-size groups
    are pooled with their neighbours: ONE ragged group of five), one far smaller and one on the other side of the 2048-row panel
    regime (their own evaluations): every result as from the model's own loss(); backward()"""
    ms = _gprs((1900,) * 8 + (1800, 1800, 1700, 1650, 1601, 600, 2100))
    groups = gpr_mod._lockstep_groups(ms)
    assert sorted(len(g) for _, g in groups) == [5, 8]
    assert [k[5] for k, g in groups if len(k) > 5] == [(1800, 1800, 1700, 1650, 1601)]
    own = _own(ms)
    losses = batched_loss_and_grad(ms)
    for m, (l0, g0), l1 in zip(ms, own, losses):
        assert torch.equal(l0.reshape(-1), l1.reshape(-1))
        for a, b in zip(g0, [p.grad for p in m.parameters() if p.grad is not None]):
            assert torch.equal(a, b)
    # folds of n and n - 1 rows: one group of all of them
    ms = _gprs((1000, 999, 1000, 999, 1000, 999))
    groups = gpr_mod._lockstep_groups(ms)
    assert len(groups) == 1 and groups[0][0][5] == (1000, 999, 1000, 999, 1000, 999)
    own = _own(ms)
    for m, (l0, g0), l1 in zip(ms, own, batched_loss_and_grad(ms)):
        assert torch.equal(l0.reshape(-1), l1.reshape(-1))
        for a, b in zip(g0, [p.grad for p in m.parameters() if p.grad is not None]):
            assert torch.equal(a, b)


def test_ragged_group_replays_the_ladder_and_fits_bitwise(device):
    """a member that needs the jitter ladder (functions.py:20-43) is replayed alone on its own points; a multi-start Adam fit over a
    ragged group follows every model's own optimize() bit for bit"""
    ms = _gprs((900, 850, 800), d=2, kind=kernels.Rbf, noise=(0.05, 1e-16, 0.05), ls=(1.5, 3000.0, 1.2))
    ms[1].loss()
    if ms[1]._holder["factor"].jitter_rung < 0:
        pytest.skip("this case does not need the ladder any more")
    assert len(gpr_mod._lockstep_groups(ms)) == 1
    own = _own(ms)
    losses = batched_loss_and_grad(ms)
    for m, (l0, g0), l1 in zip(ms, own, losses):
        assert torch.equal(l0.reshape(-1), l1.reshape(-1))
        for a, b in zip(g0, [p.grad for p in m.parameters() if p.grad is not None]):
            assert torch.equal(a, b)
    a, b = _gprs((2500, 2400, 2222)), _gprs((2500, 2400, 2222))
    with contextlib.redirect_stdout(io.StringIO()):
        mine = [m.optimize(method="Adam", max_iter=8, learning_rate=0.02)[0] for m in a]
        losses, _ = multi_start_optimize(b, method="Adam", max_iter=8, learning_rate=0.02)
    for i in range(3):
        assert np.array_equal(np.asarray(mine[i]), losses[i])
        for p, q in zip(a[i].parameters(), b[i].parameters()):
            assert torch.equal(p.data, q.data)


@pytest.mark.parametrize("seed", [201, 202])
def test_fuzz_ragged_groups(device, seed):
    """random size sets inside both panel regimes (ragged and full last blocks, sizes one apart, the regime's edges), all stationary
    kinds, ARD / isotropic, dy 1..3: terms, factors and gradients of every member bit for bit"""
    g = np.random.default_rng(seed)
    for _rep in range(5):
        lo, hi = (257, 2048) if g.integers(2) else (2049, 6000)
        nmax = int(g.choice([hi, int(g.integers(lo + 64, hi)), (int(g.integers(lo + 128, hi)) // 128) * 128]))
        sizes = [nmax] + [int(max(lo, nmax - g.integers(0, max(2, nmax // 4)))) for _ in range(int(g.integers(1, 4)))]
        if g.integers(2):
            sizes.append(nmax - 1)
        sizes = tuple(int(v) for v in g.permutation(sizes))
        d, dy = int(g.choice([1, 3, 8, 17])), int(g.choice([1, 2, 3]))
        kind = str(g.choice(["Rbf", "Matern52", "Matern32", "Exp"]))
        ard = bool(g.integers(2)) and d > 1
        X, Y, var, ls, nz, n_of = _case(device, sizes, d, dy, kind, ard, seed=int(g.integers(1 << 30)))
        fb, terms = _ops.lml_forward_batched(kind, X, Y, var, ls, nz, n_of=n_of)
        assert fb.info.tolist() == [0] * len(sizes), (sizes, kind)
        grads, _ = _ops.lml_backward_batched(kind, X, var, ls, fb, n_of=n_of)
        A3 = fb.A.view(len(sizes), fb.rows, fb.ld)
        for b, n in enumerate(sizes):
            f, t = _ops.lml_forward(kind, X[b, :n], Y[b, :n], var[b:b + 1], ls[b], nz[b:b + 1], refine=False)
            what = (sizes, b, d, dy, kind, ard)
            assert torch.equal(t, terms[b]), what
            assert torch.equal(torch.tril(f.A[:n, :n]), torch.tril(A3[b, :n, :n])), what
            gv, gl, gn, _ = _backward.lml_backward(kind, X[b, :n], var[b:b + 1], ls[b], nz[b:b + 1], f)
            assert torch.equal(grads[b], torch.cat([gv, gl, gn])), what


def test_batched_factorise_seeds_bit_identical_predictions(device):
    """cross-validation scoring: the factorisations of k fitted folds in lock step (gpr.py:104-106 for all of them at once), then
    every model's own predict_f / predict_y on its held-out rows -- bit-identical to the model predicting alone, factor not recomputed"""
    from gptorch_amd.models import batched_factorise
    a, b = _gprs((1300,) * 4, d=3), _gprs((1300,) * 4, d=3)
    xs = [torch.as_tensor(rng.normal(50 + i, (37, 3))).to(device) for i in range(4)]
    assert batched_factorise(b) == 4
    for ma, mb, x in zip(a, b, xs):
        f_seeded = mb._predict_cache[1]
        for diag in (True, False):
            m0, v0 = ma.predict_y(x, diag=diag)
            m1, v1 = mb.predict_y(x, diag=diag)
            assert torch.equal(m0, m1) and torch.equal(v0, v1)
        assert mb._predict_cache[1] is f_seeded                       # the seeded factor served both predictions
        mf0, vf0 = ma.predict_f(x)
        mf1, vf1 = mb.predict_f(x)
        assert torch.equal(mf0, mf1) and torch.equal(vf0, vf1)
    # a parameter edit invalidates the seeded factor like any cached one
    with torch.no_grad():
        b[0].kernel.variance.data += 0.1
        a[0].kernel.variance.data += 0.1
    m0, v0 = a[0].predict_y(xs[0])
    m1, v1 = b[0].predict_y(xs[0])
    assert torch.equal(m0, m1) and torch.equal(v0, v1)


def test_cross_validation_example_runs(device):
    """examples/cross_validation.py: five folds of unequal length fitted and scored in lock step"""
    import os
    import subprocess
    import sys
    root = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
    out = subprocess.run([sys.executable, os.path.join(root, "examples", "cross_validation.py"), "1503", "5", "6"], cwd=root,
                         capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, (out.stdout + out.stderr)[-2000:]
    assert "multi_start_optimize:" in out.stdout and "held-out RMSE per fold" in out.stdout


def test_composite_kernel_predictions_keep_their_factor(device):
    """GPR over a composite kernel (the reference's example model, examples/regression_1d.py:34-53) predicts through the dense path:
    its factor is kept between predictions like the native kinds' (the reference re-factorises inside every _predict, gpr.py:104-106)"""
    x, y = rng.make_regression(400, 2, 1, seed=8)
    m = GPR(x, y, kernels.Linear(2) + kernels.Rbf(2) + kernels.Constant(2), likelihood=likelihoods.Gaussian(variance=0.05))
    m.cuda()
    xs = torch.as_tensor(rng.normal(9, (21, 2))).to(device)
    calls = []
    orig = _ops.cholesky_factor
    _ops.cholesky_factor = lambda *a, **k: (calls.append(1), orig(*a, **k))[1]
    try:
        m0, v0 = m.predict_y(xs)
        m1, v1 = m.predict_y(xs)
        _, c1 = m.predict_f(xs, diag=False)
        assert len(calls) == 1 and torch.equal(m0, m1) and torch.equal(v0, v1)
        with torch.no_grad():
            m.likelihood.variance.data += 0.01
        m2, _ = m.predict_y(xs)
        assert len(calls) == 2 and not torch.equal(m2, m0)
    finally:
        _ops.cholesky_factor = orig
    fresh = GPR(x, y, kernels.Linear(2) + kernels.Rbf(2) + kernels.Constant(2), likelihood=likelihoods.Gaussian(variance=0.05))
    fresh.cuda()
    m3, v3 = fresh.predict_y(xs)
    assert torch.equal(m3, m0) and torch.equal(v3, v0)

"""Round 6 on the GPU: the persistent factorisation (csrc/ppotrf.hip) against the launch-based driver -- bit for bit --, the
optimiser step as one hipGraph replay (GPModel.optimize(capture=True); base.py:260-269), functions.inverse (functions.py:57-58),
and the bitwise default of multi_start_optimize from N = 2048."""
import contextlib
import io

import numpy as np
import pytest
import torch

from tests._util import load_json
from gptorch_amd import _native, _ops, functions, kernels, likelihoods, rng
from gptorch_amd.models import GPR, multi_start_optimize
from oracle import gp_oracle as orc

pytestmark = pytest.mark.gpu

KERN = {"Rbf": kernels.Rbf, "Matern52": kernels.Matern52}


def _factor_input(n, dy, device, d=6, noise=1e-2, seed=0):
    x, y = rng.make_regression(n, d, dy, seed=seed)
    X, Y = torch.as_tensor(x).to(device), torch.as_tensor(y).to(device)
    one = lambda v: torch.tensor([v], dtype=torch.float64, device=device)
    f = _ops.Factor(n, dy, device)
    _ops.kernel_matrix("Rbf", X, None, one(1.0), one(float(np.sqrt(d))), noise=one(noise), out=f.A, ldk=f.ld, lower=True)
    if dy:
        f.pack_rhs(Y)
    return f


def _run(f, saved, persistent):
    lib = _native.lib()
    f.A.copy_(saved)
    f.info.zero_()
    fn = lib.gpn_potrf_lower_persistent if persistent else lib.gpn_potrf_lower
    rc = fn(_ops._stream(f.device), _ops._ptr(f.A), f.n, f.e, f.ld, _ops._ptr(f.winv), _ops._ptr(f.info))
    assert rc == 0, rc
    torch.cuda.synchronize()
    return torch.tril(f.A[:f.n, :f.n]).clone(), f.A[f.n:f.n + f.e, :f.n].clone(), f.winv.clone(), int(f.info.item())


@pytest.mark.parametrize("n,dy", [(2560, 2), (3200, 0), (4096, 1), (5248, 3), (8192, 1)])
def test_persistent_factorisation_is_bit_identical_to_the_launch_based_driver(device, n, dy):
    """functions.py:46-47 through gpn_potrf_lower_persistent: factor, extra rows (alpha^T of gpr.py:62), leaf inverses and info
    equal gpn_potrf_lower's bit for bit -- for sizes that end inside an inner / outer panel too -- and stay so over repeated runs
    (a stale tile read between two workgroups of the launch would show here first)."""
    lib = _native.lib()
    assert lib.gpn_potrf_persistent_supported(n, dy) == 1
    f = _factor_input(n, dy, device)
    saved = f.A.clone()
    ref = _run(f, saved, False)
    assert ref[3] == 0
    for rep in range(4):
        got = _run(f, saved, True)
        assert got[3] == ref[3]
        for a, b, what in zip(ref[:3], got[:3], ("factor", "extra rows", "leaf inverses")):
            assert torch.equal(a, b), (what, rep, (a - b).abs().max().item())


def test_persistent_factorisation_reports_the_first_failing_pivot(device):
    """a matrix that is not positive definite: the same LAPACK-style info as the launch-based driver (the ladder of
    functions.py:20-43 climbs on it), and the call still ends (bounded spins, finite garbage downstream)."""
    n = 4096
    f = _factor_input(n, 1, device)
    f.A[1500, 1500] = -1.0                      # a negative diagonal entry in the 12th leaf block
    saved = f.A.clone()
    ref = _run(f, saved, False)
    got = _run(f, saved, True)
    assert ref[3] > 0 and got[3] == ref[3]


def test_persistent_factorisation_unsupported_sizes_enqueue_nothing(device):
    lib = _native.lib()
    for n, dy in [(2048, 1), (4096 + 64, 1), (24576, 1)]:
        assert lib.gpn_potrf_persistent_supported(n, dy) == 0
    f = _factor_input(2048, 1, device)
    before = f.A.clone()
    rc = lib.gpn_potrf_lower_persistent(_ops._stream(device), _ops._ptr(f.A), f.n, f.e, f.ld, _ops._ptr(f.winv), _ops._ptr(f.info))
    assert rc == -102                            # GPN_E_UNSUPPORTED
    torch.cuda.synchronize()
    assert torch.equal(before, f.A)


def test_persistent_factorisation_captures_into_a_hipgraph(device):
    """after one call outside capture (the plan of a size is uploaded then) the launch is a graph node like any other."""
    n, dy = 4096, 1
    lib = _native.lib()
    f = _factor_input(n, dy, device)
    saved = f.A.clone()
    ref = _run(f, saved, True)
    s = torch.cuda.Stream(device=device)
    s.wait_stream(torch.cuda.current_stream(device))
    g = torch.cuda.CUDAGraph()
    with torch.cuda.stream(s):
        # the library keeps one runtime area per caller stream and size: the first call on THIS stream allocates it (outside capture)
        f.A.copy_(saved)
        f.info.zero_()
        assert lib.gpn_potrf_lower_persistent(_ops._stream(device), _ops._ptr(f.A), n, dy, f.ld, _ops._ptr(f.winv), _ops._ptr(f.info)) == 0
        s.synchronize()
        f.A.copy_(saved)
        f.info.zero_()
        with torch.cuda.graph(g, stream=s):
            rc = lib.gpn_potrf_lower_persistent(_ops._stream(device), _ops._ptr(f.A), n, dy, f.ld, _ops._ptr(f.winv), _ops._ptr(f.info))
            assert rc == 0
    torch.cuda.current_stream(device).wait_stream(s)
    for _ in range(2):
        f.A.copy_(saved)
        f.info.zero_()
        g.replay()
        torch.cuda.synchronize()
        assert torch.equal(torch.tril(f.A[:n, :n]), ref[0]) and torch.equal(f.A[n:n + dy, :n], ref[1]) and int(f.info.item()) == 0


# ---- the optimiser step as one graph replay ---------------------------------------------------------------------------------
def _gpr(case, device):
    x, y = rng.make_regression(case["n"], case["d"], case["dy"], seed=0)
    ls = case["length_scales"]
    if case["ARD"]:
        ls = np.asarray(ls, dtype=np.float64) * np.ones(case["d"])
    m = GPR(x, y, KERN[case["kind"]](case["d"], variance=case["variance"], length_scales=ls, ARD=case["ARD"]),
            likelihood=likelihoods.Gaussian(variance=case["noise"]))
    m.cuda()
    return m


def test_captured_optimiser_loop_follows_the_reference_trajectories(device):
    """the reference's 50-step Adam runs (tests/golden/adam_cases.json; base.py:149-151, 260-269) through optimize(capture=True):
    same tolerances as the ordinary loop's test, and the two loops agree with each other to rounding."""
    for case in load_json("adam_cases.json"):
        if case["kind"] not in KERN:
            continue
        m = _gpr(case, device)
        with contextlib.redirect_stdout(io.StringIO()) as out:
            losses, _ = m.optimize(method="Adam", max_iter=50, verbose=False, capture=True)
        ref = np.asarray(case["losses"])
        assert losses.shape == (50,)
        assert np.max(np.abs(losses - ref) / np.maximum(1.0, np.abs(ref))) < 1e-8, case["name"]
        for name, p in [("kernel.variance", m.kernel.variance), ("kernel.length_scales", m.kernel.length_scales),
                        ("likelihood.variance", m.likelihood.variance)]:
            assert np.max(np.abs(p.detach().cpu().numpy() - np.asarray(case["final"][name]))) < 1e-8, (case["name"], name)
        assert out.getvalue().count("Iter:") == 3           # idx % 20 == 0, printed after the loop
        m2 = _gpr(case, device)
        with contextlib.redirect_stdout(io.StringIO()):
            l2, _ = m2.optimize(method="Adam", max_iter=50, verbose=False)
        assert np.max(np.abs(l2 - losses) / np.maximum(1.0, np.abs(l2))) < 1e-11


@pytest.mark.parametrize("method", ["SGD", "Adagrad", "RMSprop", "Adamax"])
def test_captured_optimiser_loop_other_torch_optimisers(device, method):
    case = dict(n=300, d=3, dy=1, kind="Matern52", variance=1.1, length_scales=1.3, ARD=False, noise=0.05)
    a, b = _gpr(case, device), _gpr(case, device)
    with contextlib.redirect_stdout(io.StringIO()):
        la, _ = a.optimize(method=method, max_iter=31, verbose=False, capture=True)
        lb, _ = b.optimize(method=method, max_iter=31, verbose=False)
    assert la.shape == lb.shape == (31,)
    assert np.max(np.abs(la - lb) / np.maximum(1.0, np.abs(lb))) < 1e-9, method


def test_captured_optimiser_loop_leaves_the_graph_for_the_jitter_ladder(device):
    """a model whose factorisation needs the ladder of functions.py:20-43 (duplicated points, noise 1e-13): a replay cannot climb
    it; the chunk is rolled back and repeated eagerly -- losses equal the ordinary loop's."""
    n, d = 200, 2
    x, y = rng.make_regression(n, d, 1, seed=5)
    x[1::2] = x[0::2]                            # every point twice
    # a length scale far above the data's extent: K is numerically rank-deficient by many orders (entries 1 - O(1e-5)), pivots come
    # out negative from rounding alone unless the ladder lifts the diagonal
    setting = None
    for ls, nz in [(300.0, 1e-15), (1000.0, 1e-15), (3000.0, 1e-16)]:
        f = _ops.kernel_factor("Rbf", torch.as_tensor(x).to(device), torch.tensor([1.0], dtype=torch.float64, device=device),
                               torch.tensor([ls], dtype=torch.float64, device=device), torch.tensor([nz], dtype=torch.float64, device=device),
                               R=torch.as_tensor(y).to(device))
        if f.jitter_rung >= 0:
            setting = (ls, nz)
            break
    assert setting is not None, "no setting of this case needs the ladder any more"

    def make():
        m = GPR(x, y, kernels.Rbf(d, variance=1.0, length_scales=setting[0]), likelihood=likelihoods.Gaussian(variance=setting[1]))
        m.likelihood.variance.requires_grad_(False)
        m.kernel.length_scales.requires_grad_(False)     # (keep the model in the regime that needs the ladder for all 12 steps)
        m.cuda()
        return m
    a, b = make(), make()
    with contextlib.redirect_stdout(io.StringIO()):
        la, _ = a.optimize(method="Adam", max_iter=12, verbose=False, capture=True)
        lb, _ = b.optimize(method="Adam", max_iter=12, verbose=False)
    assert b._holder["factor"].jitter_rung >= 0, "the case is meant to need the ladder"
    assert np.max(np.abs(la - lb) / np.maximum(1.0, np.abs(lb))) < 1e-9


@pytest.mark.parametrize("method", ["Adam", "SGD", "RMSprop"])
def test_captured_lockstep_fit_matches_the_stacked_loop(device, method):
    """multi_start_optimize(capture=True): the stacked lock-step iteration (base.py:260-269 for B restarts at once) as one hipGraph
    replay -- losses and final parameters equal the uncaptured stacked loop's to rounding, over two chunks of replays"""
    def fresh():
        x, y = rng.make_regression(384, 3, 1, seed=4)
        ms = [GPR(x, y, kernels.Matern52(3, variance=0.8 + 0.1 * b, length_scales=1.0 + 0.2 * b), likelihood=likelihoods.Gaussian(variance=0.05))
              for b in range(6)]
        for m in ms:
            m.cuda()
            m.X, m.Y = ms[0].X, ms[0].Y
        return ms
    a, b = fresh(), fresh()
    with contextlib.redirect_stdout(io.StringIO()):
        la, _ = multi_start_optimize(a, method=method, max_iter=40, stacked=True)
        lb, _ = multi_start_optimize(b, method=method, max_iter=40, stacked=True, capture=True)
    assert la.shape == lb.shape == (6, 40)
    assert np.max(np.abs(la - lb) / np.maximum(1.0, np.abs(la))) < 1e-9
    for ma, mb in zip(a, b):
        for p, q in zip(ma.parameters(), mb.parameters()):
            assert torch.allclose(p.data, q.data, rtol=1e-8, atol=1e-10)


def test_captured_lockstep_fit_leaves_the_graph_for_the_jitter_ladder(device):
    """one restart of the group needs the ladder (functions.py:20-43): the replays flag it, the chunk is rolled back and repeated
    eagerly (the failing model replayed alone through the ladder) -- losses equal the uncaptured loop's"""
    n, d = 200, 2
    x, y = rng.make_regression(n, d, 1, seed=5)
    x[1::2] = x[0::2]

    def fresh():
        ms = [GPR(x, y, kernels.Rbf(d, variance=1.0, length_scales=ls), likelihood=likelihoods.Gaussian(variance=nz))
              for ls, nz in ((1.0, 1e-2), (3000.0, 1e-16), (2.0, 1e-2))]
        for m in ms:
            m.likelihood.variance.requires_grad_(False)
            m.kernel.length_scales.requires_grad_(False)
            m.cuda()
            m.X, m.Y = ms[0].X, ms[0].Y
        return ms
    probe = fresh()[1]
    probe.loss()
    if probe._holder["factor"].jitter_rung < 0:
        pytest.skip("this case does not need the ladder any more")
    a, b = fresh(), fresh()
    with contextlib.redirect_stdout(io.StringIO()):
        la, _ = multi_start_optimize(a, method="Adam", max_iter=10, stacked=True)
        lb, _ = multi_start_optimize(b, method="Adam", max_iter=10, stacked=True, capture=True)
    assert np.max(np.abs(la - lb) / np.maximum(1.0, np.abs(la))) < 1e-9


def test_capture_falls_back_for_models_it_does_not_cover(device):
    x, y = rng.make_regression(150, 2, 1, seed=2)
    m = GPR(x, y, kernels.Linear(2) + kernels.Rbf(2), likelihood=likelihoods.Gaussian(variance=0.1))
    m.cuda()
    with contextlib.redirect_stdout(io.StringIO()):
        losses, _ = m.optimize(method="Adam", max_iter=4, verbose=False, capture=True)
    assert losses.shape == (4,) and np.all(np.isfinite(losses))


# ---- functions.inverse --------------------------------------------------------------------------------------------------------
def test_functions_inverse(device):
    """functions.py:57-58 for covariance matrices: value against torch.inverse on the host, gradient against autograd's."""
    n = 300
    x, _ = rng.make_regression(n, 3, 1, seed=1)
    K = orc.kernel_K("Rbf", torch.tensor(x), None, torch.tensor([1.3]), torch.tensor([0.9])) + 0.05 * torch.eye(n, dtype=torch.float64)
    Kd = K.to(device).requires_grad_(True)
    inv = functions.inverse(Kd)
    ref = torch.inverse(K)
    assert (inv.detach().cpu() - ref).abs().max().item() < 1e-9 * ref.abs().max().item()
    w = torch.tensor(rng.normal(7, (n, n)))
    (inv * w.to(device)).sum().backward()
    Kc = K.clone().requires_grad_(True)
    (torch.inverse(Kc) * w).sum().backward()
    g = Kd.grad.cpu()
    gs, rs = 0.5 * (g + g.t()), 0.5 * (Kc.grad + Kc.grad.t())       # a symmetric argument: compare the symmetric parts
    assert (gs - rs).abs().max().item() < 1e-8 * rs.abs().max().item()
    with pytest.raises(NotImplementedError):
        functions.inverse(torch.triu(Kd.detach()) + torch.eye(n, dtype=torch.float64, device=device))


# ---- multi_start_optimize: the bitwise mode is the default where it is free -------------------------------------------------------
def test_multi_start_default_is_bitwise_from_2048_rows(device):
    n, d = 2048, 4
    x, y = rng.make_regression(n, d, 1, seed=0)
    def make(v, l):
        m = GPR(x, y, kernels.Matern52(d, variance=v, length_scales=l), likelihood=likelihoods.Gaussian(variance=0.05))
        m.cuda()
        return m
    starts = [(1.0, 1.5), (1.4, 2.0), (0.7, 1.1)]
    ms = [make(*s) for s in starts]
    with contextlib.redirect_stdout(io.StringIO()):
        losses, _ = multi_start_optimize(ms, method="Adam", max_iter=6)
        own = []
        for s in starts:
            m = make(*s)
            l, _ = m.optimize(method="Adam", max_iter=6, verbose=False)
            own.append((l, [p.detach().clone() for p in m.parameters()]))
    for b, (l, ps) in enumerate(own):
        assert np.array_equal(losses[b], l)
        for p, q in zip(ms[b].parameters(), ps):
            assert torch.equal(p.detach(), q)


# ---- DistGPR over composite kernels (native tile operations) ----------------------------------------------------------------
def test_dist_gpr_example_model_native_tiles(device):
    """the reference's example model (examples/regression_1d.py:34-53: Linear + Rbf + Constant) at BASELINE configs[1]'s size on the
    block-cyclic engine with the product's native tile operations (1 x 1 grid in this process; tiles of 2048: gpn_kernel_matrix_expr
    per tile block, gpn_kernel_expr_grad per leaf and tile block, gpn_refine_resid_part_expr above the composite refinement
    threshold): loss within north_star's 1e-8, gradients and predictions against the REFERENCE (tests/golden/composite_big_case.json)."""
    from gptorch_amd.models import DistGPR
    case = load_json("composite_big_case.json")
    d = case["d"]
    x, y = rng.make_regression(case["n"], d, case["dy"], seed=0)
    k = kernels.Linear(d, variance=0.3) + kernels.Rbf(d, variance=1.2, length_scales=float(np.sqrt(d))) + kernels.Constant(d, variance=0.4)
    m = DistGPR(x, y, k, likelihood=likelihoods.Gaussian(variance=case["noise"]), tile=2048)
    m.cuda()
    loss = m.loss()
    assert m._eng().refined, "N = 8192 is above the composite kernels' refinement threshold"
    assert abs(loss.item() - case["loss"]) < 1e-8, (loss.item(), case["loss"])
    loss.backward()
    got = {n: p.grad.cpu().numpy() for n, p in m.named_parameters() if p.grad is not None}
    assert sorted(got) == sorted(case["grads"])
    for n, r in case["grads"].items():
        r = np.asarray(r)
        assert np.abs(got[n].reshape(r.shape) - r).max() < 1e-8 * max(1.0, np.abs(r).max()), (n, got[n], r)
    xs = rng.normal(case["seed_xs"], (16, d))
    mu, var = m.predict_f(xs)
    assert np.max(np.abs(mu - np.asarray(case["mean"]))) < 1e-8
    assert np.max(np.abs(var - np.asarray(case["var"]))) < 1e-8


@pytest.mark.parametrize("idx", [0, 1])
def test_dist_gpr_composite_cases_native_tiles(device, idx):
    """the reference's composite cases (Rbf + Linear, Matern32 x Rbf-ARD; tests/golden/composite_cases.json) through DistGPR's native
    tile operations with several tiles per side (tile 128 on 400 rows): loss, gradients, predictions diag and full."""
    from gptorch_amd.models import DistGPR
    case = load_json("composite_cases.json")[idx]
    x, y = rng.make_regression(case["n"], case["d"], case["dy"], seed=0)
    if case["name"] == "rbf_plus_linear":
        kern = kernels.Rbf(3, variance=1.2, length_scales=1.5) + kernels.Linear(3, variance=np.array([0.3, 0.5, 0.7]))
    else:
        kern = kernels.Matern32(3, variance=0.9, length_scales=2.0) * kernels.Rbf(3, variance=1.1, length_scales=np.array([1.0, 2.0, 3.0]), ARD=True)
    m = DistGPR(x, y, kern, likelihood=likelihoods.Gaussian(variance=case["noise"]), tile=128)
    m.cuda()
    loss = m.loss()
    assert abs(loss.item() - case["loss"]) < 1e-9 * max(1.0, abs(case["loss"]))
    loss.backward()
    for n, r in case["grads"].items():
        r = np.asarray(r)
        g = dict(m.named_parameters())[n].grad.cpu().numpy()
        assert np.abs(g.reshape(r.shape) - r).max() < 1e-8 * max(1.0, np.abs(r).max()), n
    xs = rng.normal(case["seed_xs"], (16, case["d"]))
    mu, var = m.predict_f(xs)
    _, cov = m.predict_f(xs, diag=False)
    assert np.abs(mu - np.asarray(case["mean"])).max() < 1e-8 and np.abs(var - np.asarray(case["var"])).max() < 1e-8
    assert np.abs(cov - np.asarray(case["cov"])).max() < 1e-8


def test_dist_gpr_rejects_what_the_grid_cannot_assemble(device):
    from gptorch_amd.models import DistGPR
    x, y = rng.make_regression(100, 3, 1, seed=0)
    with pytest.raises(NotImplementedError):
        DistGPR(x, y, kernels.Matern52(3) + kernels.White(3, variance=0.05))

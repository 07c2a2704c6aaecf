"""Round-5 shell behaviour on the GPU: opt-in automatic placement of CPU-constructed models (the reference builds and
evaluates on the CPU by default, gptorch/models/base.py:82-85), the predict cache's input identity, and gpn_predict /
gpn_predict_blocked with POISONED work buffers (they are allocated uninitialised: every entry the contractions read must
have been written by the call)."""
import ctypes

import numpy as np
import pytest
import torch

from gptorch_amd import _native, _ops, kernels, likelihoods, rng, settings
from gptorch_amd.models import GPR, VFE
from oracle import gp_oracle as orc

pytestmark = pytest.mark.gpu


@pytest.fixture
def auto_device():
    old = settings.auto_device
    settings.auto_device = True
    yield
    settings.auto_device = old


def test_auto_device_predict_contract_on_a_cpu_constructed_model(device, auto_device):
    """/root/reference/test/test_models/test_base.py:83-107 on a model that was never .cuda()'d: numpy in -> numpy out,
    CPU tensor in -> CPU tensor out, shapes [n_test, dy]; values against the oracle; data and parameters end up on the GPU
    after the first call (one move, like model.cuda())."""
    n, dx, dy = 5, 3, 2
    rs = np.random.RandomState(0)
    x, y = rs.randn(n, dx), rs.randn(n, dy)
    gp = GPR(x, y, kernels.Rbf(dx, ARD=True))
    assert not gp.X.is_cuda
    x_test = rs.randn(5, dx)
    for attr in ("predict_f", "predict_y"):
        mu, v = getattr(gp, attr)(x_test)
        for result in (mu, v):
            assert isinstance(result, np.ndarray) and result.ndim == 2 and result.shape == (5, dy)
        mu_t, v_t = getattr(gp, attr)(torch.tensor(x_test))
        for result in (mu_t, v_t):
            assert isinstance(result, torch.Tensor) and not result.is_cuda and result.shape == (5, dy)
        assert np.allclose(mu_t.numpy(), mu)
    assert gp.X.is_cuda and gp.Y.is_cuda and all(p.is_cuda for p in gp.parameters())
    o = orc.GPROracle(x, y, kind="Rbf", ARD=True, noise=float(gp.likelihood.variance.transform().item()))
    with torch.no_grad():
        omu, ovar = o.predict_f(x_test)
    mu, v = gp.predict_f(x_test)
    assert np.abs(mu - omu.numpy()).max() < 1e-9 and np.abs(v - ovar.numpy()).max() < 1e-9
    assert gp.predict_f_samples(x_test, n_samples=3).shape == (3, 5, dy)


def test_auto_device_loss_optimize_and_vfe(device, auto_device):
    """loss() / optimize() (torch and scipy drivers) / a VFE model on CPU-constructed models: the first call places them."""
    import contextlib, io
    x, y = rng.make_regression(120, 2, 1, seed=3)
    m = GPR(x, y, kernels.Matern52(2), likelihood=likelihoods.Gaussian(variance=0.05))
    params = list(m.parameters())
    loss = m.loss()
    assert loss.is_cuda and loss.shape == (1,) and m.X.is_cuda
    assert all(a is b for a, b in zip(params, m.parameters()))            # the Parameter objects survive the move
    o = orc.GPROracle(x, y, kind="Matern52", noise=0.05)
    assert abs(loss.item() - o.loss().item()) < 1e-9 * abs(loss.item())
    m2 = GPR(x, y, kernels.Rbf(2) + kernels.Linear(2), likelihood=likelihoods.Gaussian(variance=0.05))
    with contextlib.redirect_stdout(io.StringIO()):
        losses, _ = m2.optimize(method="Adam", max_iter=3, verbose=False)
        res = GPR(x, y, kernels.Rbf(2), likelihood=likelihoods.Gaussian(variance=0.05)).optimize(method="L-BFGS-B", max_iter=3)
    assert losses.shape == (3,) and losses[2] < losses[0] and hasattr(res, "x")
    v = VFE(x, y, kernels.Matern52(2), num_inducing_points=15)
    assert v.loss().is_cuda and v.X.is_cuda


def test_auto_device_is_off_by_default(device):
    from gptorch_amd._native import NativeError
    assert settings.auto_device is False
    x, y = rng.make_regression(12, 2, 1, seed=1)
    with pytest.raises(NativeError, match="no CPU fallback"):
        GPR(x, y, kernels.Rbf(2)).loss()


def test_predict_cache_is_not_fooled_by_a_reallocated_input(device):
    """`_predict(x_new, x=...)` with a temporary training-input tensor (gpr.py:88-100): freed and re-allocated at the same
    address with the same shape and version but other values, it must NOT hit the factor cached for the first one
    (round-4 review: the key compared data_ptr())."""
    xa, y = rng.make_regression(300, 2, 1, seed=6)
    xb = rng.normal(77, (300, 2))
    m = GPR(xa, y, kernels.Rbf(2, length_scales=0.9), likelihood=likelihoods.Gaussian(variance=0.05))
    m.cuda()
    xs = torch.tensor(rng.normal(5, (9, 2)), device=device)
    same_address = 0
    for trial in range(4):
        t1 = torch.tensor(xa, device=device)
        p1 = t1.data_ptr()
        with torch.no_grad():
            mu_a, _ = m._predict(xs, x=t1)
        del t1
        m_held = m._predict_cache[3]                 # the cache keeps the tensor it was built from alive ...
        assert m_held.data_ptr() == p1
        t2 = torch.tensor(xb, device=device)         # ... so a new tensor cannot take its address
        same_address += int(t2.data_ptr() == p1)
        with torch.no_grad():
            mu_b, _ = m._predict(xs, x=t2)
        del t2
        oa = orc.GPROracle(xa, y, kind="Rbf", length_scales=0.9, noise=0.05)
        ob = orc.GPROracle(xb, y, kind="Rbf", length_scales=0.9, noise=0.05)
        with torch.no_grad():
            assert np.abs(mu_a.cpu().numpy() - oa.predict_f(xs.cpu().numpy())[0].numpy()).max() < 1e-9
            assert np.abs(mu_b.cpu().numpy() - ob.predict_f(xs.cpu().numpy())[0].numpy()).max() < 1e-9
    assert same_address == 0


@pytest.mark.parametrize("n,ns,dy,full_cov,blocked", [
    (1000, 37, 1, 0, False), (1000, 37, 2, 1, False),          # ragged n and ns, recursive right-solve
    (4500, 130, 1, 0, True), (4500, 130, 2, 1, True),          # the blocked solve (n >= 4096), ragged last 1024-block
    (4224, 128, 5, 0, True),                                   # dy > 4: two passes of the tail kernel
])
def test_predict_with_poisoned_work_buffers(device, n, ns, dy, full_cov, blocked):
    """gpn_predict / gpn_predict_blocked write every entry of their operand buffers that a contraction reads (the shell
    allocates them with torch.empty; only the padding is zeroed, pipeline.hip zero_padding_kernel): with the work buffers
    filled with NaN before the call the results are finite and BIT-IDENTICAL to a call on zeroed buffers."""
    d = 3
    x, y = rng.make_regression(n, d, dy, seed=8)
    X, Y = torch.tensor(x, device=device), torch.tensor(y, device=device)
    Xs = torch.tensor(rng.normal(9, (ns, d)), device=device)
    var = torch.tensor([1.2], dtype=torch.float64, device=device)
    ls = torch.tensor([1.5], dtype=torch.float64, device=device)
    nz = torch.tensor([0.05], dtype=torch.float64, device=device)
    f = _ops.kernel_factor("Matern52", X, var, ls, nz, R=Y)
    lib = _native.lib()
    one = int(lib.gpn_predict_work_bytes(n, ns, dy)) // 8
    wb = _ops.block_inverses(f) if blocked else None
    outs = []
    for fill in (0.0, float("nan")):
        work = torch.full(((2 if blocked else 1) * one,), fill, dtype=torch.float64, device=device)
        mean = torch.full((ns, dy), float("nan"), dtype=torch.float64, device=device)
        out = torch.full((ns, ns) if full_cov else (ns,), float("nan"), dtype=torch.float64, device=device)
        s = ctypes.c_void_p(torch.cuda.current_stream(device).cuda_stream)
        p = lambda t: ctypes.c_void_p(t.data_ptr())
        if blocked:
            st = lib.gpn_predict_blocked(s, _ops.KINDS["Matern52"], p(X), n, d, p(Xs), ns, None, p(var), p(ls), 1, p(f.A), f.ld, p(f.winv),
                                         p(wb), dy, full_cov, p(work), p(mean), p(out))
        else:
            st = lib.gpn_predict(s, _ops.KINDS["Matern52"], p(X), n, d, p(Xs), ns, None, p(var), p(ls), 1, p(f.A), f.ld, p(f.winv),
                                 dy, full_cov, p(work), p(mean), p(out))
        assert st == 0
        assert bool(torch.isfinite(mean).all()) and bool(torch.isfinite(out).all()), fill
        outs.append((mean, out))
    assert torch.equal(outs[0][0], outs[1][0]) and torch.equal(outs[0][1], outs[1][1])
    o = orc.GPROracle(x, y, kind="Matern52", variance=1.2, length_scales=1.5, noise=0.05)
    with torch.no_grad():
        omu, ov = o.predict_f(Xs.cpu().numpy(), diag=not full_cov)
    assert np.abs(outs[1][0].cpu().numpy() - omu.numpy()).max() < 1e-8
    ov = ov.numpy() if full_cov else ov.numpy()[:, 0]
    assert np.abs(outs[1][1].cpu().numpy() - ov).max() < 1e-8

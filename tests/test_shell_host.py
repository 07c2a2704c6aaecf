"""CPU suite, part 3: host logic of the gptorch-compatible shell (no GPU, no native
compute).  Where a behaviour needs numbers, the test injects the CPU ORACLE in place
of the two native-backed GPR methods -- the product never does that itself: on CPU
tensors it fails loudly (tested below).  Mirrors test/test_models/test_gpr.py,
test_base.py, test_model.py, test_param.py, test_mean_functions.py, test_base.py
of the reference."""
import contextlib
import io

import numpy as np
import pytest
import torch

import gptorch_amd
from gptorch_amd import kernels, likelihoods, mean_functions, param, rng, settings, util
from gptorch_amd._native import NativeError
from gptorch_amd.models import GPR
from gptorch_amd.models import base as base_mod
from oracle import gp_oracle as orc
from tests._util import load_json


def quiet():
    return contextlib.redirect_stdout(io.StringIO())


def test_import_does_not_change_default_dtype():
    """test/test_base.py:10-22."""
    assert torch.get_default_dtype() == torch.float32
    assert util.torch_dtype == torch.float64 and util.TensorType is torch.DoubleTensor


def test_param_transform_roundtrip():
    """test/test_param.py:29-54: raw = log(value) under the default positive transform."""
    p = param.Param(torch.tensor([2.5], dtype=torch.float64), transform=settings.DefaultPositiveTransform())
    assert abs(p.data.item() - np.log(2.5)) < 1e-15
    assert abs(p.transform().item() - 2.5) < 1e-15
    assert p.requires_grad and p.prior is None
    q = param.Param(torch.tensor([-1.0], dtype=torch.float64))
    assert q.transform().item() == -1.0


def test_as_tensor_types():
    assert util.as_tensor(np.ones((2, 3), dtype=np.float32)).dtype == torch.float64
    assert util.as_tensor(torch.ones(2, dtype=torch.float32)).dtype == torch.float64
    assert util.as_tensor(1.5).shape == (1,)
    with pytest.raises(TypeError):
        util.as_tensor("x")


def test_mean_functions():
    z = mean_functions.Zero(2)
    assert not z.val.requires_grad
    assert torch.equal(z(torch.randn(5, 3, dtype=torch.float64)), torch.zeros(5, 2, dtype=torch.float64))
    c = mean_functions.Constant(2, val=torch.tensor([1.0, -2.0], dtype=torch.float64))
    assert c.val.requires_grad
    assert torch.equal(c(torch.randn(4, 3, dtype=torch.float64))[3], torch.tensor([1.0, -2.0], dtype=torch.float64))
    with pytest.raises(ValueError):
        mean_functions.Constant(3, val=torch.zeros(2, dtype=torch.float64))


def test_gpr_init_and_parameter_layout():
    """test_gpr.py:24-34 + the golden parameter names / default noise."""
    api = load_json("api_cases.json")
    x, y = rng.make_regression(api["n"], api["d"], api["dy"], seed=api["seed"])
    m = GPR(x, y, kernels.Rbf(3, ARD=True))
    names = [(n, bool(p.requires_grad), list(p.shape)) for n, p in m.named_parameters()]
    assert names == [tuple(e) if False else (e[0], e[1], e[2]) for e in api["param_names"]]
    assert abs(m.likelihood.variance.transform().item() - api["default_noise_numpy"]) < 1e-15
    m2 = GPR(torch.tensor(x), torch.tensor(y), kernels.Rbf(3))
    assert abs(m2.likelihood.variance.transform().item() - api["default_noise_tensor"]) < 1e-15
    GPR(x, y, kernels.Rbf(3), mean_function=torch.nn.Linear(3, 2))
    assert m.num_data == 20 and m.input_dimension == 3 and m.output_dimension == 2
    assert m.X.dtype == torch.float64 and not m.X.requires_grad
    assert "kernel" in repr(m) and "variance" in repr(m)


def test_kernel_hyperparameter_shapes():
    k = kernels.Matern52(4, variance=2.0, length_scales=0.5)
    assert k.variance.shape == (1,) and k.length_scales.shape == (1,) and not k.ARD
    assert abs(k.length_scales.transform().item() - 0.5) < 1e-15
    ka = kernels.Rbf(4, ARD=True)
    assert ka.length_scales.shape == (4,)
    kb = kernels.Rbf(3, ARD=True, length_scales=np.array([0.25, 0.5, 0.75]))
    assert np.allclose(kb.length_scales.transform().detach().numpy(), [0.25, 0.5, 0.75])
    assert kernels.SquaredExponential is kernels.Rbf
    assert torch.equal(k.Kdiag(torch.zeros(7, 4, dtype=torch.float64)).detach(), torch.full((7,), 2.0, dtype=torch.float64))
    assert isinstance(k + ka, kernels.Sum) and isinstance(k * ka, kernels.Product)


def test_cpu_tensors_fail_loudly_no_fallback():
    x, y = rng.make_regression(12, 2, 1, seed=1)
    m = GPR(x, y, kernels.Rbf(2))
    with pytest.raises(NativeError, match="no CPU fallback"):
        m.loss()
    with pytest.raises(NativeError):
        m.predict_f(x[:3])
    with pytest.raises(NativeError):
        kernels.Rbf(2).K(torch.tensor(x))
    with pytest.raises(NativeError):
        gptorch_amd.functions.cholesky(torch.eye(3, dtype=torch.float64))
    with pytest.raises(ValueError):          # size check precedes any native call (gpr.py:56-57)
        m.loss(x=torch.tensor(x[:5]))
    with pytest.raises(NativeError):         # composite kernels take the dense-K path: native too
        GPR(x, y, kernels.Rbf(2) + kernels.Linear(2)).loss()


def test_auto_device_flag_defaults_off_and_needs_a_gpu(monkeypatch):
    """settings.auto_device (opt-in placement of CPU-constructed models): off by default; switched on without a visible GPU
    it changes nothing -- the native call still fails loudly, there is no CPU arithmetic to fall back to."""
    from gptorch_amd import settings
    assert settings.auto_device is False
    monkeypatch.setattr(settings, "auto_device", True)
    x, y = rng.make_regression(12, 2, 1, seed=1)
    m = GPR(x, y, kernels.Rbf(2))
    if not torch.cuda.is_available():
        with pytest.raises(NativeError, match="no CPU fallback"):
            m.loss()
        assert not m.X.is_cuda


def test_refine_threshold_override_is_taken_verbatim(monkeypatch):
    """GPN_REFINE_MIN_N overrides the size from which the quadratic form is refined for EVERY caller as given (round-4
    advice: expression / grid callers used to scale an explicit override by 1/2 and 2/3); the factors apply to the
    built-in default only."""
    from gptorch_amd import _ops
    monkeypatch.delenv("GPN_REFINE_MIN_N", raising=False)
    assert _ops.refine_min_n() == 12288 and _ops.refine_min_n(expression=True) == 6144 and _ops.refine_min_n(grid=True) == 8192
    monkeypatch.setenv("GPN_REFINE_MIN_N", "16384")
    assert _ops.refine_min_n() == _ops.refine_min_n(expression=True) == _ops.refine_min_n(grid=True) == 16384
    monkeypatch.setenv("GPN_REFINE_MIN_N", "0")
    assert _ops.refine_min_n() == _ops.refine_min_n(expression=True) == _ops.refine_min_n(grid=True) == 1 << 62


def test_jit_op_is_the_same_ladder():
    """functions.jit_op (functions.py:20-43) for a caller-supplied op: plain try, then
    x + 10^(-10+i) I, then RuntimeError("Max tries exceeded.") -- on top of _ops._ladder."""
    from gptorch_amd import functions
    seen = []

    def op(m):
        seen.append(m[0, 0].item())
        if m[0, 0].item() < 1.0 + 5e-7:
            raise RuntimeError("not yet")
        return m * 2.0

    out = functions.jit_op(op, torch.eye(2, dtype=torch.float64))
    assert seen == [1.0] + [1.0 + 10.0 ** (-10 + i) for i in range(5)]
    assert out[0, 0].item() == 2.0 * (1.0 + 1e-6) and out[0, 1].item() == 0.0

    def never(m):
        raise RuntimeError("no")
    with pytest.raises(RuntimeError, match="Max tries exceeded."):
        functions.jit_op(never, torch.eye(2, dtype=torch.float64))
    # max_tries sets both the number of rungs and the first jitter, 10^(-max_tries + i) (functions.py:34-36)
    seen.clear()
    functions.jit_op(lambda m: op(m) if m[0, 0].item() < 1.005 else m, torch.eye(2, dtype=torch.float64), max_tries=3)
    assert seen == [1.0, 1.0 + 1e-3] and True
    calls = []

    def counting(m):
        calls.append(m[0, 0].item())
        raise RuntimeError("no")
    with pytest.raises(RuntimeError, match="Max tries exceeded."):
        functions.jit_op(counting, torch.eye(2, dtype=torch.float64), max_tries=3)
    assert calls == [1.0, 1.0 + 1e-3, 1.0 + 1e-2, 1.0 + 1e-1]
    # the reference catches any Exception on the initial try but only RuntimeError on the jittered ones (functions.py:30, 38)

    def value_error_first(m):
        if m[0, 0].item() == 1.0:
            raise ValueError("initial")
        return m
    assert functions.jit_op(value_error_first, torch.eye(2, dtype=torch.float64))[0, 0].item() == 1.0 + 1e-10

    def value_error_later(m):
        raise (RuntimeError("first") if m[0, 0].item() == 1.0 else ValueError("later"))
    with pytest.raises(ValueError):
        functions.jit_op(value_error_later, torch.eye(2, dtype=torch.float64))
    with pytest.raises(NativeError):          # the differentiable surface has no CPU path either
        gptorch_amd.util.squared_distance(torch.zeros(3, 2, dtype=torch.float64))


def test_jitter_ladder_logic():
    """functions.py:20-43 replayed by _ops._ladder on the LAPACK-style info."""
    from gptorch_amd import _ops
    calls = []

    def attempt_factory(ok_at):
        def attempt(j):
            calls.append(j)
            return 0 if (j is not None and j >= ok_at) else 3
        return attempt
    assert _ops._ladder(lambda j: 0) == -1
    calls.clear()
    assert _ops._ladder(attempt_factory(1e-7)) == 3
    assert calls[0] is None and np.allclose(calls[1:], [1e-10, 1e-9, 1e-8, 1e-7])
    with pytest.raises(RuntimeError, match="Max tries exceeded."):
        _ops._ladder(lambda j: 1)


@pytest.fixture
def oracle_backed(monkeypatch):
    """GPR whose two native-backed methods are answered by the CPU oracle (tests only)."""
    def _o(self):
        k = self.kernel
        o = orc.GPROracle(self.X.numpy(), self.Y.numpy(), kind=k._kind, ARD=k.ARD)
        o.raw_variance, o.raw_length_scales, o.raw_noise = k.variance, k.length_scales, self.likelihood.variance
        o.mean_val = self.mean_function.val
        return o

    def log_likelihood(self, x=None, y=None):
        x = x if x is not None else self.X
        y = y if y is not None else self.Y
        if not x.shape[0] == y.shape[0]:
            raise ValueError("X and Y must have same # data.")
        return _o(self).log_likelihood(x, y)

    def _predict(self, x_new, diag=True, x=None):
        return _o(self).predict_f(x_new, diag=diag)

    monkeypatch.setattr(GPR, "log_likelihood", log_likelihood)
    monkeypatch.setattr(GPR, "_predict", _predict)
    monkeypatch.setattr(base_mod, "cholesky", orc.cholesky)
    x, y = rng.make_regression(30, 2, 2, seed=2)
    return GPR(x, y, kernels.Rbf(2, ARD=True)), x, y


def test_loss_and_flat_parameter_glue(oracle_backed):
    """test/test_model.py:55-115 behaviours on a real model."""
    m, x, y = oracle_backed
    loss = m.loss()
    assert loss.shape == (1,) and m.compute_loss().item() == loss.item()
    assert m.loss(x=torch.tensor(x), y=torch.tensor(y)).item() == loss.item()
    p0 = m._get_param_array()
    assert p0.shape == (4,)                      # variance, 2 length-scales, noise; mean is frozen
    m._set_parameters(p0 + 0.1)
    assert np.allclose(m._get_param_array(), p0 + 0.1)
    with quiet():
        f, g = m._loss_and_grad(p0)
    assert isinstance(f, float) and g.shape == (4,) and g.dtype == np.float64 and np.all(np.isfinite(g))
    assert m.log_prior() == 0.0
    m.kernel.variance.prior = torch.distributions.Gamma(torch.tensor(2.0, dtype=torch.float64),
                                                         torch.tensor(1.0, dtype=torch.float64))
    assert abs(m.log_prior().item() - torch.distributions.Gamma(2.0, 1.0).log_prob(torch.tensor(1.0)).item()) < 1e-6
    assert abs(m.loss().item() - (loss.item() - m.log_prior().item())) < 1e-12


def test_optimize_torch_and_scipy(oracle_backed):
    """test/test_models/test_base.py:50-53 + return contracts (base.py:288-296, 298-320)."""
    m, _, _ = oracle_backed
    with quiet():
        losses, t = m.optimize(method="Adam", max_iter=3, verbose=False)
    assert losses.shape == (3,) and t > 0 and losses[2] < losses[0]
    with quiet():
        losses, _ = m.optimize(method="LBFGS", max_iter=2, verbose=True)
    assert len(losses) <= 2
    with quiet():
        res = m.optimize(method="L-BFGS-B", max_iter=2)
    assert hasattr(res, "x") and res.x.shape == (4,)
    with pytest.raises(ValueError):
        m.optimize(method="NotAnOptimizer")


def test_predict_wrappers(oracle_backed):
    """test_base.py:83-163: numpy in -> numpy out, tensor in -> tensor out, shapes, + noise."""
    m, x, _ = oracle_backed
    xs = rng.normal(4, (7, 2))
    mu, var = m.predict_f(xs)
    assert isinstance(mu, np.ndarray) and mu.shape == (7, 2) and var.shape == (7, 2)
    mu_t, var_t = m.predict_f(torch.tensor(xs))
    assert isinstance(mu_t, torch.Tensor) and np.allclose(mu_t.detach().numpy(), mu)
    _, var_y = m.predict_y(xs)
    assert np.allclose(var_y - var, m.likelihood.variance.transform().item())
    _, cov = m.predict_f(xs, diag=False)
    _, cov_y = m.predict_y(xs, diag=False)
    assert cov.shape == (7, 7) and np.allclose(np.diag(cov_y) - np.diag(cov), m.likelihood.variance.transform().item())
    torch.manual_seed(0)
    assert m.predict_f_samples(xs, n_samples=5).shape == (5, 7, 2)
    assert m.predict_y_samples(torch.tensor(xs), n_samples=3).shape == (3, 7, 2)


def test_gaussian_likelihood_known_answer():
    """test/test_likelihoods.py:45-59: Gaussian.logp known answer."""
    lik = likelihoods.Gaussian(variance=1.0)
    lp = lik.logp(torch.zeros(1, dtype=torch.float64), torch.zeros(1, dtype=torch.float64))
    assert abs(lp.item() - (-0.5 * np.log(2 * np.pi))) < 1e-12
    m, v = lik.predict_mean_variance(torch.zeros(3, 1, dtype=torch.float64), torch.ones(3, 1, dtype=torch.float64))
    assert torch.allclose(v, torch.full((3, 1), 2.0, dtype=torch.float64))


def test_rng_is_deterministic():
    x, y = rng.make_regression(8192, 8, 1, seed=0)
    c2 = [c for c in load_json("lml_cases.json") if c["name"] == "C2_rbf_8192_8"][0]
    assert rng.checksum(x) == c2["x_checksum"] and rng.checksum(y) == c2["y_checksum"]
    assert abs(x.mean()) < 0.01 and abs(x.std() - 1) < 0.01


def test_expression_program_builder():
    """gptorch_amd._expr.build: a Sum / Product tree (kernels.py:286-306) expands into a sum of products of leaf
    instances with all leaf parameters packed into one vector -- pure host logic, no GPU."""
    from gptorch_amd import _expr, _native
    d = 3
    rbf, m52 = kernels.Rbf(d, variance=0.7), kernels.Matern52(d, length_scales=np.array([1.0, 2.0, 3.0]), ARD=True)
    lin, bias = kernels.Linear(d, variance=0.35), kernels.Bias(d, variance=0.2)
    p = _expr.build((rbf + bias) * (m52 + lin))
    assert p.groups == [[0, 1], [0, 2], [3, 1], [3, 2]] and [type(k).__name__ for k in p.leaves] == ["Rbf", "Matern52", "Linear", "Bias"]
    assert p.offsets == [(0, 1, 1, 1), (2, 1, 3, 3), (6, 3, 9, 0), (9, 1, 10, 0)] and p.ntheta == 10
    assert list(p.gstart) == [0, 2, 4, 6, 8] and len(p.instances) == 8
    t = p.terms[1]
    assert (t.type, t.kind, t.var_off, t.ls_off, t.nls) == (_native.TERM_STATIONARY, 1, 2, 3, 3)
    assert p.terms[3].type == _native.TERM_LINEAR and p.terms[3].nvar == 3 and p.terms[4].type == _native.TERM_CONSTANT
    assert p.grad_supported(3) and not _expr.build(kernels.Rbf(20, length_scales=np.ones(20), ARD=True) + bias2(20)).grad_supported(20)
    theta = p.theta(p.params())
    assert theta.shape == (10,) and abs(theta[0].item() - 0.7) < 1e-15 and torch.allclose(theta[3:6], torch.tensor([1.0, 2.0, 3.0], dtype=torch.float64))
    # instances of one leaf add their gradients up; White is a leaf, a lone stationary kernel is one too
    g = p.scatter([torch.ones(2), torch.ones(4), torch.ones(2), torch.ones(3), torch.ones(1), torch.ones(4), torch.ones(1), torch.ones(3)], p.params())
    assert [float(x.sum()) for x in g] == [2.0, 2.0, 2.0, 6.0, 6.0, 2.0]
    assert _expr.build(kernels.Matern32(d) + kernels.White(d)).groups == [[0], [1]]
    wide = rbf + bias
    for _ in range(3):
        wide = wide * (kernels.Rbf(d) + kernels.Bias(d))
    assert _expr.build(wide) is None                     # 16 product groups > 8: composed path

    class Custom(kernels.Kernel):
        def K(self, X, X2=None):
            return X @ (X if X2 is None else X2).t()
    assert _expr.build(rbf + Custom(d)) is None          # a leaf without a native term: composed path


def bias2(d):
    return kernels.Bias(d, variance=0.1)


def test_bench_self_launch_reports_a_dead_child(tmp_path):
    """`python bench.py --gpus 2` with no launcher starts its ranks itself (a child torch.distributed.run).  Here there is no
    GPU, so the ranks die at once: the parent must relay the child's exit code AND leave one JSON line that says so
    (value null + the reason) instead of a usage message."""
    import json
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env["CUDA_VISIBLE_DEVICES"] = env["HIP_VISIBLE_DEVICES"] = ""
    out = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--workload", "c1", "--steps", "1", "--warmup", "0"],
                         env=env, capture_output=True, text=True, timeout=300, cwd=root)
    assert out.returncode != 0
    lines = [ln for ln in out.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, out.stdout[-2000:]
    d = json.loads(lines[0])
    assert d["value"] is None and d["n_gpus"] == 2 and d["scaling"] == "strong" and "without a result line" in d["error"]


def _two_models(x, y):
    return [GPR(x, y, kernels.Rbf(2, ARD=True)), GPR(x, y, kernels.Rbf(2, variance=1.3, length_scales=0.8, ARD=True))]


def _run_with_deadline(fn, seconds):
    """fn() in a helper thread: (finished in time, result or exception)."""
    import threading
    box = {}

    def target():
        try:
            box["result"] = fn()
        except BaseException as exc:              # noqa: BLE001 -- reported to the caller
            box["error"] = exc
    t = threading.Thread(target=target, daemon=True)
    t.start()
    t.join(seconds)
    return not t.is_alive(), box


@pytest.mark.parametrize("method", base_mod._SCIPY_METHODS)
def test_multi_start_scipy_never_hangs(oracle_backed, method):
    """round-5 advisor finding: multi_start_optimize(method="COBYLA") hung forever -- scipy runs COBYLA under a module-wide lock,
    so the second restart's thread never reached its objective and the collecting loop waited for it.  Every scipy method the
    reference lists (base.py:203-215) now either returns one result per restart or raises (the Hessian-based ones do, as in the
    reference) within seconds, and leaves no worker thread behind."""
    import threading
    from gptorch_amd.models import multi_start_optimize
    _, x, y = oracle_backed
    models = _two_models(x, y)
    before = set(threading.enumerate())
    with quiet():
        finished, box = _run_with_deadline(lambda: multi_start_optimize(models, method=method, max_iter=2), 120)
    assert finished, "multi_start_optimize(method=%r) did not come back" % method
    if "result" in box:
        results, seconds = box["result"]
        assert len(results) == 2 and all(hasattr(r, "x") for r in results)
    else:
        assert isinstance(box["error"], Exception)
    leftovers = [t for t in threading.enumerate() if t not in before and t.is_alive()]
    for t in leftovers:
        t.join(15)
    assert not [t for t in leftovers if t.is_alive()]


def test_multi_start_scipy_is_interruptible(oracle_backed, monkeypatch):
    """an exception that is not an evaluation failure (KeyboardInterrupt) ends the search: it propagates to the caller, every
    restart's `minimize` unwinds through _MultiStartAborted and no thread is left waiting for an answer."""
    import threading
    from gptorch_amd.models import gpr as gpr_mod
    _, x, y = oracle_backed
    models = _two_models(x, y)
    real = gpr_mod.batched_loss_and_grad
    calls = {"n": 0}

    def flaky(ms):
        calls["n"] += 1
        if calls["n"] == 2:
            raise KeyboardInterrupt()
        return real(ms)
    monkeypatch.setattr(gpr_mod, "batched_loss_and_grad", flaky)
    before = set(threading.enumerate())
    with quiet():
        finished, box = _run_with_deadline(lambda: gpr_mod.multi_start_optimize(models, method="L-BFGS-B", max_iter=5), 120)
    assert finished
    assert isinstance(box.get("error"), KeyboardInterrupt)
    leftovers = [t for t in threading.enumerate() if t not in before and t.is_alive()]
    for t in leftovers:
        t.join(15)
    assert not [t for t in leftovers if t.is_alive()]


def test_multi_start_scipy_round_does_not_wait_for_a_silent_restart(oracle_backed, monkeypatch):
    """the collecting loop's stall time-out: a restart that neither posts a request nor finishes (here: its objective blocks on
    a lock held elsewhere for a while) does not stop the others from being served."""
    import threading
    import time
    from gptorch_amd.models import gpr as gpr_mod
    _, x, y = oracle_backed
    models = _two_models(x, y)
    monkeypatch.setattr(gpr_mod, "_MULTI_START_STALL_S", 0.2)
    gate = threading.Event()
    real_get = GPR._get_param_array

    served = []
    real = gpr_mod.batched_loss_and_grad

    def spy(ms):
        served.append(len(ms))
        if len(served) == 3:
            gate.set()                       # from now on the slow restart may post too
        return real(ms)
    monkeypatch.setattr(gpr_mod, "batched_loss_and_grad", spy)
    import scipy.optimize
    real_min = scipy.optimize.minimize

    def slow_minimize(fun, x0, **kw):
        if np.allclose(x0, real_get(models[1])):
            gate.wait(30)                    # the second restart sits here while the first is served alone
        return real_min(fun=fun, x0=x0, **kw)
    monkeypatch.setattr(scipy.optimize, "minimize", slow_minimize)
    with quiet():
        finished, box = _run_with_deadline(lambda: gpr_mod.multi_start_optimize(models, method="L-BFGS-B", max_iter=3), 120)
    assert finished and "result" in box, box.get("error")
    assert served[0] == 1, "the first round served the one restart that had posted"

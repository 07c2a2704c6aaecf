"""The refinement step's back-substitution a_hat = L^-T alpha (refine.hip; follows GPR.log_likelihood, gptorch/models/gpr.py:61-67)
as ONE persistent launch whose workgroups hand a_k on through self-validating values in device memory (round 5), against the
one-launch-per-128-column-block form it replaces (tools' build switch): bit-identical refined terms at ragged sizes and several
right-hand sides, repeatable bit for bit, and against the oracle's plain value."""
import ctypes

import numpy as np
import pytest
import torch

from gptorch_amd import _native, _ops, rng
from oracle import gp_oracle as orc

pytestmark = pytest.mark.gpu


def _refined_terms(lib, kind, X, Y, var, ls, nz, f, terms):
    n, d = X.shape
    dy = Y.shape[1]
    out = terms.clone()
    st = lib.gpn_lml_refine(_ops._stream(X.device), _ops.KINDS[kind], _ops._ptr(X), n, d, _ops._ptr(Y), None, dy, _ops._ptr(var), _ops._ptr(ls),
                            ls.numel(), _ops._ptr(nz), _ops._ptr(f.A), f.ld, _ops._ptr(f.winv), _ops._ptr(f._refine_work), _ops._ptr(out))
    assert st == 0
    return out


@pytest.mark.parametrize("n,d,dy", [(129, 2, 1), (300, 2, 2), (700, 3, 5), (4096, 4, 1), (12288, 8, 1), (16001, 8, 3)])
def test_persistent_back_substitution_is_bit_identical_to_the_stepwise_one(device, n, d, dy):
    x, y = rng.make_regression(n, d, dy, seed=0)
    X, Y = torch.as_tensor(x).to(device), torch.as_tensor(y).to(device)
    var = torch.tensor([1.1], dtype=torch.float64, device=device)
    ls = torch.tensor([float(np.sqrt(d))], dtype=torch.float64, device=device)
    nz = torch.tensor([2e-2], dtype=torch.float64, device=device)
    lib = _native.debug_begin()
    try:
        lib.gpn_debug_set_backsub_persistent.restype = ctypes.c_int
        lib.gpn_debug_set_backsub_persistent.argtypes = [ctypes.c_int]
        f, terms = _ops.lml_forward("Matern52", X, Y, var, ls, nz, refine=True)       # allocates the refine workspace
        plain = f.lml_terms()
        lib.gpn_debug_set_backsub_persistent(0)
        step = _refined_terms(lib, "Matern52", X, Y, var, ls, nz, f, plain)
        lib.gpn_debug_set_backsub_persistent(1)
        runs = [_refined_terms(lib, "Matern52", X, Y, var, ls, nz, f, plain) for _ in range(5)]
    finally:
        lib.gpn_debug_set_backsub_persistent(1)
        _native.debug_end()
    for r in runs:
        assert torch.equal(r, step), (r, step)
    assert torch.equal(terms, step)                      # what lml_forward(refine=True) returned is that value
    if n <= 4096:
        o = orc.GPROracle(x, y, kind="Matern52", variance=1.1, length_scales=float(np.sqrt(d)), noise=2e-2)
        with torch.no_grad():
            ref = o.log_likelihood().item()
        assert abs(step[2].item() - ref) < 1e-8 * max(1.0, abs(ref))


def test_persistent_back_substitution_while_the_chip_is_busy(device):
    """the hand-over must not depend on all workgroups being resident at once: the same refinement with another stream
    keeping every compute unit busy (roles are handed out by ticket in start order) -- same bits, no hang."""
    n, d = 20000, 8
    x, y = rng.make_regression(n, d, 1, seed=1)
    X, Y = torch.as_tensor(x).to(device), torch.as_tensor(y).to(device)
    var = torch.tensor([1.0], dtype=torch.float64, device=device)
    ls = torch.tensor([float(np.sqrt(d))], dtype=torch.float64, device=device)
    nz = torch.tensor([1e-2], dtype=torch.float64, device=device)
    f, terms = _ops.lml_forward("Rbf", X, Y, var, ls, nz, refine=True)
    lib = _native.lib()
    plain = f.lml_terms()
    quiet = _refined_terms(lib, "Rbf", X, Y, var, ls, nz, f, plain)
    side = torch.cuda.Stream(device=device)
    big = torch.randn(8192, 8192, dtype=torch.float64, device=device)
    for _ in range(3):
        with torch.cuda.stream(side):
            for _ in range(4):
                big @ big                                # rocBLAS work on every CU underneath
        busy = _refined_terms(lib, "Rbf", X, Y, var, ls, nz, f, plain)
        torch.cuda.synchronize()
        assert torch.equal(busy, quiet)


@pytest.mark.parametrize("n,d,dy,kind", [(12288, 8, 1, "Matern52"), (5000, 3, 2, "Rbf")])
def test_refinement_from_the_saved_copy_is_bit_identical(device, monkeypatch, n, d, dy, kind):
    """gpn_lml_forward_saving keeps a pristine copy of Kyy's lower triangle next to the factor (opt-in, GPN_REFINE_SAVED_K=1);
    gpn_lml_refine_dense then READS the matrix for its residual pass where gpn_lml_refine re-computes every entry: the same
    entries, the same partial sums -- the same refined terms bit for bit (and the factor itself is untouched by the copy)."""
    x, y = rng.make_regression(n, d, dy, seed=5)
    X, Y = torch.as_tensor(x).to(device), torch.as_tensor(y).to(device)
    var = torch.tensor([0.9], dtype=torch.float64, device=device)
    ls = torch.tensor([float(np.sqrt(d)) * 0.8], dtype=torch.float64, device=device)
    nz = torch.tensor([3e-2], dtype=torch.float64, device=device)
    monkeypatch.setenv("GPN_REFINE_SAVED_K", "0")
    f0, t0 = _ops.lml_forward(kind, X, Y, var, ls, nz, refine=True)
    monkeypatch.setenv("GPN_REFINE_SAVED_K", "1")
    f1, t1 = _ops.lml_forward(kind, X, Y, var, ls, nz, refine=True)
    assert getattr(f0, "_ksave", None) is None and f1._ksave is not None
    assert torch.equal(t0, t1), (t0, t1)
    assert torch.equal(torch.tril(f0.A[:n, :n]), torch.tril(f1.A[:n, :n]))
    # the copy IS Kyy: sampled rows against the assembly
    K = _ops.kernel_matrix(kind, X[:64], X, var, ls)
    K[:, :64] += torch.eye(64, dtype=torch.float64, device=device) * nz
    assert torch.equal(torch.tril(f1._ksave[:64, :64]), torch.tril(K[:64, :64]))


def test_refined_evaluation_captures_into_a_hipgraph(device):
    """the refinement step -- the sentinel fill, the persistent back-substitution (workgroups handing a_k on through device
    memory), the double-double residual pass -- captures into ONE hipGraph together with the evaluation before it; replays
    reproduce the eager refined terms bit for bit and follow new hyper-parameters written into the captured tensors."""
    n, d = 12288, 8
    x, y = rng.make_regression(n, d, 1, seed=2)
    X, Y = torch.as_tensor(x).to(device), torch.as_tensor(y).to(device)
    var = torch.tensor([1.0], dtype=torch.float64, device=device)
    ls = torch.tensor([float(np.sqrt(d))], dtype=torch.float64, device=device)
    nz = torch.tensor([2e-2], dtype=torch.float64, device=device)
    lib = _native.lib()
    f, eager = _ops.lml_forward("Matern52", X, Y, var, ls, nz, refine=True)      # warm-up: creates side streams, workspaces
    out = torch.empty(3, dtype=torch.float64, device=device)

    def enqueue():
        st = lib.gpn_lml_forward(_ops._stream(device), _ops.KINDS["Matern52"], _ops._ptr(X), n, d, _ops._ptr(Y), None, 1, _ops._ptr(var),
                                 _ops._ptr(ls), 1, _ops._ptr(nz), _ops._ptr(f.A), f.ld, _ops._ptr(f.winv), _ops._ptr(f.info), _ops._ptr(out))
        assert st == 0
        st = lib.gpn_lml_refine(_ops._stream(device), _ops.KINDS["Matern52"], _ops._ptr(X), n, d, _ops._ptr(Y), None, 1, _ops._ptr(var),
                                _ops._ptr(ls), 1, _ops._ptr(nz), _ops._ptr(f.A), f.ld, _ops._ptr(f.winv), _ops._ptr(f._refine_work), _ops._ptr(out))
        assert st == 0

    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        enqueue()
    for _ in range(3):
        g.replay()
    torch.cuda.synchronize()
    assert int(f.info.item()) == 0 and torch.equal(out, eager), (out, eager)
    ls.mul_(1.2)                                         # same graph, new hyper-parameters
    g.replay()
    torch.cuda.synchronize()
    _, eager2 = _ops.lml_forward("Matern52", X, Y, var, ls, nz, refine=True)
    assert torch.equal(out, eager2)

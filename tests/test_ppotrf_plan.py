"""The task graph of the persistent factorisation (csrc/ppotrf.hip, gpn_potrf_persistent_plan) replayed on the host: executed
in RANDOM valid orders (only the predecessor counters decide what may run) it must give the Cholesky factor and the solved
extra rows -- i.e. the <= 3 predecessors per task really cover every tile a task reads (functions.py:46-47 is what the whole
graph replaces).  No GPU: the plan is host code."""
import ctypes

import numpy as np
import pytest

from gptorch_amd import _native

LEAF, UPD, TRSM = 0, 2, 1


def _plan(n, e):
    lib = _native.lib()
    counts = (ctypes.c_int64 * 5)()
    rc = lib.gpn_potrf_persistent_plan(n, e, counts, None, 0, None, 0)
    assert rc == 0, rc
    nt, ns = counts[0], counts[1]
    tasks = np.zeros((nt, 8), dtype=np.int32)
    succ = np.zeros(max(1, ns), dtype=np.int32)
    rc = lib.gpn_potrf_persistent_plan(n, e, counts, tasks.ctypes.data, nt, succ.ctypes.data, ns)
    assert rc == 0, rc
    return list(counts), tasks, succ[:ns]


def _succ_of(tasks, succ, t):
    b = tasks[t, 7]
    e = tasks[t + 1, 7] if t + 1 < len(tasks) else len(succ)
    return succ[b:e]


def test_supported_sizes():
    lib = _native.lib()
    assert lib.gpn_potrf_persistent_supported(8192, 1) == 1
    assert lib.gpn_potrf_persistent_supported(16384, 2) == 1
    assert lib.gpn_potrf_persistent_supported(8192 + 64, 1) == 0       # not a multiple of 128
    assert lib.gpn_potrf_persistent_supported(2048, 1) == 0            # one panel level only
    assert lib.gpn_potrf_persistent_supported(32768, 1) == 0           # the launch-based driver's regime
    assert lib.gpn_potrf_persistent_supported(8192, 17) == 0
    assert lib.gpn_potrf_persistent_plan(2048, 1, None, None, 0, None, 0) != 0


@pytest.mark.parametrize("n,e", [(2560, 2), (4096, 1), (3200, 0)])
def test_graph_shape(n, e):
    counts, tasks, succ = _plan(n, e)
    nt = counts[0]
    T = n // 128
    TR = T + (1 if e else 0)
    assert counts[2] + counts[3] + counts[4] == nt
    assert (tasks[:, 6] <= 3).all() and (tasks[:, 6] >= 0).all()
    # predecessor counts agree with the successor lists, and the listed order is a valid sequential order
    indeg = np.zeros(nt, dtype=np.int64)
    for t in range(nt):
        for s in _succ_of(tasks, succ, t):
            assert s > t, "the task list must be a topological order"
            indeg[s] += 1
    assert (indeg == tasks[:, 6]).all()
    assert (tasks[:, 0] == LEAF).sum() == T
    assert (tasks[:, 0] == TRSM).sum() == sum(TR - 1 - k for k in range(T))
    # exactly one task starts ready: the first leaf
    ready = np.nonzero(tasks[:, 6] == 0)[0]
    assert list(ready) == [0] and tasks[0, 0] == LEAF
    # queue 0 = everything but top-level updates inside one outer panel's diagonal triangle
    for t in range(nt):
        ty, q, i, j = tasks[t, :4]
        if q == 0:
            assert i < T and i // 8 == j // 8


@pytest.mark.parametrize("n,e,seed", [(2560, 2, 0), (2560, 2, 1), (4096, 1, 2), (3200, 0, 3)])
def test_replay_in_random_valid_order(n, e, seed):
    counts, tasks, succ = _plan(n, e)
    nt = counts[0]
    T = n // 128
    rng = np.random.default_rng(seed)
    d = 6
    x = rng.standard_normal((n, d))
    sq = ((x[:, None, :] - x[None, :, :]) ** 2).sum(-1) if n <= 1024 else None
    if sq is None:
        g = x @ x.T
        dg = np.diag(g)
        sq = np.maximum(dg[:, None] + dg[None, :] - 2 * g, 0.0)
    K = np.exp(-0.5 * sq / d) + 1e-2 * np.eye(n)
    R = rng.standard_normal((e, n))
    A = np.vstack([np.tril(K), R])               # rows n.. = the extra rows
    W = np.zeros((T, 128, 128))
    rows = lambda i: slice(i * 128, (i + 1) * 128) if i < T else slice(n, n + e)
    cols = lambda j: slice(j * 128, (j + 1) * 128)
    dep = tasks[:, 6].astype(np.int64).copy()
    ready = [0]
    done = 0
    while ready:
        t = ready.pop(int(rng.integers(len(ready))))
        ty, q, i, j, k0, k1 = tasks[t, :6]
        if ty == LEAF:
            blk = A[rows(i), cols(i)]
            full = np.tril(blk) + np.tril(blk, -1).T
            L = np.linalg.cholesky(full)
            A[rows(i), cols(i)] = L
            W[i] = np.linalg.inv(L)
        elif ty == TRSM:
            A[rows(i), cols(j)] = A[rows(i), cols(j)] @ W[j].T
        else:
            ks = slice(k0 * 128, k1 * 128)
            upd = A[rows(i), ks] @ A[rows(j), ks].T
            if i == j:
                upd = np.tril(upd)
            A[rows(i), cols(j)] -= upd
        done += 1
        for s in _succ_of(tasks, succ, t):
            dep[s] -= 1
            assert dep[s] >= 0
            if dep[s] == 0:
                ready.append(int(s))
    assert done == nt, "every task ran exactly once"
    Lref = np.linalg.cholesky(K)
    err = np.abs(np.tril(A[:n]) - Lref).max()
    assert err < 1e-9, err
    if e:
        want = np.linalg.solve(Lref, R.T).T       # R L^-T
        assert np.abs(A[n:] - want).max() < 1e-8

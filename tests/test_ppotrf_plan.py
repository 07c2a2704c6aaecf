"""The task graph of the persistent factorisation (csrc/ppotrf.hip, gpn_potrf_persistent_plan) replayed on the host: executed
in RANDOM valid orders (only the predecessor counters decide what may run; a chain step's first phase releases its readers
before the step is over) it must give the Cholesky factor and the solved extra rows -- i.e. the <= 4 predecessors per task really
cover every tile a task reads (functions.py:46-47 is what the whole graph replaces).  No GPU: the plan is host code."""
import ctypes

import numpy as np
import pytest

from gptorch_amd import _native

LEAF, TRSM, UPD, STEP, PRED, SUB = 0, 1, 2, 3, 4, 5
ACC_OUT, ACC_IN, HALF1, HALF0 = 1, 2, 4, 8


def _plan(n, e):
    lib = _native.lib()
    counts = (ctypes.c_int64 * 67)()
    rc = lib.gpn_potrf_persistent_plan(n, e, counts, None, 0, None, 0)
    assert rc == 0, rc
    nt, ns = counts[0], counts[1]
    tasks = np.zeros((nt, 12), dtype=np.int32)
    succ = np.zeros(max(1, ns), dtype=np.int32)
    rc = lib.gpn_potrf_persistent_plan(n, e, counts, tasks.ctypes.data, nt, succ.ctypes.data, ns)
    assert rc == 0, rc
    return list(counts), tasks, succ[:ns]


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
    assert sum(counts[3:3 + counts[2]]) == nt and counts[2] == 2 + (T + 7) // 8
    assert (tasks[:, 6] <= 8).all() and (tasks[:, 6] >= 0).all()
    # predecessor counts agree with the successor lists, and the listed order is a valid sequential order
    indeg = np.zeros(nt, dtype=np.int64)
    pos = 0
    for t in range(nt):
        b, m, e_ = tasks[t, 8:11]
        assert b == pos and b <= m <= e_
        pos = e_
        if tasks[t, 0] != STEP:
            assert m == b, "only a chain step has a first phase"
        for s in succ[b:e_]:
            assert s > t, "the task list must be a topological order"
            indeg[s] += 1
    assert pos == len(succ)
    assert (indeg == tasks[:, 6]).all()
    assert (tasks[:, 0] == LEAF).sum() == 1 and (tasks[:, 0] == STEP).sum() == T - 1
    # every off-diagonal tile below the diagonal is solved exactly once: by two half-tile solves (one for the extra-rows tile) or
    # by the step of its row
    halves = ((tasks[:, 0] == TRSM) & ((tasks[:, 7] & (HALF0 | HALF1)) != 0)).sum()
    whole = ((tasks[:, 0] == TRSM) & ((tasks[:, 7] & (HALF0 | HALF1)) == 0)).sum()
    assert halves // 2 + whole + (T - 1) == sum(TR - 1 - k for k in range(T)) and halves % 2 == 0
    ready = np.nonzero(tasks[:, 6] == 0)[0]
    assert list(ready) == [0] and tasks[0, 0] == LEAF
    # the scratch tiles are written once before they are read
    for t in range(nt):
        if tasks[t, 0] == STEP and tasks[t, 7] & ACC_IN:
            c = tasks[t, 2]
            pre = [u for u in range(t) if tasks[u, 0] == PRED and tasks[u, 2] == c]
            assert len(pre) == 1 and tasks[pre[0], 4] == tasks[t, 4] and tasks[pre[0], 5] == c - 1


@pytest.mark.parametrize("n,e,seed", [(2560, 2, 0), (2560, 2, 1), (4096, 1, 2), (3200, 0, 3)])
def test_replay_in_random_valid_order(n, e, seed):
    counts, tasks, succ = _plan(n, e)
    nt = counts[0]
    T = n // 128
    rng = np.random.default_rng(seed)
    d = 6
    x = rng.standard_normal((n, d))
    g = x @ x.T
    dg = np.diag(g)
    sq = np.maximum(dg[:, None] + dg[None, :] - 2 * g, 0.0)
    K = np.exp(-0.5 * sq / d) + 1e-2 * np.eye(n)
    R = rng.standard_normal((e, n))
    A = np.vstack([np.tril(K), R])               # rows n.. = the extra rows
    W = np.zeros((T, 128, 128))
    scr_d = {}
    scr_s = {}
    rows = lambda i: slice(i * 128, (i + 1) * 128) if i < T else slice(n, n + e)
    cols = lambda j: slice(j * 128, (j + 1) * 128)
    dep = tasks[:, 6].astype(np.int64).copy()
    # an event = (task, part): part 0 runs the task (a step: only its solve) and releases phase 1, part 1 finishes a step
    ready = [(0, 0)]
    done = 0

    def release(lo, hi):
        for s in succ[lo:hi]:
            dep[s] -= 1
            assert dep[s] >= 0
            if dep[s] == 0:
                ready.append((int(s), 0))

    def leaf(i):
        blk = A[rows(i), cols(i)]
        L = np.linalg.cholesky(np.tril(blk) + np.tril(blk, -1).T)
        A[rows(i), cols(i)] = L
        W[i] = np.linalg.inv(L)

    while ready:
        t, part = ready.pop(int(rng.integers(len(ready))))
        ty, q, i, j, k0, k1, nd, fl, sb, sm, se = tasks[t, :11]
        ks = slice(k0 * 128, k1 * 128)
        if ty == LEAF:
            leaf(i)
        elif ty == TRSM:
            half = slice(64, 128) if fl & HALF1 else (slice(0, 64) if fl & HALF0 else slice(None))
            blk = A[rows(i), cols(j)]
            blk[half] = blk[half] @ W[j].T
            A[rows(i), cols(j)] = blk
        elif ty == PRED:
            scr_d[i] = A[rows(i), ks] @ A[rows(i), ks].T
        elif ty == UPD:
            acc = A[rows(i), ks] @ A[rows(j), ks].T
            if fl & ACC_OUT:
                assert j not in scr_s
                scr_s[j] = acc
            else:
                if i == j:
                    acc = np.tril(acc)
                A[rows(i), cols(j)] -= acc
        elif ty == SUB:                           # 64 rows of a tile (all rows of the extra-rows tile): a short-K update
            half = slice(64, 128) if fl & HALF1 else (slice(0, 64) if fl & HALF0 else slice(None))
            acc = A[rows(i), ks][half] @ A[rows(j), ks].T
            if fl & ACC_IN:
                assert k1 - k0 == 1
                acc = acc + scr_s[j][half]
            if i == j:
                r0 = 64 if fl & HALF1 else 0
                acc = np.tril(acc, k=r0)
            blk = A[rows(i), cols(j)]
            blk[half] -= acc
            A[rows(i), cols(j)] = blk
        elif ty == STEP and part == 0:
            c = i
            A[rows(c), cols(c - 1)] = A[rows(c), cols(c - 1)] @ W[c - 1].T
            release(sb, sm)
            ready.append((t, 1))
            continue
        else:                                     # the rest of a step: last block of the diagonal update, then the leaf
            c = i
            last = slice((c - 1) * 128, c * 128)
            acc = A[rows(c), last] @ A[rows(c), last].T
            if fl & ACC_IN:
                acc = acc + scr_d.pop(c)
            else:
                assert k0 == c - 1
            A[rows(c), cols(c)] -= np.tril(acc)
            leaf(c)
        done += 1
        release(sm, se)
    assert done == nt, "every task ran exactly once"
    Lref = np.linalg.cholesky(K)
    err = np.abs(np.tril(A[:n]) - Lref).max()
    assert err < 1e-9, err
    if e:
        want = np.linalg.solve(Lref, R.T).T       # R L^-T
        assert np.abs(A[n:] - want).max() < 1e-8

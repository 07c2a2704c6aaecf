#!/usr/bin/env python3
"""GPU-box tool (round 6): per-task timeline of ONE persistent factorisation (ppotrf.hip trace stamps, tools' build).
usage: persistent_trace.py <n> [chain_wgs] [dy]  ->  per task type: count, run time, pop-to-run latency; the critical chain step by
step; the busy fraction of chain / bulk workgroups."""
import ctypes, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), "..")))
from gptorch_amd import _native, _ops, rng
n = int(sys.argv[1]); R = int(sys.argv[2]) if len(sys.argv) > 2 else 0; dy = int(sys.argv[3]) if len(sys.argv) > 3 else 1
dev = torch.device("cuda:0")
lib = _native.debug_begin()
st, ptr = _ops._stream(dev), _ops._ptr
counts = (ctypes.c_int64 * 67)()
assert lib.gpn_potrf_persistent_plan(n, dy, counts, None, 0, None, 0) == 0
nt, ns = counts[0], counts[1]
tasks = np.zeros((nt, 12), dtype=np.int32); succ = np.zeros(ns, dtype=np.int32)
lib.gpn_potrf_persistent_plan(n, dy, counts, tasks.ctypes.data, nt, succ.ctypes.data, ns)
x, y = rng.make_regression(n, 8, dy, seed=0)
X, Y = torch.as_tensor(x).to(dev), torch.as_tensor(y).to(dev)
one = lambda v: torch.tensor([v], dtype=torch.float64, device=dev)
f = _ops.Factor(n, dy, dev)
_ops.kernel_matrix("Rbf", X, None, one(1.0), one(float(np.sqrt(8))), noise=one(1e-2), out=f.A, ldk=f.ld, lower=True)
f.pack_rhs(Y)
saved = f.A.clone()
lib.gpn_debug_set_persistent(R, 0)
trace = torch.zeros(nt * 8, dtype=torch.int64, device=dev)
for rep in range(3):
    f.A.copy_(saved); f.info.zero_()
    lib.gpn_debug_persistent_trace(trace.data_ptr() if rep == 2 else None)
    assert lib.gpn_potrf_lower_persistent(st, ptr(f.A), n, dy, f.ld, ptr(f.winv), ptr(f.info)) == 0
torch.cuda.synchronize()
lib.gpn_debug_persistent_trace(None)
tr = trace.cpu().numpy().reshape(nt, 8).astype(np.int64)
t0 = tr[:, 1].min()
us = lambda v: (v - t0) / 100.0
pop0, pop1, run, end, rel = (us(tr[:, k]) for k in range(5))
wg = tr[:, 5] & 0xffffffff; chain = tr[:, 5] >> 32
print("n %d: %d tasks (queues %s), span %.1f us, info %d" % (n, nt, " / ".join(str(c) for c in counts[3:3 + counts[2]]), rel.max(), int(f.info.item())))
names = {0: "LEAF", 1: "TRSM", 2: "UPD", 3: "STEP", 4: "PRED", 5: "SUB"}
for ty in (0, 3, 4, 5, 1, 2):
    m = tasks[:, 0] == ty
    for q in range(int(counts[2])):
        mm = m & (tasks[:, 1] == q)
        if not mm.any(): continue
        Ks = sorted(set((tasks[mm, 5] - tasks[mm, 4] + 1000 * tasks[mm, 7]).tolist())) if ty in (2, 3, 4, 5) else [0]
        for K in Ks:
            m3 = mm & ((tasks[:, 5] - tasks[:, 4] + 1000 * tasks[:, 7]) == K) if ty in (2, 3, 4, 5) else mm
            print("  %-4s q%d K=%6d: %5d tasks  run %7.1f us avg (min %6.1f max %7.1f)  acquire %4.1f  release %4.1f  wait-in-pop %7.1f" % (
                names[ty], q, (K % 1000) * 128 + 100000 * (K // 1000), m3.sum(), (end - run)[m3].mean(), (end - run)[m3].min(), (end - run)[m3].max(),
                (run - pop1)[m3].mean(), (rel - end)[m3].mean(), (pop1 - pop0)[m3].mean()))
# the critical chain: leaf k -> solve (k+1, k) -> last update of (k+1, k+1) -> leaf k+1
T = n // 128
leaf = {int(tasks[t, 2]): t for t in range(nt) if tasks[t, 0] in (0, 3)}
print("  chain: leaf k: [popped .. released]; then when leaf k+1 was popped")
for k in list(range(0, min(T - 1, 12))) + list(range(max(12, T - 4), T - 1)):
    a, b = leaf[k], leaf[k + 1]
    print("   k %3d  leaf run %.1f..%.1f released %.1f | next leaf popped %.1f (+%.1f)  step %.1f us" % (
        k, run[a], end[a], rel[a], pop1[b], pop1[b] - rel[a], rel[b] - rel[a]))
steps = np.array([rel[leaf[k + 1]] - rel[leaf[k]] for k in range(T - 1)])
print("  chain step: mean %.1f us, median %.1f, max %.1f; sum %.1f of span %.1f" % (steps.mean(), np.median(steps), steps.max(), steps.sum(), rel.max()))
# ---- critical paths behind the largest waits of the chain
ph1 = np.where(tr[:, 6] > 0, us(tr[:, 6]), rel)
preds = [[] for _ in range(nt)]
for t in range(nt):
    b, m, e_ = tasks[t, 8:11]
    for s_ in succ[b:m]: preds[s_].append((t, 1))
    for s_ in succ[m:e_]: preds[s_].append((t, 2))
def when(t, phase): return ph1[t] if (phase == 1 and tasks[t, 0] == 3) else rel[t]
def label(t):
    ty, q, i, j, k0, k1, nd, fl = tasks[t, :8]
    return "%-4s(%2d,%2d) K=%4d fl%d q%d" % (names[ty], i, j, (k1 - k0) * 128, fl, q)
steps_t = [leaf[k] for k in range(1, T)]
gaps = sorted(((pop1[t] - max(when(p, ph) for p, ph in preds[t]) if preds[t] else 0.0, t) for t in steps_t), reverse=True)
waits = sorted(((pop1[leaf[k + 1]] - rel[leaf[k]], k) for k in range(T - 1)), reverse=True)[:int(os.environ.get('PP_PATHS', '3'))]
for wgap, k in waits:
    t = leaf[k + 1]
    print("  --- step %d waited %.1f us after step %d's end; critical path backwards:" % (k + 1, wgap, k))
    for depth in range(int(os.environ.get('PP_DEPTH', '9'))):
        if not preds[t]: break
        p, ph = max(preds[t], key=lambda pp: when(pp[0], pp[1]))
        ready = when(p, ph)
        print("      %s ready %.1f popped %.1f (queue %.1f) run %.1f..%.1f released %.1f   <- %s (ph %d) at %.1f" % (
            label(t), max(when(x, y) for x, y in preds[t]), pop1[t], pop1[t] - max(when(x, y) for x, y in preds[t]), run[t], end[t], rel[t], label(p), ph, ready))
        t = p
for role in (1, 0):
    m = chain == role
    wgs = np.unique(wg[m])
    busy = (rel - pop1)[m].sum()
    print("  %s workgroups: %d seen, busy %.1f us total = %.1f %% of %d x span" % ("chain" if role else "bulk ", len(wgs), busy, 100 * busy / (len(wgs) * rel.max()), len(wgs)))
_native.debug_end()

#!/usr/bin/env python3
"""GPU-box experiment: a "diagonal block first" factorisation assembled from the library's own entry points.

The shipped driver (potrf.hip potrf_lookahead) runs the serial chain leaf -> solve -> next-column update over ALL rows below
a 128-column block, 16 times per 2048-column panel.  Here a panel is
    (1) factor its pw x pw diagonal block alone          (the only latency-bound part: 16 short steps on <= 2048 rows)
    (2) W = L_pp^-1 explicitly (level-parallel inversion + transpose)
    (3) X = B W^T for all rows below as ONE K-clipped contraction into a side buffer (+ copy back into the factor)
    (4) trailing update from the side buffer
MODE=serial : everything on one stream (prices the phases)
MODE=overlap: look-ahead over panels -- after (3) only the next panel's diagonal square is updated on the main stream, the
              rest of the trailing update runs on a second, lowest-priority stream (PAD KiB of LDS padding per workgroup =
              occupancy cap) underneath the next panel's (1)-(3).
Prints ms per factorisation against the shipped driver on the same matrix, and the distance of sum log L_ii / |alpha|^2."""
import os, sys, time, ctypes
ROOT = os.environ.get("GRAFT_REPO_ROOT", os.path.abspath(os.path.join(os.path.dirname(__file__), "..")))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
import bench  # noqa: E402
from gptorch_amd import _native, _ops  # noqa: E402
from gptorch_amd._ops import round_up  # noqa: E402

dev = torch.device("cuda:0")
WL = os.environ.get("WORKLOAD", "c3")
PW = int(os.environ.get("PW", "2048"))
MODES = os.environ.get("MODE", "serial,overlap").split(",")
PAD = int(os.environ.get("PAD", "32"))
REPS = int(os.environ.get("REPS", "3"))

m, _, _ = bench.build_model(bench.WORKLOADS[WL], 0, dev)
lib = _native.debug_begin()
k = m.kernel
with torch.no_grad():
    var, ls, nz = k.variance.transform(), k.length_scales.transform(), m.likelihood.variance.transform()
    R = (m.Y - m.mean_function(m.X)).contiguous()
n, e = R.shape
f = _ops.Factor(n, e, dev)
ld = f.ld
vp = ctypes.c_void_p


def assemble():
    _ops.kernel_matrix(k._kind, m.X, None, var, ls, noise=nz, out=f.A, ldk=ld, lower=True)
    f.pack_rhs(R)
    f.info.zero_()


def timed(fn):
    best = 1e30
    for _ in range(REPS):
        with torch.cuda.stream(main):
            assemble()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            fn()
            torch.cuda.synchronize()
        best = min(best, time.perf_counter() - t0)
    return best * 1e3


def shipped():
    f.potrf(check=False)


main = torch.cuda.Stream(device=dev, priority=-1)     # created streams on both sides (waits against the null stream are expensive)
aux = torch.cuda.Stream(device=dev, priority=0)
ms_ref = timed(shipped)
with torch.cuda.stream(main):
    terms_ref = f.lml_terms().clone()
torch.cuda.synchronize()
print("%s n=%d  shipped driver %.2f ms   logdet/2 %.12e  |alpha|^2 %.12e" % (WL, n, ms_ref, terms_ref[0].item(), terms_ref[1].item()), flush=True)

rows_x = round_up(n + e, 128)
Xb = [torch.empty(rows_x, PW, dtype=torch.float64, device=dev) for _ in range(2)]
U = _ops.zeros(PW, PW, dev)
S = _ops.zeros(PW, PW, dev)
Wt = _ops.zeros(PW, PW, dev)
A0 = f.A.data_ptr()
W0 = f.winv.data_ptr()


def ck(st, what):
    if st != 0:
        raise RuntimeError("%s -> %d" % (what, st))


def sp(stream):
    return vp(stream.cuda_stream)


def panel_diag(stream, p0, pw):
    """(1)-(2): factor the diagonal square of panel p0, invert it into Wt."""
    D = vp(A0 + (p0 * ld + p0) * 8)
    Wp = vp(W0 + (p0 // 128) * 128 * 128 * 8)
    ck(lib.gpn_potrf_lower(sp(stream), D, pw, 0, ld, Wp, vp(f.info.data_ptr())), "potrf")
    if n + e - (p0 + pw) <= 0:
        return
    ck(lib.gpn_trtri_upper_ws(sp(stream), D, pw, ld, Wp, vp(U.data_ptr()), PW, vp(S.data_ptr()), PW), "trtri")
    ck(lib.gpn_transpose(sp(stream), vp(U.data_ptr()), pw, pw, PW, vp(Wt.data_ptr()), PW), "transpose")


def panel_solve(stream, p0, pw, X):
    """(3): X = B W^T for the rows below, copied back into the factor; returns the number of those rows."""
    pend = p0 + pw
    mrows = n + e - pend
    if mrows <= 0:
        return 0
    B = vp(A0 + (pend * ld + p0) * 8)
    ck(lib.gpn_gemm_nt(sp(stream), mrows, pw, round_up(pw, 16), 1.0, B, ld, vp(Wt.data_ptr()), PW, 0.0, vp(X.data_ptr()), PW, 0,
                       _ops.TRI_B_LOWER), "trsm-gemm")
    ck(lib.gpn_copy_matrix(sp(stream), vp(X.data_ptr()), mrows, pw, PW, B, ld, 0), "copy")
    return mrows


PHASES = {}


def diagfirst_serial():
    marks = []
    with torch.cuda.stream(main):
        def mark(name):
            ev = torch.cuda.Event(enable_timing=True)
            ev.record(main)
            marks.append((name, ev))
        mark("start")
        for i, p0 in enumerate(range(0, n, PW)):
            pw = min(PW, n - p0)
            pend = p0 + pw
            X = Xb[i & 1]
            D = vp(A0 + (p0 * ld + p0) * 8)
            Wp = vp(W0 + (p0 // 128) * 128 * 128 * 8)
            ck(lib.gpn_potrf_lower(sp(main), D, pw, 0, ld, Wp, vp(f.info.data_ptr())), "potrf")
            mark("diag potrf")
            if n + e - pend > 0:
                ck(lib.gpn_trtri_upper_ws(sp(main), D, pw, ld, Wp, vp(U.data_ptr()), PW, vp(S.data_ptr()), PW), "trtri")
                ck(lib.gpn_transpose(sp(main), vp(U.data_ptr()), pw, pw, PW, vp(Wt.data_ptr()), PW), "transpose")
            mark("diag inverse")
            mrows = panel_solve(main, p0, pw, X)
            mark("solve + copy")
            if mrows <= 0:
                break
            C = vp(A0 + (pend * ld + pend) * 8)
            ck(lib.gpn_gemm_nt(sp(main), mrows, mrows, round_up(pw, 16), -1.0, vp(X.data_ptr()), PW, vp(X.data_ptr()), PW, 1.0, C, ld, 1, 0), "syrk")
            mark("trailing update")
    torch.cuda.synchronize()
    PHASES.clear()
    for (_, e0), (name, e1) in zip(marks[:-1], marks[1:]):
        PHASES[name] = PHASES.get(name, 0.0) + e0.elapsed_time(e1)


def diagfirst_overlap():
    ev_front = [torch.cuda.Event() for _ in range(2)]
    ev_col = ev_bulk = None
    for i, p0 in enumerate(range(0, n, PW)):
        pw = min(PW, n - p0)
        pend = p0 + pw
        X = Xb[i & 1]
        panel_diag(main, p0, pw)             # its square was completed on this stream (diag strip of the panel before)
        if ev_col is not None:
            main.wait_event(ev_col)          # the rows under it: column strip of the panel before, on aux
        mrows = panel_solve(main, p0, pw, X)
        if mrows <= 0:
            break
        ev_front[i & 1].record(main)
        pw2 = min(PW, n - pend)
        Xp = X.data_ptr()
        C = A0 + (pend * ld + pend) * 8
        if ev_bulk is not None:
            main.wait_event(ev_bulk)         # bulk(i-1) wrote the region the next strip accumulates into
        if pw2 <= 0 or mrows - pw2 <= 0:
            ck(lib.gpn_gemm_nt(sp(main), mrows, mrows, round_up(pw, 16), -1.0, vp(Xp), PW, vp(Xp), PW, 1.0, vp(C), ld, 1, 0), "syrk-last")
            ev_col = ev_bulk = None
            continue
        # next panel's diagonal square on main
        ck(lib.gpn_gemm_nt(sp(main), pw2, pw2, round_up(pw, 16), -1.0, vp(Xp), PW, vp(Xp), PW, 1.0, vp(C), ld, 1, 0), "diag-strip")
        # everything else on aux: the column strip under that square, then the lower square right of it
        aux.wait_event(ev_front[i & 1])
        m2 = mrows - pw2
        X2 = Xp + pw2 * PW * 8
        lib.gpn_debug_set_gemm_variant(PAD << 8)
        ck(lib.gpn_gemm_nt(sp(aux), m2, pw2, round_up(pw, 16), -1.0, vp(X2), PW, vp(Xp), PW, 1.0, vp(C + pw2 * ld * 8), ld, 0, 0), "col-strip")
        ev_col = torch.cuda.Event()
        ev_col.record(aux)
        ck(lib.gpn_gemm_nt(sp(aux), m2, m2, round_up(pw, 16), -1.0, vp(X2), PW, vp(X2), PW, 1.0, vp(C + (pw2 * ld + pw2) * 8), ld, 1, 0), "bulk")
        lib.gpn_debug_set_gemm_variant(0)
        ev_bulk = torch.cuda.Event()
        ev_bulk.record(aux)
    if ev_bulk is not None:
        main.wait_event(ev_bulk)


for mode in MODES:
    fn = {"serial": diagfirst_serial, "overlap": diagfirst_overlap}[mode]
    ms = timed(fn)
    with torch.cuda.stream(main):
        t = f.lml_terms()
    torch.cuda.synchronize()
    print("diag-first %-8s PW=%d PAD=%d: %.2f ms (shipped %.2f)   d(logdet/2) %.3e   d|alpha|^2 %.3e   info %d" % (
        mode, PW, PAD, ms, ms_ref, (t[0] - terms_ref[0]).item(), (t[1] - terms_ref[1]).item(), int(f.info.item())), flush=True)
    if mode == "serial":
        print("   phases (ms, last repetition): " + ", ".join("%s %.2f" % kv for kv in PHASES.items()), flush=True)
_native.debug_end()

"""
Host-side orchestration over the C ABI (include/gpnative.h): turns torch CUDA
(HIP) tensors into device pointers + the current HIP stream, owns the factor
buffers, and exposes the differentiable log-marginal-likelihood.

Everything here requires fp64 tensors resident on a HIP device.  There is no
CPU code path: a CPU tensor raises `NativeError`.
"""
import ctypes
import math

import torch

from . import _native
from ._native import NativeError

KINDS = {"Rbf": 0, "Matern52": 1, "Matern32": 2, "Exp": 3, "Matern12": 3, "SqDist": 4, "Periodic": 5}
GPN_FULL, GPN_LOWER = 0, 1
LEAF = 128   # leaf block of the factorisation = padding granule of factor buffers (gpn_common.h)


def _ptr(t):
    return None if t is None else ctypes.c_void_p(t.data_ptr())


def _stream(device):
    return ctypes.c_void_p(torch.cuda.current_stream(device).cuda_stream)


def _req(*tensors):
    for t in tensors:
        if t is None:
            continue
        if not t.is_cuda:
            raise NativeError(
                "gptorch_amd runs on an AMD GPU only (no CPU fallback): got a tensor on %s; "
                "move the model/data with .cuda() first" % t.device)
        if t.dtype != torch.float64:
            raise TypeError("gptorch_amd computes in fp64 (gptorch TensorType); got %s" % t.dtype)


def _c(t):
    return t if t.is_contiguous() else t.contiguous()


def round_up(x, m):
    return (x + m - 1) // m * m


# ----------------------------------------------------------------------------
# K assembly
# ----------------------------------------------------------------------------
def kernel_matrix(kind, X, X2, variance, length_scales, noise=None, out=None, ldk=None, lower=False):
    """K(X, X2) as a new [n, m] tensor, or written into `out` (leading dim ldk)."""
    _req(X, X2, variance, length_scales, noise)
    X = _c(X.detach())
    n, d = X.shape
    if X2 is not None:
        X2 = _c(X2.detach())
        m = X2.shape[0]
        if X2.shape[1] != d:
            raise ValueError("X and X2 must have the same input dimension")
    else:
        m = n
    if out is None:
        out = torch.empty(n, m, dtype=torch.float64, device=X.device)
        ldk = m
    variance, length_scales = _c(variance.detach()), _c(length_scales.detach())
    noise = None if noise is None else _c(noise.detach())
    st = _native.lib().gpn_kernel_matrix(
        _stream(X.device), KINDS[kind], _ptr(X), n, _ptr(X2), m, d, _ptr(variance), _ptr(length_scales),
        length_scales.numel(), _ptr(noise), GPN_LOWER if lower else GPN_FULL, _ptr(out), ldk)
    _native.check(st, "gpn_kernel_matrix")
    return out


def zeros(rows, cols, device):
    """a zeroed [rows, cols] fp64 buffer: torch owns the memory, the library clears it (hipMemsetAsync on the current
    stream) -- the large factor / inverse buffers are not cleared by an elementwise torch kernel."""
    t = torch.empty(rows, cols, dtype=torch.float64, device=device)
    st = _native.lib().gpn_fill_zero(_stream(t.device), _ptr(t), t.numel() * 8)
    _native.check(st, "gpn_fill_zero")
    return t


# ----------------------------------------------------------------------------
# factor buffers + Cholesky
# ----------------------------------------------------------------------------
class Factor:
    """A factor buffer (see gpnative.h): n x n lower Cholesky factor in the
    top-left corner of `A` (rows x ld, zero padded), `e` extra rows holding
    (L^-1 R)^T, and the inverses of the 128x128 diagonal leaf blocks in `winv`."""

    def __init__(self, n, e, device):
        lib = _native.lib()
        self.n, self.e = int(n), int(e)
        self.ld = int(lib.gpn_factor_ld(n, e))
        self.rows = int(lib.gpn_factor_rows(n, e))
        self.A = zeros(self.rows, self.ld, device)
        self.winv = torch.empty(max(1, int(lib.gpn_winv_bytes(n)) // 8), dtype=torch.float64, device=device)
        self.info = torch.zeros(1, dtype=torch.int32, device=device)
        self.jitter_rung = -1
        self.generation = 0          # bumped by every factorisation into this buffer
        self._winv_full = None       # explicit L^-1 (lower_inverse), valid for one generation
        self._refine_work = None     # workspace of gpn_lml_refine (allocated on first use)
        self.refined = False

    @property
    def device(self):
        return self.A.device

    def extra(self):
        """[e, n] view: (L^-1 R)^T after factorisation."""
        return self.A[self.n:self.n + self.e, :self.n]

    def lower(self):
        """n x n copy of L with a zeroed upper triangle (torch.cholesky's output)."""
        out = torch.empty(self.n, self.n, dtype=torch.float64, device=self.device)
        if self.n:
            st = _native.lib().gpn_copy_matrix(_stream(self.device), _ptr(self.A), self.n, self.n, self.ld,
                                               _ptr(out), self.n, 1)
            _native.check(st, "gpn_copy_matrix")
        return out

    def pack_rhs(self, R, M=None):
        """extra rows <- (R - M)^T, R [n, e]."""
        _req(R, M)
        R = _c(R.detach())
        M = None if M is None else _c(M.detach())
        if self.e:
            self.A[self.n:self.n + self.e, self.n:].zero_()   # corner accumulates -alpha alpha^T
            st = _native.lib().gpn_pack_rhs(_stream(self.device), _ptr(R), _ptr(M), self.n, self.e,
                                            _ptr(self.A[self.n]), self.ld)
            _native.check(st, "gpn_pack_rhs")

    def potrf(self, check=True):
        """In-place factorisation; returns the LAPACK-style info (host int; syncs) or,
        with check=False, enqueues only and returns None (read self.info later)."""
        self.info.zero_()
        self.generation += 1
        self._winv_full = None
        st = _native.lib().gpn_potrf_lower(_stream(self.device), _ptr(self.A), self.n, self.e, self.ld,
                                           _ptr(self.winv), _ptr(self.info))
        _native.check(st, "gpn_potrf_lower")
        return int(self.info.item()) if check else None

    def lml_terms(self):
        """tensor [3]: sum log L_ii, ||extra||_F^2, LML (gpr.py:63-67)."""
        out = torch.empty(3, dtype=torch.float64, device=self.device)
        st = _native.lib().gpn_lml_reduce(_stream(self.device), _ptr(self.A), self.n, self.e, self.ld, _ptr(out))
        _native.check(st, "gpn_lml_reduce")
        return out

    def solve_right_lt(self, B, m):
        """B[m, :n] <- B * L^-T in place; B must be a [rows >= round_up(m,16), ld] buffer
        with ld == self.ld whose padding is zero."""
        st = _native.lib().gpn_trsm_right_lt(_stream(self.device), _ptr(self.A), self.n, self.ld, _ptr(self.winv),
                                             _ptr(B), m, B.stride(0))
        _native.check(st, "gpn_trsm_right_lt")
        return B


class FactorBatch:
    """`batch` factor buffers of one shape in ONE allocation (problem b at stride sA / sW), for
    gpn_lml_forward_batched / gpn_potrf_lower_batched; factor(b) is a Factor VIEW of problem b (same memory),
    usable with every single-model entry point afterwards (predict, backward ...)."""

    def __init__(self, batch, n, e, device):
        lib = _native.lib()
        self.batch, self.n, self.e = int(batch), int(n), int(e)
        self.ld = int(lib.gpn_factor_ld(n, e))
        self.rows = int(lib.gpn_factor_rows(n, e))
        self.sA = self.rows * self.ld
        self.sW = max(2, int(lib.gpn_winv_bytes(n)) // 8)
        self.A = zeros(self.batch * self.rows, self.ld, device)
        self.winv = torch.empty(self.batch * self.sW, dtype=torch.float64, device=device)
        self.info = torch.zeros(self.batch, dtype=torch.int32, device=device)
        self.out = torch.empty(self.batch, 3, dtype=torch.float64, device=device)
        self.generation = 0          # bumped by every lml_forward_batched into these buffers
        self._backward_work = None   # workspace of gpn_lml_backward_batched (allocated on first use, reused by a fit loop)

    def nbytes(self):
        w = 0 if self._backward_work is None else self._backward_work.numel()
        r = getattr(self, "_refine_work", None)
        w += 0 if r is None else r.numel()
        return 8 * (self.A.numel() + self.winv.numel() + w)

    def factor(self, b):
        f = Factor.__new__(Factor)
        f.n, f.e, f.ld, f.rows = self.n, self.e, self.ld, self.rows
        f.A = self.A[b * self.rows:(b + 1) * self.rows]
        f.winv = self.winv[b * self.sW:(b + 1) * self.sW]
        f.info = self.info[b:b + 1]
        f.jitter_rung = -1
        f.generation = 0
        f._winv_full = None
        f._refine_work = None
        f.refined = False
        return f


def lml_forward_batched(kind, X, R, variance, length_scales, noise, fb=None, refine=False, n_of=None):
    """GPR.log_likelihood (gpr.py:47-67) of `batch` models in lock step (gpn_lml_forward_batched): X [n, d] shared or
    [batch, n, d]; R = Y - m(X) [n, dy] shared or [batch, n, dy]; variance [batch], length_scales [batch, nls],
    noise [batch].  -> (FactorBatch, terms [batch, 3]); NO host synchronisation: the caller reads fb.info and
    replays the models whose info != 0 through the sequential path (jitter ladder).
    refine: follow the lock-step factorisation with gpn_lml_refine on every model's factor (what lml_forward does from
    refine_min_n() rows on: the same call on the same factor, so the refined terms are bit-identical too).
    n_of (int32 device tensor [batch]): a RAGGED batch (gpn_lml_forward_ragged) -- model b has n_of[b] <= n points, X [batch, n, d] and
    R [batch, n, dy] are padded to n rows (the padding is not read into any result); no refinement."""
    _req(X, R, variance, length_scales, noise)
    batch = int(variance.numel())
    shared_x, shared_r = X.dim() == 2, R.dim() == 2
    n, d = X.shape[-2], X.shape[-1]
    e = R.shape[-1]
    if R.shape[-2] != n:
        raise ValueError("X and Y must have same # data.")
    if fb is None or fb.batch != batch or fb.n != n or fb.e != e or fb.A.device != X.device:
        fb = FactorBatch(batch, n, e, X.device)
    Xc, Rc = _c(X.detach()), _c(R.detach())
    var, ls, nz = _c(variance.detach().reshape(batch)), _c(length_scales.detach().reshape(batch, -1)), _c(noise.detach().reshape(batch))
    fb.generation += 1
    if n_of is not None:
        if shared_x or shared_r or refine:
            raise ValueError("ragged batches hold every model's own (padded) data and are not refined")
        st = _native.lib().gpn_lml_forward_ragged(
            _stream(X.device), KINDS[kind], batch, _ptr(Xc), n * d, n, _ptr(n_of), d, _ptr(Rc), n * e, e, _ptr(var), _ptr(ls), ls.shape[1],
            _ptr(nz), _ptr(fb.A), fb.ld, fb.sA, _ptr(fb.winv), fb.sW, _ptr(fb.info), _ptr(fb.out))
        _native.check(st, "gpn_lml_forward_ragged")
        return fb, fb.out
    st = _native.lib().gpn_lml_forward_batched(
        _stream(X.device), KINDS[kind], batch, _ptr(Xc), 0 if shared_x else n * d, n, d, _ptr(Rc), 0 if shared_r else n * e,
        None, 0, e, _ptr(var), _ptr(ls), ls.shape[1], _ptr(nz), _ptr(fb.A), fb.ld, fb.sA, _ptr(fb.winv), fb.sW,
        _ptr(fb.info), _ptr(fb.out))
    _native.check(st, "gpn_lml_forward_batched")
    if refine and n > 0:
        lib = _native.lib()
        if getattr(fb, "_refine_work", None) is None:
            fb._refine_work = torch.empty(max(1, int(lib.gpn_lml_refine_work_bytes(n, e)) // 8), dtype=torch.float64, device=X.device)
        for b in range(batch):                       # (3 % of an evaluation each; enqueued before anybody reads `info`)
            Xb = Xc if shared_x else Xc[b]
            Rb = Rc if shared_r else Rc[b]
            st = lib.gpn_lml_refine(_stream(X.device), KINDS[kind], _ptr(Xb), n, d, _ptr(Rb), None, e, _ptr(var[b:b + 1]), _ptr(ls[b]), ls.shape[1],
                                    _ptr(nz[b:b + 1]), _ptr(fb.A[b * fb.rows:]), fb.ld, _ptr(fb.winv[b * fb.sW:]), _ptr(fb._refine_work),
                                    _ptr(fb.out[b]))
            _native.check(st, "gpn_lml_refine")
    return fb, fb.out


def lml_backward_batched(kind, X, variance, length_scales, fb, need_resid=False, n_of=None):
    """the closed-form backward of the `batch` models of an lml_forward_batched call in lock step (gpn_lml_backward_batched):
    -> (grads [batch, 2 + nls] = dLML/d(variance, length_scales, noise) per model w.r.t. the CONSTRAINED values,
        dLML/dR [batch, n, dy] or None).  Per model bit-identical to _backward.lml_backward on that model's factor."""
    _req(X, variance, length_scales)
    batch, n, dy = fb.batch, fb.n, fb.e
    ls = _c(length_scales.detach().reshape(batch, -1))
    nls = ls.shape[1]
    lib = _native.lib()
    need = max(1, int(lib.gpn_lml_backward_batched_work_bytes(n, dy, nls, batch)) // 8)
    if fb._backward_work is None or fb._backward_work.numel() < need:
        fb._backward_work = torch.empty(need, dtype=torch.float64, device=fb.A.device)
    out = torch.empty(batch, 2 + nls, dtype=torch.float64, device=fb.A.device)
    g_R = torch.empty(batch, n, dy, dtype=torch.float64, device=fb.A.device) if need_resid else None
    Xc = _c(X.detach())
    if n_of is not None:                              # the backward of a ragged batch (gpn_lml_backward_ragged): no dLML/dR
        if need_resid or Xc.dim() != 3:
            raise ValueError("ragged batches: every model's own padded points, zero mean functions")
        st = lib.gpn_lml_backward_ragged(_stream(fb.A.device), KINDS[kind], batch, _ptr(Xc), n * Xc.shape[-1], n, _ptr(n_of), Xc.shape[-1],
                                         _ptr(_c(variance.detach().reshape(batch))), _ptr(ls), nls, _ptr(fb.A), fb.ld, fb.sA,
                                         _ptr(fb.winv), fb.sW, dy, _ptr(fb._backward_work), _ptr(out))
        _native.check(st, "gpn_lml_backward_ragged")
        return out, None
    st = lib.gpn_lml_backward_batched(_stream(fb.A.device), KINDS[kind], batch, _ptr(Xc), 0 if Xc.dim() == 2 else n * Xc.shape[-1], n,
                                      Xc.shape[-1], _ptr(_c(variance.detach().reshape(batch))), _ptr(ls), nls, _ptr(fb.A), fb.ld, fb.sA,
                                      _ptr(fb.winv), fb.sW, dy, _ptr(fb._backward_work), _ptr(out), _ptr(g_R))
    _native.check(st, "gpn_lml_backward_batched")
    return out, g_R


REFINE_MIN_N = 12288
JITTER_TRIES = 10  # functions.py:21 max_tries


# ---- evaluations without a host read-back (hipGraph capture of an optimiser step: models/base.py _optimize_captured) --------
# While a DeferredInfo is active, lml_forward enqueues ONE attempt (no jitter), leaves the factorisation's `info` word on the
# device and ORs "info != 0" into the holder's flag instead of reading it back: nothing in the evaluation synchronises the host,
# so evaluation + closed-form backward + optimiser step capture into one graph.  The caller looks at the flag every so many
# replays and repeats a chunk that saw a failure through the ordinary path (jitter ladder of functions.py:20-43).
_DEFERRED = []


class DeferredInfo:
    def __init__(self, device):
        self.flag = torch.zeros(1, dtype=torch.int32, device=device)

    def __enter__(self):
        _DEFERRED.append(self)
        return self

    def __exit__(self, *exc):
        _DEFERRED.pop()
        return False

    def note(self, info):
        torch.maximum(self.flag, (info != 0).to(torch.int32), out=self.flag)


def _ladder(attempt, tries=None):
    """functions.py:20-43: plain try, then +10^(-max_tries+i) I for i = 0..max_tries-1 (max_tries = 10 by default), then
    RuntimeError("Max tries exceeded.").  `attempt(jitter)` -> info."""
    tries = JITTER_TRIES if tries is None else int(tries)
    def run(jitter):
        info = attempt(jitter)
        if info < 0:
            # GPN_INFO_INTERNAL: a hand-over inside the leaf kernel failed -- says nothing about the
            # matrix, so adding jitter would only hide it
            raise NativeError("factorisation reported the internal status %d (not a property of the matrix)" % info)
        return info

    if run(None) == 0:
        return -1
    for i in range(tries):
        if run(10.0 ** (-tries + i)) == 0:
            return i
    raise RuntimeError("Max tries exceeded.")


def cholesky_factor(x, rhs=None):
    """functions.cholesky (functions.py:46-47) of a dense SPD matrix `x` [n, n];
    optional rhs [n, k] is forward-substituted on the way (extra rows)."""
    _req(x, rhs)
    if x.dim() != 2 or x.shape[0] != x.shape[1]:
        raise RuntimeError("cholesky: expected a square matrix")
    n = x.shape[0]
    e = 0 if rhs is None else rhs.shape[1]
    f = Factor(n, e, x.device)
    xs = _c(x.detach())

    def attempt(jitter):
        if n:
            st = _native.lib().gpn_copy_matrix(_stream(x.device), _ptr(xs), n, n, n, _ptr(f.A), f.ld, 1)
            _native.check(st, "gpn_copy_matrix")
            if jitter is not None:
                f.A.diagonal()[:n].add_(jitter)
        if e:
            f.pack_rhs(rhs)
        return f.potrf()

    f.jitter_rung = _ladder(attempt)
    return f


def kernel_factor_async(kind, X, variance, length_scales, noise, R=None, factor=None):
    """kernel_factor without the host read-back of `info`: everything is enqueued on the
    current stream and the caller inspects f.info afterwards (a non-zero info must be
    replayed through kernel_factor for the jitter ladder).  Lets several independent
    models (hyper-parameter restarts) be in flight on different HIP streams."""
    _req(X, variance, length_scales, noise, R)
    n = X.shape[0]
    e = 0 if R is None else R.shape[1]
    f = factor if (factor is not None and factor.n == n and factor.e == e and factor.device == X.device) \
        else Factor(n, e, X.device)
    kernel_matrix(kind, X, None, variance, length_scales, noise=noise, out=f.A, ldk=f.ld, lower=True)
    if e:
        f.pack_rhs(R)
    f.potrf(check=False)
    f.jitter_rung = -1
    return f


def kernel_factor(kind, X, variance, length_scales, noise, R=None, factor=None):
    """Fused K(X)+noise*I assembly -> Cholesky (+ forward substitution of R).
    gpr.py:61-62 / 104-106.  Reuses `factor` (same n, e) when given."""
    _req(X, variance, length_scales, noise, R)
    n = X.shape[0]
    e = 0 if R is None else R.shape[1]
    f = factor if (factor is not None and factor.n == n and factor.e == e and factor.device == X.device) \
        else Factor(n, e, X.device)

    def attempt(jitter):
        if jitter is None:
            nz = noise
        elif noise is None:
            nz = torch.full((1,), jitter, dtype=torch.float64, device=X.device)
        else:
            nz = noise + jitter
        kernel_matrix(kind, X, None, variance, length_scales, noise=nz, out=f.A, ldk=f.ld, lower=True)
        if e:
            f.pack_rhs(R)
        return f.potrf()

    f.jitter_rung = _ladder(attempt)
    return f


def refine_min_n(expression=False, grid=False):
    """from this many rows on, lml_forward follows the factorisation with one refinement step of the quadratic form
    (gpn_lml_refine).  The plain value's distance to the exact one grows like N^1.85 (5.8e-10 at N = 8192, 7.6e-9 at
    32768, measured) and north_star's tolerance is 1e-8 ABSOLUTE against a reference that is itself 3.4e-9 off at
    32768: below about 10^4 rows the step buys nothing, above it costs about 3 %.  GPN_REFINE_MIN_N overrides
    (0 = never) and is taken VERBATIM for every caller: the 1/2 and 2/3 factors below scale the built-in default only.
    expression=True: covariance expressions (Linear / Constant terms grow the top eigenvalue like N |x|^2, so the
    quadratic form's sensitivity to the factor's rounding is an order of magnitude above a stationary kernel's: the
    reference's example model at N = 8192 sits 2e-8 from its golden unrefined, whichever leaf kernel factors it) refine
    from half that size on.
    grid=True: the block-cyclic drivers (their tile-wide trailing updates accumulate over K = 1024..2048 per launch, a
    different rounding profile from the single-GPU panels: C2's matrix on a 1 x 1 grid of 2048-wide tiles sits 0.9-1.1e-8
    from the golden unrefined) refine from two thirds of that size on (8192 rows)."""
    import os
    env = os.environ.get("GPN_REFINE_MIN_N")
    if env is not None:
        v = int(env)
    elif expression:
        v = REFINE_MIN_N // 2
    elif grid:
        v = (2 * REFINE_MIN_N) // 3
    else:
        v = REFINE_MIN_N
    return v if v > 0 else 1 << 62


def lml_forward(kind, X, R, variance, length_scales, noise, factor=None, refine=None):
    """GPR.log_likelihood (gpr.py:47-67) as ONE library call (gpn_lml_forward: assembly ->
    factorisation with the residual riding along -> reductions) plus the jitter ladder of
    functions.py:20-43 on its info word.  -> (Factor, terms [3]: sum log L_ii, |alpha|^2, LML).
    refine (default: N >= refine_min_n()): follow it with gpn_lml_refine, after which terms[1] is
    y^T Kyy^-1 y corrected to second order in the factor's rounding error (and terms[2] the LML with it).
    GPN_REFINE_SAVED_K=1 (opt-in): the assembly also keeps a pristine copy of Kyy (gpn_lml_forward_saving: n * ld * 8 more
    bytes) that the refinement's residual pass reads back instead of re-computing every entry -- the same refined value bit
    for bit; C3: refinement 3.3 -> 2.1 ms, assembly 1.8 -> 2.6 ms (twice the writes), evaluation 183.3 -> 182.8 ms.  Off by
    default: 0.25 % for 8.6 GB."""
    _req(X, R, variance, length_scales, noise)
    n, e = R.shape
    if X.shape[0] != n:
        raise ValueError("X and Y must have same # data.")
    f = factor if (factor is not None and factor.n == n and factor.e == e and factor.device == X.device) \
        else Factor(n, e, X.device)
    Xc, Rc = _c(X.detach()), _c(R.detach())
    var, ls, nz0 = _c(variance.detach()), _c(length_scales.detach()), _c(noise.detach())
    out = torch.empty(3, dtype=torch.float64, device=X.device)
    lib = _native.lib()

    def attempt(jitter):
        nz = nz0 if jitter is None else nz0 + jitter
        f.generation += 1
        f._winv_full = None
        if do_refine and f._refine_work is None:
            f._refine_work = torch.empty(max(1, int(lib.gpn_lml_refine_work_bytes(n, e)) // 8), dtype=torch.float64, device=X.device)
        if do_refine and save_k:
            # the assembly also leaves a pristine copy of Kyy's lower triangle (the factorisation overwrites it in place): the
            # refinement's residual pass READS it instead of re-computing every kernel entry
            if getattr(f, "_ksave", None) is None:
                f._ksave = torch.empty(n, f.ld, dtype=torch.float64, device=X.device)
            st = lib.gpn_lml_forward_saving(_stream(X.device), KINDS[kind], _ptr(Xc), n, Xc.shape[1], _ptr(Rc), None, e,
                                            _ptr(var), _ptr(ls), ls.numel(), _ptr(nz), _ptr(f.A), f.ld, _ptr(f.winv),
                                            _ptr(f.info), _ptr(out), _ptr(f._ksave))
            _native.check(st, "gpn_lml_forward_saving")
            st = lib.gpn_lml_refine_dense(_stream(X.device), _ptr(f._ksave), f.ld, 0.0, n, _ptr(Rc), None, e, _ptr(f.A), f.ld,
                                          _ptr(f.winv), _ptr(f._refine_work), _ptr(out))
            _native.check(st, "gpn_lml_refine_dense")
            return int(f.info.item())
        st = lib.gpn_lml_forward(_stream(X.device), KINDS[kind], _ptr(Xc), n, Xc.shape[1], _ptr(Rc), None, e,
                                 _ptr(var), _ptr(ls), ls.numel(), _ptr(nz), _ptr(f.A), f.ld, _ptr(f.winv),
                                 _ptr(f.info), _ptr(out))
        _native.check(st, "gpn_lml_forward")
        if do_refine:
            # enqueued before the info word is read back (no idle GPU during the host round trip); if the
            # factorisation failed its result is discarded with the attempt
            st = lib.gpn_lml_refine(_stream(X.device), KINDS[kind], _ptr(Xc), n, Xc.shape[1], _ptr(Rc), None, e,
                                    _ptr(var), _ptr(ls), ls.numel(), _ptr(nz), _ptr(f.A), f.ld, _ptr(f.winv),
                                    _ptr(f._refine_work), _ptr(out))
            _native.check(st, "gpn_lml_refine")
        return int(f.info.item())

    do_refine = (n >= refine_min_n()) if refine is None else bool(refine)
    import os
    save_k = os.environ.get("GPN_REFINE_SAVED_K", "0") == "1"
    if _DEFERRED:
        # one attempt, no read-back (see DeferredInfo): the caller inspects the flag later
        save_k = False
        _no_sync = _DEFERRED[-1]
        f.generation += 1
        f._winv_full = None
        if do_refine and f._refine_work is None:
            f._refine_work = torch.empty(max(1, int(lib.gpn_lml_refine_work_bytes(n, e)) // 8), dtype=torch.float64, device=X.device)
        st = lib.gpn_lml_forward(_stream(X.device), KINDS[kind], _ptr(Xc), n, Xc.shape[1], _ptr(Rc), None, e,
                                 _ptr(var), _ptr(ls), ls.numel(), _ptr(nz0), _ptr(f.A), f.ld, _ptr(f.winv),
                                 _ptr(f.info), _ptr(out))
        _native.check(st, "gpn_lml_forward")
        if do_refine:
            st = lib.gpn_lml_refine(_stream(X.device), KINDS[kind], _ptr(Xc), n, Xc.shape[1], _ptr(Rc), None, e,
                                    _ptr(var), _ptr(ls), ls.numel(), _ptr(nz0), _ptr(f.A), f.ld, _ptr(f.winv),
                                    _ptr(f._refine_work), _ptr(out))
            _native.check(st, "gpn_lml_refine")
        _no_sync.note(f.info)
        f.jitter_rung = -1
        f.refined = do_refine
        return f, out
    f.jitter_rung = _ladder(attempt)
    f.refined = do_refine
    return f, out


TRI_A_UPPER, TRI_A_LOWER, TRI_B_UPPER, TRI_B_LOWER = 1, 2, 4, 8


def gemm_nt(A, B, M, N, K, alpha=1.0, beta=0.0, C=None, lower=False, tri=0):
    """C[M,N] = alpha*A[M,K]*B[N,K]^T + beta*C (row strides taken from the tensors)."""
    _req(A, B, C)
    if C is None:
        C = torch.empty(M, N, dtype=torch.float64, device=A.device)
    st = _native.lib().gpn_gemm_nt(_stream(A.device), M, N, K, alpha, _ptr(A), A.stride(0), _ptr(B), B.stride(0),
                                   beta, _ptr(C), C.stride(0), int(lower), tri)       # lower: False/True or 2 = trapezoid
    _native.check(st, "gpn_gemm_nt")
    return C


def gemm_nt_stair(A, B, C, M, nblocks, blk, K, step, diag, alpha=-1.0, beta=1.0):
    """staircase contraction (gpnative.h gpn_gemm_nt_stair): C[M, nblocks*blk] = alpha A B^T + beta C where column
    block b only has the rows from b*step on and, with diag, a lower-only first square."""
    _req(A, B, C)
    st = _native.lib().gpn_gemm_nt_stair(_stream(A.device), M, nblocks, blk, K, alpha, _ptr(A), A.stride(0), _ptr(B), B.stride(0),
                                         beta, _ptr(C), C.stride(0), step, 1 if diag else 0)
    _native.check(st, "gpn_gemm_nt_stair")
    return C


def matmul_nt(A, B, alpha=1.0):
    """alpha * A @ B^T for arbitrary 2-D fp64 device tensors through the native contraction (pads K to
    a multiple of 16 and re-packs operands that are not 16-byte aligned row-major); for the
    autograd nodes of the off-hot-path `functions` / `util` surface."""
    _req(A, B)
    M, K = A.shape
    N = B.shape[0]
    if B.shape[1] != K:
        raise RuntimeError("matmul_nt: inner dimensions differ")
    out = torch.zeros(M, N, dtype=torch.float64, device=A.device)
    if M == 0 or N == 0 or K == 0:
        return out
    Kp = round_up(K, 16)

    def prep(T, rows):
        T = T.detach()
        ok = T.stride(1) == 1 and T.stride(0) % 2 == 0 and T.stride(0) >= Kp and T.data_ptr() % 16 == 0 \
            and K == Kp and T.shape[0] % 16 == 0
        if ok:
            return T
        P = torch.zeros(round_up(rows, 16), Kp, dtype=torch.float64, device=T.device)
        P[:rows, :K] = T
        return P
    return gemm_nt(prep(A, M), prep(B, N), M, N, Kp, alpha=alpha, C=out)


def gemm_nt_batched(A, B, M, N, K, batch, sA, sB, C, alpha=1.0, beta=0.0, lower=False, tri=0):
    """`batch` contractions of one shape in one launch: C[z] = alpha*A_z*B_z^T + beta*C[z], A_z = A + z*sA
    elements (row stride from the tensor), C [batch, rows, ld] contiguous."""
    _req(A, B, C)
    st = _native.lib().gpn_gemm_nt_batched(_stream(A.device), M, N, K, alpha, _ptr(A), A.stride(0), sA, _ptr(B), B.stride(0), sB,
                                           beta, _ptr(C), C.stride(1), C.stride(0), 1 if lower else 0, tri, batch)
    _native.check(st, "gpn_gemm_nt_batched")
    return C


def transpose(src):
    _req(src)
    src = _c(src.detach())
    rows, cols = src.shape
    dst = torch.empty(cols, rows, dtype=torch.float64, device=src.device)
    st = _native.lib().gpn_transpose(_stream(src.device), _ptr(src), rows, cols, cols, _ptr(dst), rows)
    _native.check(st, "gpn_transpose")
    return dst


def row_sumsq(A, rows, cols):
    out = torch.empty(rows, dtype=torch.float64, device=A.device)
    st = _native.lib().gpn_row_sumsq(_stream(A.device), _ptr(A), rows, cols, A.stride(0), _ptr(out))
    _native.check(st, "gpn_row_sumsq")
    return out


def dot2d_raw(x, ldx, sx, y, ldy, sy, rows, cols, batch, device):
    """out[z] = <x_z, y_z> over a rows x cols view (gpn_dot2d_batched; x, y: pointers, y None: the plain sum) -> [batch]."""
    out = torch.empty(batch, dtype=torch.float64, device=device)
    st = _native.lib().gpn_dot2d_batched(_stream(device), x, ldx, sx, y, ldy, sy, rows, cols, _ptr(out), batch)
    _native.check(st, "gpn_dot2d_batched")
    return out


def dot2d(x, y=None):
    """sum(x * y) (y None: sum(x)) of 2-D fp64 device tensors with unit inner stride, as a 0-dim tensor: ONE launch with a fixed
    summation order (the scalar sums of the sparse bound, sparse_gpr.py:139-151; their lock-step form is the same kernel)."""
    _req(x, y)
    x = x if x.stride(1) == 1 else x.contiguous()
    if y is not None:
        y = y if y.stride(1) == 1 else y.contiguous()
    return dot2d_raw(_ptr(x), x.stride(0), 0, _ptr(y), 0 if y is None else y.stride(0), 0, x.shape[0], x.shape[1], 1, x.device)[0]


def diag_sum(A, m):
    """sum of the first m diagonal entries of a row-major matrix (0-dim tensor; see dot2d)."""
    _req(A)
    return dot2d_raw(_ptr(A), A.stride(0) + 1, 0, None, 0, 0, m, 1, 1, A.device)[0]


def padded_like_factor(f, m):
    """zeroed [round_up(m,128), f.ld] buffer for right-hand sides of solve_right_lt."""
    return zeros(round_up(max(m, 1), LEAF), f.ld, f.device)


def trtrs_lower(b, f):
    """functions.trtrs(b, L, lower=True) (functions.py:71-76): X = L^-1 b, b [n, k]."""
    _req(b)
    n, k = b.shape
    if n != f.n:
        raise RuntimeError("trtrs: size mismatch")
    Bt = padded_like_factor(f, k)
    if n and k:
        bs = _c(b.detach())
        st = _native.lib().gpn_transpose(_stream(b.device), _ptr(bs), n, k, k, _ptr(Bt), f.ld)
        _native.check(st, "gpn_transpose")
        f.solve_right_lt(Bt, k)
    out = torch.empty(n, k, dtype=torch.float64, device=b.device)
    if n and k:
        st = _native.lib().gpn_transpose(_stream(b.device), _ptr(Bt), k, n, f.ld, _ptr(out), k)
        _native.check(st, "gpn_transpose")
    return out


# ----------------------------------------------------------------------------
# GPR predictive equations (gpr.py:88-117) in transposed storage
# ----------------------------------------------------------------------------
def lower_inverse(f):
    """W = L^-1 (lower, row-major, zero padded [rows, ld]) of a factor, cached on it: with W every
    right-solve X L^T = B is ONE K-clipped contraction X = B W^T instead of a chain of ~2 n/128
    small launches -- worth its n^3/3 flops for a model that serves many predictions."""
    W = f._winv_full
    if W is None:
        from . import _backward
        U = _backward._upper_inverse(f)
        W = zeros(U.shape[0], U.shape[1], U.device)
        if f.n:
            st = _native.lib().gpn_transpose(_stream(f.device), _ptr(U), f.n, f.n, f.ld, _ptr(W), f.ld)
            _native.check(st, "gpn_transpose")
        f._winv_full = W
    return W


BLOCKED_PREDICT_MIN_N = 4096      # from here on a repeated prediction pays for the inverted 1024 x 1024 diagonal blocks


def block_inverses(f):
    """inverses of the 1024 x 1024 diagonal blocks of a factor (gpn_block_inverse), cached on it for one generation."""
    wb = getattr(f, "_wblock", None)
    if wb is None or wb[0] != f.generation:
        lib = _native.lib()
        buf = torch.empty(max(1, int(lib.gpn_block_inverse_bytes(f.n)) // 8), dtype=torch.float64, device=f.device)
        _native.check(lib.gpn_block_inverse(_stream(f.device), _ptr(f.A), f.n, f.ld, _ptr(f.winv), _ptr(buf)), "gpn_block_inverse")
        f._wblock = wb = (f.generation, buf)
    return wb[1]


def gpr_predict(kind, X, x_new, variance, length_scales, f, diag=True, use_inverse=False, mean_new=None, blocked=False):
    """Returns (m(x*) + A^T V  [n*, dy],  var) with A = L^-1 K(X, x*), V = L^-1 (Y - m) held
    in f.extra(); var = rowsumsq-reduced diag [n*] or full K(x*) - A^T A [n*, n*].  mean_new [n*, dy]: the mean
    function at the test points (gpr.py:107-108), None = zero.  blocked: the right-solve through the inverted 1024 x 1024
    diagonal blocks (built on first use, cached on the factor)."""
    ns = x_new.shape[0]
    n, dy = f.n, f.e
    if blocked and not use_inverse and dy > 0 and n >= BLOCKED_PREDICT_MIN_N:
        _req(X, x_new, variance, length_scales)
        lib = _native.lib()
        wb = block_inverses(f)
        Xc, Xs = _c(X.detach()), _c(x_new.detach())
        var, ls = _c(variance.detach()), _c(length_scales.detach())
        work = torch.empty(max(1, 2 * int(lib.gpn_predict_work_bytes(n, ns, dy)) // 8), dtype=torch.float64, device=f.device)
        mean = torch.empty(ns, dy, dtype=torch.float64, device=f.device)
        out = torch.empty((ns,) if diag else (ns, ns), dtype=torch.float64, device=f.device)
        ms = None if mean_new is None else _c(mean_new.detach().expand(ns, dy))
        st = lib.gpn_predict_blocked(_stream(f.device), KINDS[kind], _ptr(Xc), n, Xc.shape[1], _ptr(Xs), ns, _ptr(ms), _ptr(var), _ptr(ls),
                                     ls.numel(), _ptr(f.A), f.ld, _ptr(f.winv), _ptr(wb), dy, 0 if diag else 1, _ptr(work), _ptr(mean),
                                     _ptr(out))
        _native.check(st, "gpn_predict_blocked")
        return mean, out
    if not use_inverse and dy > 0:
        # one library call: K(x*, X) -> right-solve chain -> mean / variance (gpn_predict)
        _req(X, x_new, variance, length_scales)
        lib = _native.lib()
        Xc, Xs = _c(X.detach()), _c(x_new.detach())
        var, ls = _c(variance.detach()), _c(length_scales.detach())
        work = torch.empty(max(1, int(lib.gpn_predict_work_bytes(n, ns, dy)) // 8), dtype=torch.float64, device=f.device)
        mean = torch.empty(ns, dy, dtype=torch.float64, device=f.device)
        out = torch.empty((ns,) if diag else (ns, ns), dtype=torch.float64, device=f.device)
        ms = None if mean_new is None else _c(mean_new.detach().expand(ns, dy))
        st = lib.gpn_predict(_stream(f.device), KINDS[kind], _ptr(Xc), n, Xc.shape[1], _ptr(Xs), ns, _ptr(ms), _ptr(var), _ptr(ls),
                             ls.numel(), _ptr(f.A), f.ld, _ptr(f.winv), dy, 0 if diag else 1, _ptr(work), _ptr(mean), _ptr(out))
        _native.check(st, "gpn_predict")
        return mean, out
    Bt = padded_like_factor(f, ns)
    if use_inverse:
        Ks = padded_like_factor(f, ns)
        kernel_matrix(kind, x_new, X, variance, length_scales, out=Ks, ldk=f.ld)
        gemm_nt(Ks, lower_inverse(f), ns, n, round_up(n, 16), C=Bt, tri=TRI_B_LOWER)   # A^T = K(x*,X) W^T
    else:
        kernel_matrix(kind, x_new, X, variance, length_scales, out=Bt, ldk=f.ld)   # K(x*, X) = k_ys^T
        f.solve_right_lt(Bt, ns)                                                   # A^T = K(x*,X) L^-T
    kpad = round_up(n, 16)
    mean = gemm_nt(Bt, f.A[n:], ns, dy, kpad)                                  # A^T V
    if mean_new is not None:
        mean = mean + mean_new
    if diag:
        var = variance.detach().expand(ns) - row_sumsq(Bt, ns, n)              # Kdiag - colsumsq(A)
    else:
        var = kernel_matrix(kind, x_new, None, variance, length_scales)
        gemm_nt(Bt, Bt, ns, ns, kpad, alpha=-1.0, beta=1.0, C=var)
    return mean, var


# ----------------------------------------------------------------------------
# differentiable LML
# ----------------------------------------------------------------------------
class GPRLogLik(torch.autograd.Function):
    """log p(y | X, theta) of gpr.py:47-67 as one autograd node over the native
    pipeline  K assembly -> Cholesky (+ fused forward substitution) -> reductions.
    Inputs are the CONSTRAINED hyper-parameters (1-element / [D] device tensors)
    and the residual R = y - mean(x); output has shape (1,) like the reference."""

    @staticmethod
    def forward(ctx, X, R, variance, length_scales, noise, kind, holder):
        f, terms = lml_forward(kind, X, R, variance, length_scales, noise, factor=holder.get("factor"))
        holder["factor"] = f
        ctx.kind = kind
        ctx.factor = f
        ctx.generation = f.generation
        ctx.save_for_backward(X, R, variance, length_scales, noise)
        return terms[2:3].clone()

    @staticmethod
    def backward(ctx, grad_out):
        from . import _backward
        X, R, variance, length_scales, noise = ctx.saved_tensors
        f = ctx.factor
        if f.generation != ctx.generation:
            # the model's reusable buffer was refactorised by a later forward (two losses alive at
            # once): this node's factor is gone, rebuild it privately rather than differentiate
            # the wrong one
            f = kernel_factor(ctx.kind, X, variance, length_scales, noise, R=R)
        g_var, g_ls, g_noise, g_R = _backward.lml_backward(ctx.kind, X, variance, length_scales, noise, f)
        go = grad_out.reshape(())
        return (None, go * g_R if ctx.needs_input_grad[1] else None, go * g_var, go * g_ls, go * g_noise, None, None)


class BatchedGPRLogLik(torch.autograd.Function):
    """GPRLogLik for `batch` independent models of one shape in LOCK STEP (hyper-parameter restarts; the reference runs
    loss(); backward(); step() one model at a time, gptorch/models/base.py:260-269): X [n, d] shared or [batch, n, d],
    R = Y - m(X) [n, dy] shared or [batch, n, dy], variance [batch], length_scales [batch, nls], noise [batch] (CONSTRAINED
    values) -> LML [batch].  Forward = gpn_lml_forward_batched, backward = gpn_lml_backward_batched; a model whose
    factorisation reports info != 0 is replayed ALONE through lml_forward (jitter ladder of functions.py:20-43) into a
    private factor, and its backward runs on that factor.  Values and gradients are bit-identical, model by model, to
    GPRLogLik -- also from refine_min_n() rows on, where every model's quadratic form is refined as GPRLogLik's is."""

    @staticmethod
    def forward(ctx, X, R, variance, length_scales, noise, kind, holder, n_of=None):
        # n_of (int32 device tensor [batch]; `sizes` = the same numbers on the host in holder["sizes"]): a RAGGED group -- X, R padded
        # to the largest model (lml_forward_batched(n_of=...)): no refinement, no gradient w.r.t. R
        batch = int(variance.numel())
        sizes = holder.get("sizes") if n_of is not None else None
        fb, terms = lml_forward_batched(kind, X, R, variance, length_scales, noise, fb=holder.get("fb"),
                                        refine=n_of is None and X.shape[-2] >= refine_min_n(), n_of=n_of)
        holder["fb"] = fb
        out = terms[:, 2].clone()
        replayed = {}
        if _DEFERRED:
            # under hipGraph capture (multi_start_optimize(capture=True)): nothing may read `info` back -- it is OR-ed into the
            # capture's device flag, and a chunk of replays that saw a failure is repeated eagerly (through the ladder below)
            _DEFERRED[-1].note(fb.info.abs().max().reshape(1))
            info = None
        else:
            info = fb.info.cpu()                     # ONE read-back for the batch (the reference: one per model and step)
        for b in (torch.nonzero(info).reshape(-1).tolist() if (info is not None and bool(info.any())) else ()):
            if int(info[b]) != 0:
                nb = X.shape[-2] if sizes is None else sizes[b]
                f, t = lml_forward(kind, (X if X.dim() == 2 else X[b])[:nb], (R if R.dim() == 2 else R[b])[:nb], variance.reshape(batch)[b:b + 1],
                                   length_scales.reshape(batch, -1)[b], noise.reshape(batch)[b:b + 1])
                out[b] = t[2]
                replayed[b] = f
        ctx.kind, ctx.fb, ctx.generation, ctx.replayed = kind, fb, fb.generation, replayed
        ctx.n_of, ctx.sizes = n_of, sizes
        ctx.save_for_backward(X, R, variance, length_scales, noise)
        return out

    @staticmethod
    def backward(ctx, grad_out):
        from . import _backward
        X, R, variance, length_scales, noise = ctx.saved_tensors
        batch = int(variance.numel())
        fb = ctx.fb
        if fb.generation != ctx.generation:
            # the shared buffers were refactorised by a later forward: rebuild this node's factors privately
            fb, _ = lml_forward_batched(ctx.kind, X, R, variance, length_scales, noise, n_of=ctx.n_of)
        need_r = ctx.needs_input_grad[1]
        grads, g_R = lml_backward_batched(ctx.kind, X, variance, length_scales, fb, need_resid=need_r, n_of=ctx.n_of)
        nls = grads.shape[1] - 2
        for b, f in ctx.replayed.items():
            nb = X.shape[-2] if ctx.sizes is None else ctx.sizes[b]
            gv, gl, gn, gr = _backward.lml_backward(ctx.kind, (X if X.dim() == 2 else X[b])[:nb], variance.reshape(batch)[b:b + 1],
                                                    length_scales.reshape(batch, -1)[b], noise.reshape(batch)[b:b + 1], f)
            grads[b, 0:1], grads[b, 1:1 + nls], grads[b, 1 + nls:] = gv, gl, gn
            if need_r:
                g_R[b] = gr
        go = grad_out.reshape(batch)
        g_resid = None
        if need_r:
            g_resid = go[:, None, None] * g_R
            if R.dim() == 2:
                g_resid = g_resid.sum(0)
        return (None, g_resid, (go * grads[:, 0]).reshape(variance.shape), (go[:, None] * grads[:, 1:1 + nls]).reshape(length_scales.shape),
                (go * grads[:, 1 + nls]).reshape(noise.shape), None, None, None)


class DenseLogLik(torch.autograd.Function):
    """The same LML for a kernel matrix that arrives as a dense tensor (Sum / Product /
    Linear / White ... kernels, whose K(X) is composed by autograd from several assemblies):
    Kyy = K + noise I is copied into the factor buffer, the factorisation and the closed-form
    backward are the native ones, and dLML/dK = 1/2 (a a^T - dy Kyy^-1) is handed back to
    autograd as a dense [n, n] gradient for the kernels' own backward sweeps."""

    @staticmethod
    def forward(ctx, K, R, noise):
        n = K.shape[0]
        Kyy = K.detach().clone()
        Kyy.diagonal().add_(noise.detach()[0])
        f = cholesky_factor(Kyy, rhs=R)
        ctx.factor = f
        terms = f.lml_terms()
        f.refined = False
        if n >= refine_min_n():
            # the refinement step of the quadratic form; the residual pass reads the dense Kyy it was factorised from
            lib = _native.lib()
            e = R.shape[1]
            jitter = 0.0 if f.jitter_rung < 0 else 10.0 ** (-JITTER_TRIES + f.jitter_rung)
            work = torch.empty(max(1, int(lib.gpn_lml_refine_work_bytes(n, e)) // 8), dtype=torch.float64, device=K.device)
            Rc = _c(R.detach())
            st = lib.gpn_lml_refine_dense(_stream(K.device), _ptr(Kyy), Kyy.stride(0), jitter, n, _ptr(Rc), None, e, _ptr(f.A), f.ld,
                                          _ptr(f.winv), _ptr(work), _ptr(terms))
            _native.check(st, "gpn_lml_refine_dense")
            f.refined = True
        return terms[2:3].clone()

    @staticmethod
    def backward(ctx, grad_out):
        from . import _backward
        f = ctx.factor
        n, dy = f.n, f.e
        U = _backward._upper_inverse(f)
        Kinv = _backward._kinv_lower(f, U)[:n, :n]
        a_t = gemm_nt(f.A[n:], U, dy, n, round_up(n, 16), tri=TRI_B_UPPER)          # a^T = alpha^T U^T
        ap = torch.zeros(round_up(n, 16), round_up(dy, 16), dtype=torch.float64, device=f.device)
        ap[:n, :dy] = a_t.t()
        G = gemm_nt(ap, ap, n, n, round_up(dy, 16), alpha=0.5)                      # 1/2 a a^T
        G -= 0.5 * dy * (torch.tril(Kinv) + torch.tril(Kinv, -1).t())
        go = grad_out.reshape(())
        return (go * G if ctx.needs_input_grad[0] else None,
                -go * a_t.t() if ctx.needs_input_grad[1] else None,
                (go * G.diagonal().sum()).reshape(1) if ctx.needs_input_grad[2] else None)


LOG_2PI = math.log(2.0 * math.pi)

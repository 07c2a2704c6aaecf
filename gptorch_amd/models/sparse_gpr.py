"""
Sparse GP regression: VFE (Titsias' collapsed bound), gptorch/models/sparse_gpr.py:22-195,
BASELINE config 5 / SURVEY 8(f)-1 (N = 1e6 points, M = 4096 inducing points).

Forward.  Everything N-sized is STREAMED in row chunks of x (CHUNK_ROWS at a time), so the
M x N matrices Kuf / A of the reference are never held -- only two chunk-sized scratch
buffers and a handful of M x M matrices live in HBM -- and every contraction is the NT
fp64-MFMA form of the library:

    per chunk c:  A_c^T = K(x_c, Z) L^-T        [nc, M]  assembly + in-place right-solve
                  A_c   = (A_c^T)^T             [M, nc]  HBM-bound transpose
                  AAT  += A_c A_c^T / s2        [M, M]   SYRK (lower), beta = 1
                  Aerr += A_c err_c             [M, dy]
    then          B = AAT + I = LB LB^T, with (Aerr)^T carried as the factor's extra rows,
                  so c * s2 = LB^-1 Aerr falls out of the factorisation.

Backward (closed form, verified against autograd through the reference's op chain in
tests/golden/make_golden.py; the reference gets it from autograd through two Cholesky
backwards and two M x N triangular-solve backwards).  With p = dy, s = s2,
beta = B^-1 A err / s, gamma = L^-T beta:

    dF/dKuu = L^-T [ p/2 (2I - B^-1 - B) - 1/2 beta beta^T ] L^-1                (M x M)
    dF/dKuf = 1/s ( P Kuf + gamma err^T ),  P = L^-T [ p (I - B^-1) - beta beta^T ] L^-1
    dF/ds   = p/(2s) (M - tr B^-1) - |c|^2/s + beta^T (B - I) beta/(2s) - p tr(AAT)/(2s)
              - p N/(2s) + (|err|^2 + p tr Kff)/(2 s^2),        dF/dvar += -p N/(2s)

so the N-sized part is ONE dense [nc, M] x [M, M] contraction per chunk (no solve), followed
by the HBM-bound sweeps that contract dF/dKuf with dK/dtheta (gpn_kernel_grad) and dK/dZ
(gpn_kernel_grad_x2).  Forward + backward cost 4 N M^2 flops instead of holding 2 x 33 GB.

Why the forward keeps the N-sized solve: A A^T = L^-1 (Kuf Kfu) L^-T would halve the forward
flops (no N x M TRSM), but only the Gram form A A^T of a COMPUTED A is positive semi-definite
by construction; the sandwich loses definiteness by ~eps |Kuf|^2 / lambda_min(Kuu), and at
config 5 (4096 inducing points, Kuu at the edge of the jitter ladder) chol(B) then fails on
every rung.  Measured, not assumed (round 1).

SVGP / FITC (sparse_gpr.py:76-90, 198-381) are out of scope.
"""
import math

import numpy as np
import torch

from .. import _backward, _ops
from ..mean_functions import Zero
from ..param import Param
from ..util import as_tensor
from .base import GPModel

CHUNK_ROWS = 65536   # rows of x per streamed chunk (scratch: 2 x CHUNK_ROWS x ld(M) x 8 B)
# Measured alternative, OFF by default (round 3): from INVERSE_MIN_M inducing points on (and N >= 4 M) form W = L^-1 once per
# evaluation (M^3/3 flops) and compute every chunk's A_c = L^-1 Kuf_c as W Kuf_c -- already in the [M, rows] layout the A A^T
# accumulation reads, so the right-solve recursion AND the transpose go.  Same bound to 8e-12 relative at config 5, gradients
# equal to the solve path's wherever Kuu is not numerically singular (tests: test_vfe_inverse_path_vs_oracle) -- but not
# faster: as ONE K-clipped launch per chunk 0.610 s (uneven tiles, dealt round-robin over the XCDs), as block rows of
# INVERSE_BLOCK rows with plain launches 0.561-0.564 s, against 0.549-0.552 s for the right-solve (same box, interleaved;
# profiles/r3_vfe_inverse_ab.txt): with two chunk pipelines in flight the bound is throughput-bound at ~62 TFLOP/s and the
# inverse form carries 6 % more flops.
INVERSE_MIN_M = int(__import__('os').environ.get('GPN_VFE_INVERSE_MIN_M', 1 << 62))
INVERSE_BLOCK = int(__import__('os').environ.get('GPN_VFE_INVERSE_BLOCK', 512))
# N-sharding over the GPUs of a node: when set to a torch.distributed process group
# (or True for the default group) every rank holds a ROW SHARD of (x, y) and the same Z and
# hyper-parameters; the M-sized sums A A^T, A err and the scalars N, |y|^2 are all-reduced in the
# forward and the gradients in the backward, so every rank sees the bound and the gradients of the
# WHOLE data set (optimisers on all ranks stay in step).
SHARD_GROUP = None
BLOCKED_SOLVE_MIN_M = 2048   # from this many inducing points on, chunk right-solves go through the inverted 1024 x 1024 blocks
LANES = 2                # chunk pipelines in flight (1: strictly one chunk after the other)
SPLIT_K = 8              # partial accumulators of the A A^T accumulation (1: none)
SYRK_K_SLICE = 8192      # columns of a chunk per accumulation launch (0: the whole chunk at once)


def _all_reduce(t):
    import torch.distributed as dist
    if SHARD_GROUP is None:
        return t
    dist.all_reduce(t, op=dist.ReduceOp.SUM, group=None if SHARD_GROUP is True else SHARD_GROUP)
    return t


def _shard_rank():
    import torch.distributed as dist
    if SHARD_GROUP is None:
        return 0
    return dist.get_rank(None if SHARD_GROUP is True else SHARD_GROUP)


class _InducingPointsGP(GPModel):
    """sparse_gpr.py:22-73; default inducing points = k-means centres (util.py:34-49)."""

    def __init__(self, x, y, kernel, num_inducing_points=None, inducing_points=None, mean_function=None,
                 likelihood=None):
        super().__init__(x, y, kernel, likelihood, mean_function)
        if inducing_points is None:
            from scipy.cluster.vq import kmeans2
            if num_inducing_points is None:
                num_inducing_points = np.clip(x.shape[0] // 10, 1, 100)
            xn = x.detach().cpu().numpy() if isinstance(x, torch.Tensor) else np.asarray(x)
            try:
                inducing_points = kmeans2(xn, int(num_inducing_points))[0]
            except np.linalg.LinAlgError:
                xp = xn + 1.0e-4 * xn.std(axis=0) * np.random.randn(*xn.shape)
                inducing_points = kmeans2(xp, int(num_inducing_points))[0]
        self.Z = Param(as_tensor(inducing_points))

    @property
    def num_inducing(self) -> int:
        return self.Z.shape[0]


def _zeros(rows, cols, device):
    return torch.zeros(rows, cols, dtype=torch.float64, device=device)


def _chunks(n, nc):
    for c0 in range(0, n, nc):
        yield c0, min(nc, n - c0)


class _State:
    """what one evaluation of the bound leaves behind (all M-sized)."""
    __slots__ = ("f_uu", "fB", "AAT", "Aerr", "s2", "tr", "terms", "n", "n_all", "yy_all", "trkff")


class _NativeAsm:
    """K(x_c, Z), K(Z) and their gradient sweeps for a kernel with a native kind: the fused assembly
    kernel and the native sweeps (gpn_kernel_grad, gpn_kernel_grad_x2)."""

    def __init__(self, kind, var, ls):
        self.kind, self.var, self.ls = kind, var, ls

    def factor_uu(self, Z):
        return _ops.kernel_factor(self.kind, Z, self.var, self.ls, None)          # L = chol(K(Z)) (+ladder)

    def kuf(self, xc, Z, out, ldk):
        _ops.kernel_matrix(self.kind, xc, Z, self.var, self.ls, out=out, ldk=ldk)

    def trkff(self, x):
        return x.shape[0] * self.var[0]                                           # Kdiag = variance (kernels.py:174-179)

    # -- backward: accumulators for (variance, length_scales, Z)
    def begin(self, Z):
        self.g_var = torch.zeros(1, dtype=torch.float64, device=Z.device)
        self.g_ls = torch.zeros_like(self.ls)
        self.g_Z = torch.zeros_like(Z)

    def grad_uu(self, Z, Guu):
        gv, gl = _backward.kernel_backward(self.kind, Z, None, self.var, self.ls, Guu)
        self.g_var += gv
        self.g_ls += gl
        _backward.kernel_backward_x2(self.kind, Z, Z, self.var, self.ls, Guu, scale=2.0, out=self.g_Z)

    def grad_uf(self, xc, Z, G):
        gv, gl = _backward.kernel_backward(self.kind, xc, Z, self.var, self.ls, G)
        self.g_var += gv
        self.g_ls += gl
        _backward.kernel_backward_x2(self.kind, xc, Z, self.var, self.ls, G, out=self.g_Z)

    def grad_trkff(self, coef, n_all):
        self.g_var += coef * n_all                                                 # d tr Kff / d variance = N

    def tensors(self):
        return [self.g_var, self.g_ls, self.g_Z]


class _GenericAsm:
    """The same for ANY kernel object (sparse_gpr.py:126-129 takes whatever `self.kernel` is: sums,
    products, Linear, ...): K(x_c, Z) and K(Z) come from the kernel's own `K` (whose stationary
    leaves are the native assembly), and the gradient of sum(G * K) goes back through the kernel's own
    autograd nodes, chunk by chunk -- to the RAW parameters directly, in `params` order."""

    def __init__(self, kernel, params):
        self.kernel, self.params = kernel, params

    def factor_uu(self, Z):
        with torch.no_grad():
            return _ops.cholesky_factor(self.kernel.K(Z))

    def kuf(self, xc, Z, out, ldk):
        with torch.no_grad():
            out[:xc.shape[0], :Z.shape[0]] = self.kernel.K(xc, Z)

    def trkff(self, x):
        with torch.no_grad():
            return self.kernel.Kdiag(x).sum()

    def begin(self, Z):
        self.g_params = [torch.zeros_like(p) for p in self.params]
        self.g_Z = torch.zeros_like(Z)

    def _pull(self, out, G, Zg):
        grads = torch.autograd.grad(out, self.params + [Zg], grad_outputs=G, allow_unused=True)
        for acc, g in zip(self.g_params + [self.g_Z], grads):
            if g is not None:
                acc += g

    def grad_uu(self, Z, Guu):
        with torch.enable_grad():
            Zg = Z.detach().requires_grad_(True)
            self._pull(self.kernel.K(Zg), Guu, Zg)

    def grad_uf(self, xc, Z, G):
        with torch.enable_grad():
            Zg = Z.detach().requires_grad_(True)
            self._pull(self.kernel.K(xc, Zg), G, Zg)

    def grad_trkff(self, coef, n_all):
        # row shards: every rank differentiates the diagonal of ITS rows; the all-reduce sums them
        with torch.enable_grad():
            tr = self.kernel.Kdiag(self._x).sum()
            grads = torch.autograd.grad(tr, self.params, allow_unused=True)
        for acc, g in zip(self.g_params, grads):
            if g is not None:
                acc += coef * g

    def tensors(self):
        return self.g_params + [self.g_Z]


def _vfe_forward(asm, x, err, Z, s2):
    """streamed evaluation of sparse_gpr.py:126-137 -> _State."""
    dev = x.device
    n, dy = err.shape
    m = Z.shape[0]
    st = _State()
    st.n, st.s2 = n, s2
    st.f_uu = f_uu = asm.factor_uu(Z)
    fB = _ops.Factor(m, dy, dev)
    st.AAT = AAT = torch.zeros_like(fB.A)
    mp = _ops.round_up(m, 16)
    Aerr = _zeros(mp, dy, dev)
    nc = min(_ops.round_up(n, _ops.LEAF), _ops.round_up(CHUNK_ROWS, _ops.LEAF))
    nchunks = (n + nc - 1) // nc
    # Two chunk pipelines on two streams: while one chunk's SYRK (whose 2080 tiles fill the
    # 1280 workgroup slots 1.6 times: a poor last round) accumulates, the next chunk's assembly,
    # right-solve and transpose already run next to it.  The accumulations into AAT / Aerr are
    # ordered by events.
    lanes = min(LANES, 2) if nchunks > 1 else 1
    cur = torch.cuda.current_stream(dev)
    streams = [cur] + [torch.cuda.Stream(device=dev) for _ in range(lanes - 1)]
    bufs = [(_zeros(nc + 16, f_uu.ld, dev), _zeros(mp, nc, dev), _zeros(_ops.round_up(dy, 16), nc, dev))
            for _ in range(lanes)]
    lib = _ops._native.lib()
    # round 4: the chunk's right-solve through the inverted 1024 x 1024 diagonal blocks of L_uu (gpn_trsm_right_lt_blocked:
    # M / 1024 steps of two large contractions) instead of the recursion down to the 128-wide leaf inverses, whose K <= 256
    # levels ran at ~12 TFLOP/s (round-3 review: colpanel_kernel 19 % of C5's kernel time); one more chunk-sized buffer per lane
    blocked = m >= BLOCKED_SOLVE_MIN_M and n >= 4 * m
    wb_uu = _ops.block_inverses(f_uu) if blocked else None
    xbufs = [_zeros(nc + 16, f_uu.ld, dev) for _ in range(lanes)] if blocked else None
    # split-K partial accumulators (see below): only when M^2/2 has too few 128x128 tiles to fill the GPU
    mt128 = (m + 127) // 128
    split = SPLIT_K if (mt128 * (mt128 + 1) // 2 < 4096 and nc >= 4096 * SPLIT_K) else 1
    parts = torch.zeros(split, AAT.shape[0], AAT.shape[1], dtype=torch.float64, device=dev) if split > 1 else None
    aerr_parts = torch.zeros(split, mp, dy, dtype=torch.float64, device=dev) if split > 1 else None
    acc_done = None                                                        # event: AAT/Aerr updated through chunk c-1
    W_uu = _ops.lower_inverse(f_uu) if (m >= INVERSE_MIN_M and n >= 4 * m) else None
    for stq in streams[1:]:
        stq.wait_stream(cur)
    for ci, (c0, r) in enumerate(_chunks(n, nc)):
        stq = streams[ci % lanes]
        At, A, errT = bufs[ci % lanes]
        with torch.cuda.stream(stq):
            stream = _ops._stream(dev)
            if r < nc:                                                     # ragged tail: stale entries -> 0
                At.zero_(), A.zero_(), errT.zero_()
            asm.kuf(x[c0:c0 + r], Z, At, f_uu.ld)
            if W_uu is not None:
                # A_c = W Kuf_c, W = L^-1 lower triangular: block rows of INVERSE_BLOCK rows, each with the K range it
                # needs (k < its last row) as a plain contraction
                for b0 in range(0, m, INVERSE_BLOCK):
                    rows = min(INVERSE_BLOCK, m - b0)
                    _ops.gemm_nt(W_uu[b0:], At, rows, r, _ops.round_up(b0 + rows, 16), C=A[b0:])
            elif blocked:
                Xo = xbufs[ci % lanes]
                _ops._native.check(lib.gpn_trsm_right_lt_blocked(stream, _ops._ptr(f_uu.A), m, f_uu.ld, _ops._ptr(wb_uu), _ops._ptr(At), r,
                                                                 At.stride(0), _ops._ptr(Xo), Xo.stride(0)), "gpn_trsm_right_lt_blocked")
                _ops._native.check(lib.gpn_transpose(stream, _ops._ptr(Xo), r, m, Xo.stride(0), _ops._ptr(A), nc), "gpn_transpose")
            else:
                f_uu.solve_right_lt(At, r)                                 # A_c^T = Kuf_c^T L^-T
                _ops._native.check(lib.gpn_transpose(stream, _ops._ptr(At), r, m, At.stride(0), _ops._ptr(A), nc),
                                   "gpn_transpose")
            errT[:dy, :r] = err[c0:c0 + r].t()
            kp = _ops.round_up(r, 16)
            first = 0.0 if c0 == 0 else 1.0
            if acc_done is not None:
                stq.wait_event(acc_done)
            if split > 1 and kp % (16 * split) == 0:
                # split-K: M^2/2 alone is too few tiles for the GPU (M = 4096: 2080 64x64 tiles = 1.6
                # rounds of the 1280 slots, 528 128x128 tiles = 1.03 rounds of 512), so the chunk's K
                # range is dealt over `split` partial accumulators in ONE launch (8 x 528 tiles =
                # 8.25 rounds); the partials are summed once, after the last chunk
                _ops.gemm_nt_batched(A, A, m, m, kp // split, split, kp // split, kp // split, parts,
                                     alpha=1.0 / s2, beta=first, lower=True)
                # A err likewise: a 4096 x 65536 matrix-vector product is 128 workgroups of 4096
                # K-steps each as one skinny contraction (0.98 ms), 8x more of 8x shorter ones here
                _ops.gemm_nt_batched(A, errT, m, dy, kp // split, split, kp // split, kp // split, aerr_parts, beta=first)
            else:
                # K slices in sequence: one launch over the whole chunk keeps every workgroup slot
                # for ~10 ms and the other lane's short kernels starve behind it (no pre-emption)
                ks = SYRK_K_SLICE or kp
                tgt = parts[0] if split > 1 else AAT
                for k0 in range(0, kp, ks):
                    kk = min(ks, kp - k0)
                    _ops.gemm_nt(A[:, k0:], A[:, k0:], m, m, kk, alpha=1.0 / s2, beta=(first if k0 == 0 else 1.0), C=tgt, lower=True)
                _ops.gemm_nt(A, errT, m, dy, kp, beta=first, C=(aerr_parts[0] if split > 1 else Aerr))
            acc_done = torch.cuda.Event()
            acc_done.record(stq)
    for stq in streams[1:]:
        cur.wait_stream(stq)
    if acc_done is not None:
        cur.wait_event(acc_done)
    if split > 1:
        torch.sum(parts, dim=0, out=AAT)
        torch.sum(aerr_parts, dim=0, out=Aerr)
        del parts
    # row shards: one all-reduce of the M-sized sums and of (N, |err|^2)
    scal = torch.tensor([float(n), 0.0, 0.0], dtype=torch.float64, device=dev)
    scal[1] = _ops.dot2d(err, err)
    scal[2] = asm.trkff(x)                                                 # tr Kff (sparse_gpr.py:139-141)
    if SHARD_GROUP is not None:
        _all_reduce(AAT)
        _all_reduce(Aerr)
        _all_reduce(scal)
    st.n_all, st.yy_all, st.trkff = int(round(scal[0].item())), scal[1], scal[2]
    st.Aerr = Aerr[:m]
    st.tr = _ops.diag_sum(AAT, m)

    def attempt(jitter):
        fB.A.copy_(AAT)
        fB.A.diagonal()[:m].add_(1.0 if jitter is None else 1.0 + jitter)  # B = AAT + I
        fB.pack_rhs(st.Aerr)
        return fB.potrf()
    _ops._ladder(attempt)
    st.fB = fB
    st.terms = fB.lml_terms()            # [sum log LB_ii, || LB^-1 A err ||^2 = s2^2 |c|^2, ...]
    return st


def _sandwich(U, W, m):
    """U W U^T for upper-triangular U = L^-T and dense symmetric W (zero-padded buffers)."""
    kp = _ops.round_up(m, 16)
    T = torch.zeros_like(U)
    _ops.gemm_nt(U, W, m, m, kp, C=T, tri=_ops.TRI_A_UPPER)              # T = U W   (W = W^T)
    R = torch.zeros_like(U)
    _ops.gemm_nt(T, U, m, m, kp, C=R, tri=_ops.TRI_B_UPPER)              # R = T U^T
    return R


def _vfe_backward(asm, x, err, Z, st):
    """-> dF/d noise [1] (constrained value); the kernel / inducing-point gradients are left in `asm`
    (native kinds: w.r.t. the constrained variance / length-scales and Z; any other kernel: w.r.t. its
    raw parameters and Z)."""
    dev = x.device
    n, p = err.shape
    m, s = Z.shape[0], st.s2
    f_uu, fB = st.f_uu, st.fB
    mp, pp = _ops.round_up(m, 16), _ops.round_up(p, 16)
    U = _backward._upper_inverse(f_uu)                                     # L^-T
    UB = _backward._upper_inverse(fB)                                      # LB^-T
    Binv = _backward._kinv_lower(fB, UB)[:m, :m]
    Binv = torch.tril(Binv) + torch.tril(Binv, -1).t()
    Bd = torch.tril(st.AAT[:m, :m]) + torch.tril(st.AAT[:m, :m], -1).t()
    Bd.diagonal().add_(1.0)
    eye = torch.eye(m, dtype=torch.float64, device=dev)
    # beta^T = c^T LB^-1: the extra rows hold (LB^-1 A err)^T = s c^T
    bt = _zeros(pp, f_uu.ld, dev)
    _ops.gemm_nt(fB.A[m:], UB, p, m, mp, alpha=1.0 / s, C=bt, tri=_ops.TRI_B_UPPER)
    beta = _zeros(mp, pp, dev)
    beta[:m, :p] = bt[:p, :m].t()
    bbT = _ops.gemm_nt(beta, beta, m, m, pp)
    W = torch.zeros_like(U)
    W[:m, :m] = p * (eye - Binv) - bbT
    P = _sandwich(U, W, m)                                                 # s * dF/dKuf = P Kuf + gamma err^T
    W[:m, :m] = 0.5 * p * (2.0 * eye - Binv - Bd) - 0.5 * bbT
    Guu = _sandwich(U, W, m)                                               # dF/dKuu
    gt = _ops.gemm_nt(bt, U, p, m, mp, tri=_ops.TRI_B_UPPER)              # gamma^T = beta^T L^-1

    # K(Z, Z) part (identical on every rank of a sharded run: counted on the first one only)
    asm.begin(Z)
    if _shard_rank() == 0:
        asm.grad_uu(Z, Guu[:m, :m])

    # K(x, Z) part, streamed:  G_c = 1/s [K(x_c, Z) | err_c] [P | gamma]^T
    ldk = mp + pp
    Bq = _zeros(mp, ldk, dev)
    Bq[:m, :m] = P[:m, :m]
    Bq[:m, mp:mp + p] = gt[:p, :m].t()
    nc = min(_ops.round_up(n, _ops.LEAF), _ops.round_up(CHUNK_ROWS, _ops.LEAF))
    Kx = _zeros(nc + 16, ldk, dev)
    G = torch.empty(nc, mp, dtype=torch.float64, device=dev)
    for c0, r in _chunks(n, nc):
        xc = x[c0:c0 + r]
        asm.kuf(xc, Z, Kx, ldk)
        Kx[:r, mp:mp + p] = err[c0:c0 + r]
        _ops.gemm_nt(Kx, Bq, r, m, ldk, alpha=1.0 / s, C=G)
        asm.grad_uf(xc, Z, G[:r, :m])
    asm._x = x
    asm.grad_trkff(-0.5 * p / s, n if SHARD_GROUP is not None else st.n_all)   # -p/(2s) d tr Kff (this rank's rows)

    if SHARD_GROUP is not None:                                            # sum the row shards' contributions
        for t in asm.tensors():
            _all_reduce(t)
    n_all = st.n_all
    c2 = st.terms[1] / (s * s)
    b = beta[:m, :p]
    quad = _ops.dot2d(b, st.Aerr) / s - _ops.dot2d(b, b)                   # beta^T (B - I) beta
    g_noise = (0.5 * p / s) * (m - _ops.diag_sum(Binv, m)) - c2 / s + 0.5 * quad / s - 0.5 * p * st.tr / s \
        - 0.5 * p * n_all / s + 0.5 * (st.yy_all + p * st.trkff) / (s * s)
    return g_noise.reshape(1)


def _elbo(st, p, s2):
    """sparse_gpr.py:139-151 from the M-sized state."""
    n = st.n_all                                                           # N of the whole data set
    elbo = -0.5 * p * n * math.log(2.0 * math.pi)
    elbo = elbo - p * st.terms[0]
    elbo = elbo - 0.5 * p * n * math.log(s2)
    elbo = elbo - 0.5 * (st.yy_all + p * st.trkff) / s2
    elbo = elbo + 0.5 * st.terms[1] / (s2 * s2)                            # c = LB^-1 (A err) / s2
    elbo = elbo + 0.5 * p * st.tr
    return elbo


class _VFEBound(torch.autograd.Function):
    """The collapsed bound as one autograd node over (variance, length_scales, noise, Z)."""

    @staticmethod
    def forward(ctx, variance, length_scales, noise, Z, kind, x, err, holder):
        s2 = float(noise.item())
        asm = _NativeAsm(kind, variance.detach(), length_scales.detach())
        st = _vfe_forward(asm, x, err, Z.detach(), s2)
        ctx.asm, ctx.x, ctx.err, ctx.st = asm, x, err, st
        ctx.save_for_backward(length_scales, Z)
        holder["state"] = st
        return _elbo(st, err.shape[1], s2)

    @staticmethod
    def backward(ctx, grad_out):
        length_scales, Z = ctx.saved_tensors
        g_noise = _vfe_backward(ctx.asm, ctx.x, ctx.err, Z.detach(), ctx.st)
        g_var, g_ls, g_Z = ctx.asm.tensors()
        g = grad_out
        return (g * g_var, g * g_ls.reshape(length_scales.shape), g * g_noise, g * g_Z,
                None, None, None, None)


class _VFEBoundGeneric(torch.autograd.Function):
    """The same bound for any kernel object: node over (noise, Z, *raw kernel parameters)."""

    @staticmethod
    def forward(ctx, noise, Z, kernel, x, err, holder, *params):
        s2 = float(noise.item())
        asm = _GenericAsm(kernel, list(params))
        st = _vfe_forward(asm, x, err, Z.detach(), s2)
        ctx.asm, ctx.x, ctx.err, ctx.st = asm, x, err, st
        ctx.save_for_backward(Z)
        holder["state"] = st
        return _elbo(st, err.shape[1], s2)

    @staticmethod
    def backward(ctx, grad_out):
        Z, = ctx.saved_tensors
        g_noise = _vfe_backward(ctx.asm, ctx.x, ctx.err, Z.detach(), ctx.st)
        *g_params, g_Z = ctx.asm.tensors()
        g = grad_out
        return (g * g_noise, g * g_Z, None, None, None, None) + tuple(g * t for t in g_params)


class VFE(_InducingPointsGP):
    def __init__(self, *args, **kwargs):
        super().__init__(*args, **kwargs)
        assert isinstance(self.mean_function, Zero), "Mean functions not implemented for VFE yet."

    def _native_kernel(self):
        """the kernel if it is one of the native stationary kinds (fused assembly + native sweeps),
        else None (sums / products / Linear / static kernels: the kernel's own K and autograd)."""
        from .. import kernels
        k = self.kernel
        return k if isinstance(k, kernels.Stationary) and k._kind is not None else None

    def _bound(self, x):
        k = self._native_kernel()
        holder = {}
        s2 = self.likelihood.variance.transform()
        if k is not None:
            elbo = _VFEBound.apply(k.variance.transform(), k.length_scales.transform(), s2, self.Z, k._kind, x,
                                   self.Y, holder)                       # sparse_gpr.py:125 quirk: err = self.Y
        else:
            params = [p for p in self.kernel.parameters() if p.requires_grad]
            elbo = _VFEBoundGeneric.apply(s2, self.Z, self.kernel, x, self.Y, holder, *params)
        return elbo, holder["state"]

    def log_likelihood(self, x=None, y=None):
        """variational lower bound, sparse_gpr.py:108-153 (0-dim tensor)."""
        x = x if x is not None else self.X
        y = y if y is not None else self.Y
        if not x.shape[0] == y.shape[0]:
            raise ValueError("X and Y must have same # data.")
        return self._bound(x)[0]

    def _state_for_predict(self, x):
        """the M-sized state of the bound (chol K(Z), chol B, c) that a prediction starts from.  The reference re-evaluates the bound
        inside every _predict (sparse_gpr.py:155-170: two factorisations and the N-sized solves again); here it is kept between
        predictions, like GPR's factor (models/gpr.py:_factor_for_predict): the cache HOLDS the tensors it was built from and
        compares identity + version counters for the data and VALUES for every parameter (inducing points included) -- one
        concatenation + one comparison per prediction."""
        key = (x._version, tuple(x.shape), self.Y._version)
        with torch.no_grad():
            params = torch.cat([p.detach().reshape(-1) for p in self.parameters()])
            c = getattr(self, "_predict_cache", None)
            if c is None or c[0] != key or c[3] is not x or c[4] is not self.Y or c[2].shape != params.shape \
                    or not torch.equal(c[2], params):
                _, st = self._bound(x)
                self._predict_cache = (key, st, params, x, self.Y)
        return self._predict_cache[1]

    def _predict(self, x_new, diag=True, x=None):
        """sparse_gpr.py:155-195."""
        x = x if x is not None else self.X
        kern = self.kernel
        with torch.no_grad():
            st = self._state_for_predict(x)
            f_uu, fB, s2 = st.f_uu, st.fB, st.s2
            ns, m, dy = x_new.shape[0], self.Z.shape[0], self.Y.shape[1]
            T1 = _ops.padded_like_factor(f_uu, ns)                                    # tmp1^T = K(x*, Z) L^-T
            T1[:ns, :m] = kern.K(x_new, self.Z.detach())
            f_uu.solve_right_lt(T1, ns)
            T2 = T1.clone()
            fB.solve_right_lt(T2, ns)                                                 # tmp2^T = tmp1^T LB^-T
            kp = _ops.round_up(m, 16)
            mean = _ops.gemm_nt(T2, fB.A[m:], ns, dy, kp) / s2                        # tmp2^T c
            if diag:
                v = kern.Kdiag(x_new).detach() - _ops.row_sumsq(T1, ns, m) + _ops.row_sumsq(T2, ns, m)
                return mean, v[:, None].expand_as(mean)
            cov = kern.K(x_new).clone()
            _ops.gemm_nt(T2, T2, ns, ns, kp, alpha=1.0, beta=1.0, C=cov)
            _ops.gemm_nt(T1, T1, ns, ns, kp, alpha=-1.0, beta=1.0, C=cov)
        return mean, cov

"""
The collapsed sparse bound (VFE, gptorch/models/sparse_gpr.py:108-153) of B INDEPENDENT models of one shape in LOCK STEP.

The reference evaluates one model per optimiser step (gptorch/models/base.py:260-269); a multi-start search over inducing
points / hyper-parameters runs many VFE models of one (kernel kind, N, M, D, dy), and at the sizes such searches run at (N up
to ~10^4, M up to ~10^3) one model's ~100 launches each leave most of the chip idle.  Here every launch of sparse_gpr.py's
`_vfe_forward` / `_vfe_backward` goes out ONCE over all models:

    chol K(Z_b)            gpn_kernel_matrix_batched (lower) + gpn_potrf_lower_batched
    A_b^T = K(x, Z_b) L^-T gpn_kernel_matrix_batched + gpn_trsm_right_lt_batched
    AAT_b, A_b err         gpn_gemm_nt_batched_scaled (1 / noise variance of model b from device memory), gpn_gemm_nt_batched
    chol(AAT_b + I)        gpn_potrf_lower_batched, with (A err)^T as extra rows; gpn_lml_reduce_batched
    backward               gpn_trtri_upper_batched (L^-T, LB^-T), strided-batch contractions for B^-1, beta, the two sandwiches,
                           dF/dKuf; gpn_kernel_grad_batched / gpn_kernel_grad_x2_batched for the sweeps

Same kernels per model, same per-entry summation order, and the host-side scalar arithmetic of the sequential code restated
operation by operation (a tensor divided by a Python float is a multiplication by its reciprocal in PyTorch: the reciprocals
are formed on the host exactly as there): every model's bound and gradients are BIT-IDENTICAL to its own
`log_likelihood()` / `loss(); backward()` (tests/test_gpu_vfe_lockstep.py).  The jitter ladder (functions.py:20-43) runs in lock
step too: the models whose chol K(Z) (or chol(B)) reports info != 0 climb it together as a sub-batch, rung by rung, and a
factor that succeeds is copied into its slot -- the same assembly + factorisation per model as the sequential ladder.

Scope: native stationary kinds, the single-chunk regime (N below 32768 rows: no split-K accumulators, no chunk pipelines),
right-solves by recursion (M < BLOCKED_SOLVE_MIN_M or N < 4 M).  Anything else -- config 5's N = 10^6 / M = 4096 fills the
chip by itself -- stays sequential (gptorch_amd/models/gpr.py:_vfe_groups decides).
"""
import math

import torch

from .. import _native, _ops
from . import sparse_gpr as _sg

_ptr, _stream, round_up = _ops._ptr, _ops._stream, _ops.round_up
LADDER_CLIMBS = 0  # factorisations that needed the jitter ladder so far (diagnostics: tools/vfe_batched_bench.py)


def supported(n, m):
    """the shapes the lock-step form covers (see the module docstring): what _vfe_forward would run as ONE chunk with plain
    accumulation and the recursive right-solve."""
    nc = round_up(n, _ops.LEAF)
    if n <= 0 or m <= 0 or nc > round_up(_sg.CHUNK_ROWS, _ops.LEAF) or nc >= 4096 * _sg.SPLIT_K:
        return False
    if (m >= _sg.BLOCKED_SOLVE_MIN_M or m >= _sg.INVERSE_MIN_M) and n >= 4 * m:
        return False
    return _sg.SHARD_GROUP is None


def per_model_bytes(n, m, dy):
    nc, ld = round_up(n, _ops.LEAF), round_up(m + dy, _ops.LEAF) + _ops.LEAF
    return 8 * (4 * (nc + 16) * ld + 14 * (ld + 16) * ld)


def _check(st, what):
    _native.check(st, what)


class _BState:
    """what one lock-step evaluation leaves behind for its backward"""
    pass


def _z3(B, rows, cols, dev):
    return _ops.zeros(B * rows, cols, dev).view(B, rows, cols)


def _potrf_b(fb):
    _check(_native.lib().gpn_potrf_lower_batched(_stream(fb.A.device), _ptr(fb.A), fb.n, fb.e, fb.ld, fb.sA, _ptr(fb.winv), fb.sW,
                                                 _ptr(fb.info), fb.batch), "gpn_potrf_lower_batched")


def _climb(fb, fill):
    """functions.py:20-43 for the factors of a FactorBatch whose first (plain) attempt has been enqueued: the failing problems
    retry TOGETHER with +10^(-tries+i) I, i = 0 .. tries-1 -- fill(sub, idx, jitter) assembles problems idx into the FactorBatch
    `sub` -- and every factor that succeeds is copied into its slot of fb.  One read-back of `info` per attempt."""
    global LADDER_CLIMBS

    def failing(info):
        if any(v < 0 for v in info):
            raise _ops.NativeError("factorisation reported the internal status %d (not a property of the matrix)" % min(info))
        return [j for j, v in enumerate(info) if v != 0]

    bad = failing(fb.info.tolist())
    if not bad:
        return
    LADDER_CLIMBS += len(bad)
    tries = int(_ops.JITTER_TRIES)
    A3, W = fb.A.view(fb.batch, fb.rows, fb.ld), fb.winv.view(fb.batch, fb.sW)
    for i in range(tries):
        sub = _ops.FactorBatch(len(bad), fb.n, fb.e, fb.A.device)
        fill(sub, bad, 10.0 ** (-tries + i))
        _potrf_b(sub)
        still = set(failing(sub.info.tolist()))
        S3, SW = sub.A.view(sub.batch, sub.rows, sub.ld), sub.winv.view(sub.batch, sub.sW)
        for j, b in enumerate(bad):
            if j not in still:
                A3[b].copy_(S3[j])
                W[b].copy_(SW[j])
        bad = [b for j, b in enumerate(bad) if j in still]
        if not bad:
            fb.info.zero_()
            return
    raise RuntimeError("Max tries exceeded.")


def _gemm_b(A, lda, sA, Bm, ldb, sB, C, ldc, sC, M, N, K, batch, alpha=1.0, alphas=None, beta=0.0, lower=False, tri=0, dev=None):
    """one strided-batch contraction; alphas: a device vector of per-problem scales (gpn_gemm_nt_batched_scaled)"""
    lib = _native.lib()
    if alphas is None:
        _check(lib.gpn_gemm_nt_batched(_stream(dev), M, N, K, alpha, A, lda, sA, Bm, ldb, sB, beta, C, ldc, sC,
                                       1 if lower else 0, tri, batch), "gpn_gemm_nt_batched")
    else:
        _check(lib.gpn_gemm_nt_batched_scaled(_stream(dev), M, N, K, _ptr(alphas), A, lda, sA, Bm, ldb, sB, beta, C, ldc, sC,
                                              1 if lower else 0, tri, batch), "gpn_gemm_nt_batched_scaled")


def _forward(kind, var, ls, nz, s2, Z, X, Y):
    """sparse_gpr._vfe_forward for B models: var [B], ls [B, nls], nz [B] (device), s2 = nz.tolist(), Z [B, m, d] contiguous,
    X [n, d] (shared) or [B, n, d], Y [n, dy] or [B, n, dy]  ->  _BState."""
    lib = _native.lib()
    dev = Z.device
    stream = _stream(dev)
    B, m, d = Z.shape
    n, dy, nls = X.shape[-2], Y.shape[-1], ls.shape[1]
    k = _ops.KINDS[kind]
    sX = 0 if X.dim() == 2 else n * d
    st = _BState()
    st.B, st.n, st.m, st.d, st.dy, st.nls, st.kind, st.s2 = B, n, m, d, dy, nls, kind, s2
    # host-side scalars of the sequential code, per model, as ONE table (what `x / s` is for a Python float s: x * (1 / s))
    p = dy
    tab = torch.tensor([[1.0 / s for s in s2],                                   # 0: 1 / s
                        [1.0 / (s * s) for s in s2],                             # 1: 1 / s^2
                        [0.5 * p * n * math.log(s) for s in s2],                 # 2: p N / 2 log s
                        [0.5 * p / s for s in s2],                               # 3: p / (2 s)
                        [0.5 * p * n / s for s in s2],                           # 4: p N / (2 s)
                        [(-0.5 * p / s) * n for s in s2]],                       # 5: -p N / (2 s)   (d tr Kff / d variance = N)
                       dtype=torch.float64).to(dev)
    st.tab = tab
    inv_s = tab[0]
    # L_b = chol K(Z_b)
    f_uu = _ops.FactorBatch(B, m, 0, dev)
    _check(lib.gpn_kernel_matrix_batched(stream, k, B, _ptr(Z), m * d, m, None, 0, m, d, _ptr(var), _ptr(ls), nls, None,
                                         _ops.GPN_LOWER, _ptr(f_uu.A), f_uu.ld, f_uu.sA), "gpn_kernel_matrix_batched")
    _potrf_b(f_uu)

    def fill_uu(sub, idx, jitter):
        ix = torch.tensor(idx, device=dev)
        Zs, vs, lss = Z[ix].contiguous(), var[ix].contiguous(), ls[ix].contiguous()
        nzs = torch.full((len(idx),), jitter, dtype=torch.float64, device=dev)
        _check(lib.gpn_kernel_matrix_batched(stream, k, len(idx), _ptr(Zs), m * d, m, None, 0, m, d, _ptr(vs), _ptr(lss), nls, _ptr(nzs),
                                             _ops.GPN_LOWER, _ptr(sub.A), sub.ld, sub.sA), "gpn_kernel_matrix_batched")
    _climb(f_uu, fill_uu)
    st.f_uu = f_uu
    # A_b^T = K(x, Z_b) L_b^-T, A_b = its transpose
    nc, mp, pp = round_up(n, _ops.LEAF), round_up(m, 16), round_up(dy, 16)
    ld = f_uu.ld
    At = _z3(B, nc + 16, ld, dev)
    _check(lib.gpn_kernel_matrix_batched(stream, k, B, _ptr(X), sX, n, _ptr(Z), m * d, m, d, _ptr(var), _ptr(ls), nls, None,
                                         _ops.GPN_FULL, _ptr(At), ld, (nc + 16) * ld), "gpn_kernel_matrix_batched")
    _check(lib.gpn_trsm_right_lt_batched(stream, _ptr(f_uu.A), m, ld, f_uu.sA, _ptr(f_uu.winv), f_uu.sW, _ptr(At), n, ld,
                                         (nc + 16) * ld, B), "gpn_trsm_right_lt_batched")
    A = _z3(B, mp, nc, dev)
    A[:, :m, :n] = At[:, :n, :m].transpose(1, 2)
    del At
    shared_y = Y.dim() == 2
    errT = torch.zeros((pp, nc) if shared_y else (B, pp, nc), dtype=torch.float64, device=dev)
    errT[..., :dy, :n] = Y.transpose(-1, -2)
    kp = round_up(n, 16)
    fB = _ops.FactorBatch(B, m, dy, dev)
    AAT = torch.zeros_like(fB.A).view(B, fB.rows, fB.ld)
    ks = _sg.SYRK_K_SLICE or kp
    for k0 in range(0, kp, ks):
        kk = min(ks, kp - k0)
        _gemm_b(A.data_ptr() + 8 * k0, nc, mp * nc, A.data_ptr() + 8 * k0, nc, mp * nc, _ptr(AAT), fB.ld, fB.sA, m, m, kk, B,
                alphas=inv_s, beta=(0.0 if k0 == 0 else 1.0), lower=True, dev=dev)
    Aerr = torch.zeros(B, mp, dy, dtype=torch.float64, device=dev)
    _gemm_b(_ptr(A), nc, mp * nc, _ptr(errT), nc, 0 if shared_y else pp * nc, _ptr(Aerr), dy, mp * dy, m, dy, kp, B, dev=dev)
    del A
    # the scalar sums: the kernel of the sequential code (_ops.dot2d / diag_sum), one launch over the models
    Yc = Y if Y.stride(-1) == 1 else Y.contiguous()
    if shared_y:
        yy = _ops.dot2d(Yc, Yc).expand(B)
    else:
        yy = _ops.dot2d_raw(_ptr(Yc), Yc.stride(1), Yc.stride(0), _ptr(Yc), Yc.stride(1), Yc.stride(0), n, dy, B, dev)
    st.yy, st.trkff = yy, n * var
    st.AAT, st.Aerr = AAT, Aerr
    st.tr = _ops.dot2d_raw(_ptr(AAT), fB.ld + 1, fB.sA, None, 0, 0, m, 1, B, dev)
    # B_b = AAT_b + I = LB LB^T with (A err)^T as extra rows
    A3 = fB.A.view(B, fB.rows, fB.ld)
    A3.copy_(AAT)
    A3.diagonal(dim1=1, dim2=2)[:, :m].add_(1.0)
    A3[:, m:m + dy, :m] = Aerr[:, :m, :].transpose(1, 2)
    _potrf_b(fB)

    def fill_B(sub, idx, jitter):
        ix = torch.tensor(idx, device=dev)
        S3 = sub.A.view(sub.batch, sub.rows, sub.ld)
        S3.copy_(AAT[ix])
        S3.diagonal(dim1=1, dim2=2)[:, :m].add_(1.0 + jitter)
        S3[:, m:m + dy, :m] = Aerr[ix][:, :m, :].transpose(1, 2)
    _climb(fB, fill_B)
    _check(lib.gpn_lml_reduce_batched(stream, _ptr(fB.A), m, dy, fB.ld, fB.sA, _ptr(fB.out), B), "gpn_lml_reduce_batched")
    st.fB, st.terms = fB, fB.out
    return st


def _elbo(st):
    """sparse_gpr._elbo, operation by operation, over the B models"""
    p, n, tab, t = st.dy, st.n, st.tab, st.terms
    elbo = -0.5 * p * n * math.log(2.0 * math.pi)
    elbo = elbo - p * t[:, 0]
    elbo = elbo - tab[2]
    elbo = elbo - (0.5 * (st.yy + p * st.trkff)) * tab[0]
    elbo = elbo + (0.5 * t[:, 1]) * tab[1]
    elbo = elbo + 0.5 * p * st.tr
    return elbo


def _sandwich(U, W, m):
    B, rows, ld = U.shape
    kp = round_up(m, 16)
    dev = U.device
    T = torch.zeros_like(U)
    _gemm_b(_ptr(U), ld, rows * ld, _ptr(W), ld, rows * ld, _ptr(T), ld, rows * ld, m, m, kp, B, tri=_ops.TRI_A_UPPER, dev=dev)
    R = torch.zeros_like(U)
    _gemm_b(_ptr(T), ld, rows * ld, _ptr(U), ld, rows * ld, _ptr(R), ld, rows * ld, m, m, kp, B, tri=_ops.TRI_B_UPPER, dev=dev)
    return R


def _upper_inverse(f):
    """_backward._upper_inverse for a FactorBatch -> [B, rows, ld]"""
    lib = _native.lib()
    dev = f.A.device
    U = _z3(f.batch, f.rows, f.ld, dev)
    S = _z3(f.batch, f.rows, f.ld, dev) if f.n > 2 * _ops.LEAF else None
    _check(lib.gpn_trtri_upper_batched(_stream(dev), _ptr(f.A), f.n, f.ld, f.sA, _ptr(f.winv), f.sW, _ptr(U), f.ld, f.rows * f.ld,
                                       _ptr(S), f.ld, f.rows * f.ld, f.batch), "gpn_trtri_upper_batched")
    if S is not None:
        S.record_stream(torch.cuda.current_stream(dev))
    return U


def _backward(st, var, ls, Z, X, Y):
    """sparse_gpr._vfe_backward for the B models of `st` -> (g_var [B], g_ls [B, nls], g_noise [B], g_Z [B, m, d]) w.r.t. the
    CONSTRAINED variance / length-scales / noise and the inducing points."""
    lib = _native.lib()
    dev = Z.device
    stream = _stream(dev)
    B, n, m, d, p, nls = st.B, st.n, st.m, st.d, st.dy, st.nls
    k = _ops.KINDS[st.kind]
    f_uu, fB, tab = st.f_uu, st.fB, st.tab
    inv_s = tab[0]
    mp, pp = round_up(m, 16), round_up(p, 16)
    ld, rows = f_uu.ld, f_uu.rows
    sX = 0 if X.dim() == 2 else n * d
    U = _upper_inverse(f_uu)                                               # L^-T
    UB = _upper_inverse(fB)                                                # LB^-T
    Kinv = torch.empty(B, round_up(max(m, 1), 64), fB.ld, dtype=torch.float64, device=dev)
    _gemm_b(_ptr(UB), fB.ld, fB.sA, _ptr(UB), fB.ld, fB.sA, _ptr(Kinv), fB.ld, Kinv.stride(0), m, m, round_up(m, 16), B, lower=True,
            tri=_ops.TRI_A_UPPER | _ops.TRI_B_UPPER, dev=dev)
    Binv = Kinv[:, :m, :m]
    Binv = torch.tril(Binv) + torch.tril(Binv, -1).transpose(1, 2)
    Bd = torch.tril(st.AAT[:, :m, :m]) + torch.tril(st.AAT[:, :m, :m], -1).transpose(1, 2)
    Bd.diagonal(dim1=1, dim2=2).add_(1.0)
    eye = torch.eye(m, dtype=torch.float64, device=dev)
    bt = torch.zeros(B, pp, ld, dtype=torch.float64, device=dev)
    _gemm_b(fB.A.data_ptr() + 8 * m * fB.ld, fB.ld, fB.sA, _ptr(UB), fB.ld, fB.sA, _ptr(bt), ld, pp * ld, p, m, mp, B,
            alphas=inv_s, tri=_ops.TRI_B_UPPER, dev=dev)
    beta = torch.zeros(B, mp, pp, dtype=torch.float64, device=dev)
    beta[:, :m, :p] = bt[:, :p, :m].transpose(1, 2)
    bbT = torch.empty(B, m, m, dtype=torch.float64, device=dev)
    _gemm_b(_ptr(beta), pp, mp * pp, _ptr(beta), pp, mp * pp, _ptr(bbT), m, m * m, m, m, pp, B, dev=dev)
    W = torch.zeros_like(U)
    W[:, :m, :m] = p * (eye - Binv) - bbT
    P = _sandwich(U, W, m)                                           # s * dF/dKuf = P Kuf + gamma err^T
    W[:, :m, :m] = 0.5 * p * (2.0 * eye - Binv - Bd) - 0.5 * bbT
    Guu = _sandwich(U, W, m)                                         # dF/dKuu
    gt = torch.empty(B, p, m, dtype=torch.float64, device=dev)
    _gemm_b(_ptr(bt), ld, pp * ld, _ptr(U), ld, rows * ld, _ptr(gt), m, p * m, p, m, mp, B, tri=_ops.TRI_B_UPPER, dev=dev)

    g_var = torch.zeros(B, dtype=torch.float64, device=dev)
    g_ls = torch.zeros(B, nls, dtype=torch.float64, device=dev)
    g_Z = torch.zeros(B, m, d, dtype=torch.float64, device=dev)

    def sweeps(Xp, sXp, rows_x, G, ldg, sG, scale):
        X2 = None if Xp is None else _ptr(Z)
        xp = _ptr(Z) if Xp is None else Xp
        work = torch.empty(max(1, B * int(lib.gpn_grad_work_bytes(rows_x, m, nls, 0)) // 8), dtype=torch.float64, device=dev)
        out = torch.empty(B, 1 + nls, dtype=torch.float64, device=dev)
        _check(lib.gpn_kernel_grad_batched(stream, k, B, xp, sXp, rows_x, X2, m * d, m, d, _ptr(var), _ptr(ls), nls, G, ldg, sG,
                                           _ptr(work), _ptr(out)), "gpn_kernel_grad_batched")
        g_var.add_(out[:, 0])
        g_ls.add_(out[:, 1:])
        work2 = torch.empty(max(1, B * int(lib.gpn_grad_x2_work_bytes(rows_x, m, d)) // 8), dtype=torch.float64, device=dev)
        _check(lib.gpn_kernel_grad_x2_batched(stream, k, B, xp, sXp, rows_x, _ptr(Z), m * d, m, d, _ptr(var), _ptr(ls), nls, G, ldg, sG,
                                              scale, 1, _ptr(work2), _ptr(g_Z)), "gpn_kernel_grad_x2_batched")

    # K(Z, Z) part
    sweeps(None, m * d, m, _ptr(Guu), ld, rows * ld, 2.0)
    # K(x, Z) part:  G_b = 1/s [K(x, Z_b) | err] [P_b | gamma_b]^T
    ldk = mp + pp
    Bq = torch.zeros(B, mp, ldk, dtype=torch.float64, device=dev)
    Bq[:, :m, :m] = P[:, :m, :m]
    Bq[:, :m, mp:mp + p] = gt[:, :p, :m].transpose(1, 2)
    nc = round_up(n, _ops.LEAF)
    Kx = _z3(B, nc + 16, ldk, dev)
    _check(lib.gpn_kernel_matrix_batched(stream, k, B, _ptr(X), sX, n, _ptr(Z), m * d, m, d, _ptr(var), _ptr(ls), nls, None,
                                         _ops.GPN_FULL, _ptr(Kx), ldk, (nc + 16) * ldk), "gpn_kernel_matrix_batched")
    Kx[:, :n, mp:mp + p] = Y
    G = torch.empty(B, nc, mp, dtype=torch.float64, device=dev)
    _gemm_b(_ptr(Kx), ldk, (nc + 16) * ldk, _ptr(Bq), ldk, mp * ldk, _ptr(G), mp, nc * mp, n, m, ldk, B, alphas=inv_s, dev=dev)
    sweeps(_ptr(X), sX, n, _ptr(G), mp, nc * mp, 1.0)
    g_var.add_(tab[5])                                                     # -p/(2s) d tr Kff

    t = st.terms
    c2 = t[:, 1] * tab[1]
    s_ba = _ops.dot2d_raw(_ptr(beta), pp, mp * pp, _ptr(st.Aerr), p, mp * p, m, p, B, dev)
    s_bb = _ops.dot2d_raw(_ptr(beta), pp, mp * pp, _ptr(beta), pp, mp * pp, m, p, B, dev)
    trb = _ops.dot2d_raw(_ptr(Binv), m + 1, m * m, None, 0, 0, m, 1, B, dev)
    quad = s_ba * tab[0] - s_bb                                            # beta^T (B - I) beta
    g_noise = tab[3] * (m - trb) - c2 * tab[0] + (0.5 * quad) * tab[0] - (0.5 * p * st.tr) * tab[0] \
        - tab[4] + (0.5 * (st.yy + p * st.trkff)) * tab[1]
    return g_var, g_ls, g_noise, g_Z


class BatchedVFEBound(torch.autograd.Function):
    """The bounds of B lock-step VFE models as ONE autograd node over (variance [B], length_scales [B, nls], noise [B],
    Z [B, M, D]); sparse_gpr._VFEBound model by model, bit for bit."""

    @staticmethod
    def forward(ctx, variance, length_scales, noise, Z, kind, X, Y):
        var, ls, nz = _ops._c(variance.detach()), _ops._c(length_scales.detach()), _ops._c(noise.detach())
        Zc = _ops._c(Z.detach())
        s2 = [float(v) for v in nz.tolist()]                               # (sparse_gpr._VFEBound reads noise.item() per model)
        st = _forward(kind, var, ls, nz, s2, Zc, X, Y)
        elbo = _elbo(st)
        ctx.st, ctx.X, ctx.Y = st, X, Y
        ctx.save_for_backward(var, ls, Zc)
        ctx.shapes = (variance.shape, length_scales.shape, noise.shape, Z.shape)
        return elbo

    @staticmethod
    def backward(ctx, grad_out):
        var, ls, Zc = ctx.saved_tensors
        st, X, Y = ctx.st, ctx.X, ctx.Y
        g_var, g_ls, g_noise, g_Z = _backward(st, var, ls, Zc, X, Y)
        g = grad_out
        sv, sl, sn, sz = ctx.shapes
        return ((g * g_var).reshape(sv), (g[:, None] * g_ls).reshape(sl), (g * g_noise).reshape(sn),
                (g[:, None, None] * g_Z).reshape(sz), None, None, None)

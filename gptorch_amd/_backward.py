"""
Closed-form backward of the GPR log marginal likelihood and of Kernel.K, over
the native library (verified against autograd through the
reference's op chain in tests/golden/make_golden.py):

    a = Kyy^-1 (y - m),   G = 1/2 (a a^T - dy Kyy^-1)
    dLML/dtheta = sum_ij G_ij dKyy_ij/dtheta,   dLML/d(y - m) = -a

Kyy^-1 comes from the Cholesky factor already held by the forward pass:
U = L^-T by a recursive triangular inversion made of the same fp64 MFMA
contraction + in-place right-solves as the factorisation (gpn_trtri_upper),
then Kyy^-1 = U U^T as one SYRK whose K range is clipped per tile to the
non-zero part of U, a^T = alpha^T U^T as one skinny contraction, and a single
HBM-bound sweep that re-computes dK/dtheta from the points (gpn_lml_grad).
The reference spends ~2 N^3 flops in CholeskyBackward0 here; this
is 2/3 N^3.
"""
import torch

from . import _native, _ops
from ._ops import _c, _ptr, _req, _stream, round_up


def _upper_inverse(f, workspace=True):
    """U = L^-T in a zeroed [rows, ld] buffer (workspace: the level-parallel variant, which
    needs a second zeroed buffer of the same shape while it runs)."""
    U = _ops.zeros(f.rows, f.ld, f.device)
    if workspace and f.n > 2 * _ops.LEAF:
        S = _ops.zeros(f.rows, f.ld, f.device)
        st = _native.lib().gpn_trtri_upper_ws(_stream(f.device), _ptr(f.A), f.n, f.ld, _ptr(f.winv), _ptr(U), f.ld,
                                              _ptr(S), f.ld)
        _native.check(st, "gpn_trtri_upper_ws")
        S.record_stream(torch.cuda.current_stream(f.device))
        return U
    st = _native.lib().gpn_trtri_upper(_stream(f.device), _ptr(f.A), f.n, f.ld, _ptr(f.winv), _ptr(U), f.ld)
    _native.check(st, "gpn_trtri_upper")
    return U


def _kinv_lower(f, U):
    """lower triangle of (L L^T)^-1 = U U^T in an [n_pad, ld] buffer (upper part unspecified)."""
    n = f.n
    Kinv = torch.empty(round_up(max(n, 1), 64), f.ld, dtype=torch.float64, device=f.device)
    _ops.gemm_nt(U, U, n, n, round_up(n, 16), C=Kinv, lower=True, tri=_ops.TRI_A_UPPER | _ops.TRI_B_UPPER)
    return Kinv


def lml_backward(kind, X, variance, length_scales, noise, f):
    """-> (dLML/dvariance [1], dLML/dlength_scales [nls], dLML/dnoise [1], dLML/dR [n, dy]):
    ONE library call (gpn_lml_backward: U = L^-T, Kyy^-1 = U U^T, a = U alpha, gradient sweep)."""
    _req(X, variance, length_scales)
    n, dy = f.n, f.e
    nls = length_scales.numel()
    lib = _native.lib()
    work = torch.empty(max(1, int(lib.gpn_lml_backward_work_bytes(n, dy, nls)) // 8), dtype=torch.float64, device=f.device)
    out = torch.empty(2 + nls, dtype=torch.float64, device=f.device)
    g_R = torch.empty(n, dy, dtype=torch.float64, device=f.device)
    Xc = _c(X.detach())
    st = lib.gpn_lml_backward(_stream(f.device), _ops.KINDS[kind], _ptr(Xc), n, Xc.shape[1],
                              _ptr(_c(variance.detach())), _ptr(_c(length_scales.detach())), nls,
                              _ptr(f.A), f.ld, _ptr(f.winv), dy, _ptr(work), _ptr(out), _ptr(g_R))
    _native.check(st, "gpn_lml_backward")
    return out[0:1], out[1:1 + nls], out[1 + nls:2 + nls], g_R


def _rowmajor(t):
    """a 2-D tensor usable as a row-major matrix with a leading dimension (no copy for
    row-slices / column-prefixes of a wider buffer)."""
    t = t.detach()
    if t.dim() == 2 and t.stride(1) == 1 and t.stride(0) >= t.shape[1]:
        return t
    return t.contiguous()


def kernel_backward(kind, X, X2, variance, length_scales, gK):
    """-> (sum gK*dK/dvariance [1], sum gK*dK/dlength_scales [nls])."""
    _req(X, X2, variance, length_scales, gK)
    lib = _native.lib()
    Xc = _c(X.detach())
    n, d = Xc.shape
    X2c = None if X2 is None else _c(X2.detach())
    m = n if X2c is None else X2c.shape[0]
    nls = length_scales.numel()
    g = _rowmajor(gK)
    work = torch.empty(max(1, int(lib.gpn_grad_work_bytes(n, m, nls, 0)) // 8), dtype=torch.float64, device=X.device)
    out = torch.empty(1 + nls, dtype=torch.float64, device=X.device)
    st = lib.gpn_kernel_grad(_stream(X.device), _ops.KINDS[kind], _ptr(Xc), n, _ptr(X2c), m, d,
                             _ptr(_c(variance.detach())), _ptr(_c(length_scales.detach())), nls,
                             _ptr(g), g.stride(0), _ptr(work), _ptr(out))
    _native.check(st, "gpn_kernel_grad")
    return out[0:1], out[1:1 + nls]


def kernel_backward_x2(kind, X, X2, variance, length_scales, gK, scale=1.0, out=None):
    """-> d sum(gK * K(X, X2)) / dX2  [m, d]  (out given: accumulated into it).
    gK is [n, m]; for dX pass (X2, X, gK^T); for a symmetric K(Z, Z) with symmetric gK pass
    X = X2 = Z and scale = 2."""
    _req(X, X2, variance, length_scales, gK)
    lib = _native.lib()
    Xc, X2c = _c(X.detach()), _c(X2.detach())
    n, d = Xc.shape
    m = X2c.shape[0]
    nls = length_scales.numel()
    g = _rowmajor(gK)
    work = torch.empty(max(1, int(lib.gpn_grad_x2_work_bytes(n, m, d)) // 8), dtype=torch.float64, device=X.device)
    acc = 1
    if out is None:
        out, acc = torch.empty(m, d, dtype=torch.float64, device=X.device), 0
    st = lib.gpn_kernel_grad_x2(_stream(X.device), _ops.KINDS[kind], _ptr(Xc), n, _ptr(X2c), m, d,
                                _ptr(_c(variance.detach())), _ptr(_c(length_scales.detach())), nls,
                                _ptr(g), g.stride(0), float(scale), acc, _ptr(work), _ptr(out))
    _native.check(st, "gpn_kernel_grad_x2")
    return out


def potri_full(f):
    """dense symmetric (L L^T)^-1 [n, n] (functions.cholesky_inverse, functions.py:50-54)."""
    n = f.n
    Kinv = _kinv_lower(f, _upper_inverse(f))[:n, :n]
    low = torch.tril(Kinv)
    return low + torch.tril(Kinv, -1).t()

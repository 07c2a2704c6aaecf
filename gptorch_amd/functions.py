"""
Numerical primitives with the call surface of gptorch/functions.py, backed by
the native library and DIFFERENTIABLE like the torch ops the reference wraps:

  cholesky(x)                       functions.py:46-47 (+ jitter ladder 20-43)
  trtrs(b, a, lower=True)           functions.py:71-76
  lt_log_determinant(L)             functions.py:61-68
  cholesky_inverse(L, upper=False)  functions.py:50-54
  inverse(x)                        functions.py:57-58
  jit_op(op, x)                     functions.py:20-43

Each is one autograd node: the forward is the native factorisation / solve, the backward a
closed form built from the same native pieces (explicit L^-T from the level-parallel triangular
inversion + fp64 MFMA contractions).  `cholesky` returns a dense lower-triangular tensor like
torch.cholesky and remembers the native factor (padded buffer + inverted diagonal blocks) on
the returned tensor -- together with the tensor's version counter, so an in-place edit of L
invalidates it -- so that a following `trtrs(b, L)` reuses it.
"""
import torch

from . import _ops
from ._ops import _native, _ptr, _stream


def jit_op(op, x, max_tries: int = 10, verbose: bool = False):
    """functions.py:20-43 for callers that bring their own `op`: the ladder itself is
    `_ops._ladder` (the one `cholesky` below climbs on the device-side info word); here an
    attempt "fails" when `op` raises."""
    out = []

    def attempt(jitter):
        xj = x if jitter is None else x + jitter * torch.eye(*x.shape, dtype=x.dtype, device=x.device)
        try:
            out[:] = [op(xj)]
            return 0
        except Exception as exc:          # the reference catches Exception on the initial try (functions.py:30) and
            if jitter is not None and not isinstance(exc, RuntimeError):     # RuntimeError on the jittered ones (:38)
                raise
            if verbose:
                print("Op {} failed ({})".format(getattr(op, "__name__", "op"), "initial try" if jitter is None else "jitter %g" % jitter))
            return 1
    _ops._ladder(attempt, tries=max_tries)
    return out[0]


def _remember(L, f):
    L._gpn_factor = f
    L._gpn_version = L._version
    return L


def _upper_inv(f):
    """U = L^-T as a dense [n, n] tensor (cached on the factor for its generation)."""
    from . import _backward
    U = getattr(f, "_u_dense", None)
    if U is None or getattr(f, "_u_generation", None) != f.generation:
        U = _backward._upper_inverse(f)[:f.n, :f.n].contiguous()
        f._u_dense, f._u_generation = U, f.generation
    return U


class _Cholesky(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x):
        f = _ops.cholesky_factor(x)
        L = f.lower()
        ctx.factor = f
        ctx.save_for_backward(L)
        return L

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, gL):
        # A_bar = L^-T Phi(L^T L_bar) L^-1, symmetrised (Phi: lower triangle, diagonal halved)
        L, = ctx.saved_tensors
        n = L.shape[0]
        if n == 0:
            return gL
        U = _upper_inv(ctx.factor)
        P = _ops.matmul_nt(_ops.transpose(L), _ops.transpose(torch.tril(gL)))          # L^T L_bar
        P = torch.tril(P)
        P.diagonal().mul_(0.5)
        M = _ops.matmul_nt(P, U)                                                        # Phi U^T
        A = _ops.matmul_nt(U, _ops.transpose(M))                                        # U Phi U^T
        return 0.5 * (A + A.t())


def cholesky(x: torch.Tensor) -> torch.Tensor:
    if torch.is_grad_enabled() and x.requires_grad:
        L = _Cholesky.apply(x)
        f = L.grad_fn.factor
    else:
        f = _ops.cholesky_factor(x)
        L = f.lower()
    return _remember(L, f)


def _factor_of(a, transpose=False):
    """native factor behind a lower-triangular tensor (cached by cholesky() while the tensor is
    unmodified, or built from `a` itself: copy + inversion of its diagonal blocks)."""
    f = getattr(a, "_gpn_factor", None)
    if (not transpose and f is not None and f.n == a.shape[0] and f.device == a.device
            and getattr(a, "_gpn_version", None) == a._version):
        return f
    _ops._req(a)
    n = a.shape[0]
    f = _ops.Factor(n, 0, a.device)
    if n:
        lib = _native.lib()
        src = _ops.transpose(a) if transpose else _ops._c(a.detach())
        _native.check(lib.gpn_copy_matrix(_stream(a.device), _ptr(src), n, n, n, _ptr(f.A), f.ld, 1), "gpn_copy_matrix")
        _native.check(lib.gpn_trtri_diag(_stream(a.device), _ptr(f.A), n, f.ld, _ptr(f.winv), _ptr(f.info)),
                      "gpn_trtri_diag")
        bad = int(f.info.item())
        if bad:
            raise RuntimeError("trtrs: the triangular matrix is singular (zero pivot %d)" % bad)
    return f


class _Trtrs(torch.autograd.Function):
    @staticmethod
    def forward(ctx, b, a, lower):
        # lower: x = L^-1 b by the native right-solve on b^T.  upper (a = L^T): x = L^-T b as ONE
        # contraction with the explicit U = L^-T of the transposed (lower) matrix.
        f = _factor_of(a, transpose=not lower)
        x = _ops.trtrs_lower(b, f) if lower else _ops.matmul_nt(_upper_inv(f), _ops.transpose(b))
        ctx.factor, ctx.lower = f, lower
        ctx.save_for_backward(x)
        return x

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, gx):
        x, = ctx.saved_tensors
        f = ctx.factor
        # b_bar = a^-T x_bar;  a_bar = -tri(b_bar x^T)
        if ctx.lower:
            gb = _ops.matmul_nt(_upper_inv(f), _ops.transpose(gx))
        else:
            gb = _ops.trtrs_lower(gx.contiguous(), f)
        ga = None
        if ctx.needs_input_grad[1]:
            ga = -_ops.matmul_nt(gb, x)
            ga = torch.tril(ga) if ctx.lower else torch.triu(ga)
        return gb, ga, None


def trtrs(b: torch.Tensor, a: torch.Tensor, lower=True) -> torch.Tensor:
    """Solve a x = b with triangular a (functions.py:71-76)."""
    if b.shape[0] != a.shape[0]:
        raise RuntimeError("trtrs: size mismatch")
    return _Trtrs.apply(b, a, bool(lower))


class _LogDet(torch.autograd.Function):
    @staticmethod
    def forward(ctx, L):
        f = getattr(L, "_gpn_factor", None)
        if not (f is not None and f.e == 0 and getattr(L, "_gpn_version", None) == L._version):
            f = _factor_of(L)
        ctx.save_for_backward(L)
        return f.lml_terms()[0].clone()

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, g):
        L, = ctx.saved_tensors
        return torch.diag_embed(g / L.diagonal())


def lt_log_determinant(L):
    """sum(log(diag(L))) (functions.py:61-68)."""
    return _LogDet.apply(L)


class _CholInverse(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, upper):
        from . import _backward
        f = _factor_of(x, transpose=upper)        # upper: x = L^T
        Y = _backward.potri_full(f)
        ctx.upper = upper
        ctx.save_for_backward(x, Y)
        return Y

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, gY):
        # Y = (L L^T)^-1:  L_bar = -tril((S + S^T) L),  S = Y Y_bar Y
        x, Y = ctx.saved_tensors
        S = _ops.matmul_nt(_ops.matmul_nt(Y, _ops.transpose(gY)), Y)     # Y symmetric: Y gY Y
        S = S + S.t()
        if ctx.upper:
            return -torch.triu(_ops.matmul_nt(x, S)), None               # U_bar = -triu(U (S + S^T))
        return -torch.tril(_ops.matmul_nt(S, _ops.transpose(x))), None


def cholesky_inverse(x: torch.Tensor, upper=False) -> torch.Tensor:
    """(L L^T)^-1 from the Cholesky factor, or (U^T U)^-1 from an upper one (functions.py:50-54)."""
    return _CholInverse.apply(x, bool(upper))


def inverse(x: torch.Tensor) -> torch.Tensor:
    """x^-1 through the jitter ladder (functions.py:57-58: `jit_op(torch.inverse, x)`).  The reference's callers invert
    covariance matrices; natively that is the Cholesky route -- `cholesky` (which climbs the ladder of functions.py:20-43 on
    the device-side info word, adding the same jitter to the diagonal that the reference would add before retrying
    torch.inverse) followed by `cholesky_inverse` -- two autograd nodes, no LU.  A matrix that is not symmetric has no
    native path: NotImplementedError (there is no CPU fallback in this package)."""
    if x.dim() != 2 or x.shape[0] != x.shape[1]:
        raise ValueError("inverse expects a square matrix")
    if not torch.equal(x.detach(), x.detach().t()):
        raise NotImplementedError("gptorch_amd.functions.inverse: symmetric positive definite matrices only "
                                  "(native Cholesky route; the reference's torch.inverse is a general LU)")
    return cholesky_inverse(cholesky(x))

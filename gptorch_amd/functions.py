"""
Numerical primitives with the call surface of gptorch/functions.py, backed by
the native library:

  cholesky(x)                 functions.py:46-47 (+ jitter ladder 20-43)
  trtrs(b, a, lower=True)     functions.py:71-76
  lt_log_determinant(L)       functions.py:61-68
  cholesky_inverse(L)         functions.py:50-54

`cholesky` returns a dense lower-triangular tensor like torch.cholesky, and
remembers the native factor (padded buffer + inverted diagonal blocks) on the
returned tensor so that a following `trtrs(b, L)` reuses it.
"""
import torch

from . import _ops
from ._ops import _native, _ptr, _stream


def jit_op(op, x, max_tries: int = 10, verbose: bool = False):
    """Generic retry-with-jitter wrapper (functions.py:20-43) for callers that pass
    their own `op`; cholesky() below implements the same ladder natively."""
    try:
        return op(x)
    except Exception:
        if verbose:
            print("Op {} failed (initial try)".format(op.__name__))
    for i in range(max_tries):
        try:
            this_jitter = 10.0 ** (-max_tries + i) * torch.eye(*x.shape, dtype=x.dtype, device=x.device)
            return op(x + this_jitter)
        except RuntimeError:
            if verbose:
                print("Op {} failed (try {} / {})".format(op.__name__, i + 1, max_tries))
    raise RuntimeError("Max tries exceeded.")


def cholesky(x: torch.Tensor) -> torch.Tensor:
    f = _ops.cholesky_factor(x)
    L = f.lower()
    L._gpn_factor = f
    return L


def _factor_of(a):
    """native factor behind a lower-triangular tensor (cached by cholesky(), or
    built from `a` itself: copy + inversion of its diagonal blocks)."""
    f = getattr(a, "_gpn_factor", None)
    if f is not None and f.n == a.shape[0] and f.device == a.device:
        return f
    _ops._req(a)
    n = a.shape[0]
    f = _ops.Factor(n, 0, a.device)
    if n:
        lib = _native.lib()
        src = _ops._c(a.detach())
        _native.check(lib.gpn_copy_matrix(_stream(a.device), _ptr(src), n, n, n, _ptr(f.A), f.ld, 1), "gpn_copy_matrix")
        _native.check(lib.gpn_trtri_diag(_stream(a.device), _ptr(f.A), n, f.ld, _ptr(f.winv), _ptr(f.info)),
                      "gpn_trtri_diag")
        bad = int(f.info.item())
        if bad:
            raise RuntimeError("trtrs: the triangular matrix is singular (zero pivot %d)" % bad)
    return f


def trtrs(b: torch.Tensor, a: torch.Tensor, lower=True) -> torch.Tensor:
    """Solve a x = b with triangular a.  lower=False is solved through the
    transpose-free identity only when `a` is the transpose of a cached factor."""
    if not lower:
        raise NotImplementedError(
            "gptorch_amd.functions.trtrs: upper-triangular solves are not on the GPR hot path "
            "(gpr.py uses lower solves only) and are not implemented natively")
    return _ops.trtrs_lower(b, _factor_of(a))


def lt_log_determinant(L):
    """sum(log(diag(L))) (functions.py:61-68)."""
    f = getattr(L, "_gpn_factor", None)
    if f is not None and f.e == 0:
        return f.lml_terms()[0]
    return _factor_of(L).lml_terms()[0]


def cholesky_inverse(x: torch.Tensor, upper=False) -> torch.Tensor:
    """(L L^T)^-1 from the Cholesky factor (functions.py:50-54)."""
    if upper:
        raise NotImplementedError("cholesky_inverse(upper=True) is not implemented natively")
    from . import _backward
    return _backward.potri_full(_factor_of(x))

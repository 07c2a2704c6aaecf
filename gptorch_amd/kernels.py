"""
Covariance functions with the call surface of gptorch/kernels.py.

Stationary kernels (Rbf / SquaredExponential, Matern52, Matern32, Exp /
Matern12) evaluate K(X, X2) in ONE native pass (csrc/kmat.hip): distance,
length-scale scaling, the kernel's epilogue -- nothing N x M is materialised in
between (the reference chains >= 8 elementwise N x M passes: util.py:82-88,
kernels.py:149-222).  Hyper-parameters are exp-transformed Params exactly as in
kernels.py:116-147, so gradients are w.r.t. the logs.
"""
import numpy as np
import torch

from . import _ops
from .model import Model
from .param import Param
from .settings import DefaultPositiveTransform
from .util import as_tensor, torch_dtype


def _k_shape(X, X2):
    return (X.size(0),) * 2 if X2 is None else (X.size(0), X2.size(0))


class Kernel(Model):
    """Base class (kernels.py:28-64)."""

    def __init__(self, input_dim):
        self.input_dim = int(input_dim)
        super().__init__()

    def __add__(self, other):
        return Sum(self, other)

    def __mul__(self, other):
        return Product(self, other)


class _StationaryK(torch.autograd.Function):
    """K(X, X2) with gradients w.r.t. the constrained variance / length-scales and, when
    asked for, the points themselves (the reference gets these from autograd through its
    elementwise chain, util.py:73-88 / kernels.py:149-222)."""

    @staticmethod
    def forward(ctx, variance, length_scales, X, X2, kind):
        K = _ops.kernel_matrix(kind, X, X2, variance, length_scales)
        ctx.kind = kind
        ctx.has_x2 = X2 is not None
        ctx.save_for_backward(variance, length_scales, X, X2 if X2 is not None else X)
        return K

    @staticmethod
    def backward(ctx, gK):
        from . import _backward
        variance, length_scales, X, X2 = ctx.saved_tensors
        g_var, g_ls = _backward.kernel_backward(ctx.kind, X, X2 if ctx.has_x2 else None, variance,
                                                length_scales, gK)
        g_x = g_x2 = None
        if ctx.needs_input_grad[2]:
            # rows of X enter as the first argument (and, for K(X), also as the second)
            g_x = _backward.kernel_backward_x2(ctx.kind, X2, X, variance, length_scales, _ops.transpose(gK))
            if not ctx.has_x2:
                _backward.kernel_backward_x2(ctx.kind, X, X, variance, length_scales, gK, out=g_x)
        if ctx.has_x2 and ctx.needs_input_grad[3]:
            g_x2 = _backward.kernel_backward_x2(ctx.kind, X, X2, variance, length_scales, gK)
        return g_var, g_ls, g_x, g_x2, None


class _SqDistPointGrad(torch.autograd.Function):
    """d sum(gK * r^2(X, Z)) / d(X or Z) as its own node, so that the SECOND derivative of
    util.squared_distance w.r.t. the points exists (test/test_util.py:78-106 pins it at r = 0):
    with w = 1/ell^2, R_i = sum_j gK_ij, C_j = sum_i gK_ij
        g_X = 2 w (R x - gK Z),      g_Z = 2 w (C z - gK^T X)
    forward = native sweep (gpn_kernel_grad_x2, kind SQDIST); backward = the closed-form derivative
    of the expressions above (native contractions + elementwise)."""

    @staticmethod
    def forward(ctx, gK, X, Z, ls, wrt_z):
        from . import _backward
        one = torch.ones(1, dtype=torch_dtype, device=X.device)
        if wrt_z:
            g = _backward.kernel_backward_x2("SqDist", X, Z, one, ls, gK)
        else:
            g = _backward.kernel_backward_x2("SqDist", Z, X, one, ls, _ops.transpose(gK))
        ctx.wrt_z = wrt_z
        ctx.save_for_backward(gK, X, Z, ls)
        return g

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, H):
        gK, X, Z, ls = ctx.saved_tensors
        w = (1.0 / (ls * ls)).expand(X.shape[1])
        Hw = H * w
        G = gK.t() if ctx.wrt_z else gK              # rows = the points differentiated in `forward`
        P, Q = (Z, X) if ctx.wrt_z else (X, Z)       # g = 2 w (rowsum(G) P - G Q)
        g_gK = g_P = g_Q = None
        if ctx.needs_input_grad[0]:
            t = 2.0 * ((Hw * P).sum(1, keepdim=True) - _ops.matmul_nt(Hw.contiguous(), Q))
            g_gK = t.t() if ctx.wrt_z else t
        g_P = 2.0 * G.sum(1, keepdim=True) * Hw
        g_Q = -2.0 * _ops.matmul_nt(_ops.transpose(G), _ops.transpose(Hw))
        g_X, g_Z = (g_Q, g_P) if ctx.wrt_z else (g_P, g_Q)
        return g_gK, g_X, g_Z, None, None


class _SqDist(torch.autograd.Function):
    """util.squared_distance / Stationary.squared_dist (util.py:73-88, kernels.py:149-159): r^2 by
    direct differences in the native assembly kernel (never negative, so the reference's
    clamp-and-detach is the identity), differentiable w.r.t. the points (twice) and the
    length-scales."""

    @staticmethod
    def forward(ctx, X, Z, ls):
        one = torch.ones(1, dtype=torch_dtype, device=X.device)
        ctx.save_for_backward(X, Z, ls)
        return _ops.kernel_matrix("SqDist", X, Z, one, ls)

    @staticmethod
    def backward(ctx, gK):
        from . import _backward
        X, Z, ls = ctx.saved_tensors
        g_x = _SqDistPointGrad.apply(gK, X, Z, ls, False) if ctx.needs_input_grad[0] else None
        g_z = _SqDistPointGrad.apply(gK, X, Z, ls, True) if ctx.needs_input_grad[1] else None
        g_ls = None
        if ctx.needs_input_grad[2]:
            one = torch.ones(1, dtype=torch_dtype, device=X.device)
            g_ls = _backward.kernel_backward("SqDist", X.detach(), Z.detach(), one, ls.detach(), gK.detach())[1]
        return g_x, g_z, g_ls


class Stationary(Kernel):
    """Kernels of r = ||(x - x') / ell||; ARD = one length-scale per input
    dimension (kernels.py:108-179)."""

    _kind = None

    def __init__(self, input_dim, variance=1.0, length_scales=None, ARD=False):
        super().__init__(input_dim)
        self.variance = Param(torch.tensor([float(variance)], dtype=torch_dtype),
                              transform=DefaultPositiveTransform())
        self.ARD = ARD
        if ARD:
            if length_scales is None:
                length_scales = np.ones(input_dim)
            elif isinstance(length_scales, np.ndarray):
                assert len(length_scales) == input_dim
            else:
                length_scales = length_scales * np.ones(input_dim)
            ls = torch.as_tensor(np.asarray(length_scales, dtype=np.float64)).clone()
        else:
            ls = torch.tensor([1.0 if length_scales is None else float(length_scales)], dtype=torch_dtype)
        self.length_scales = Param(ls, transform=DefaultPositiveTransform())

    def K(self, X, X2=None):
        if isinstance(X, np.ndarray):
            X = as_tensor(X).to(self.variance.device)
        if isinstance(X2, np.ndarray):
            X2 = as_tensor(X2).to(self.variance.device)
        return _StationaryK.apply(self.variance.transform(), self.length_scales.transform(), X, X2, self._kind)

    def Kdiag(self, X):
        """variance broadcast to [n] (kernels.py:174-179)."""
        if isinstance(X, np.ndarray):
            X = as_tensor(X)
        return self.variance.transform().expand(X.size(0))

    def squared_dist(self, X, X2):
        """scaled squared distance (kernels.py:149-159), differentiable w.r.t. the length-scales
        and the points."""
        return _SqDist.apply(X, X if X2 is None else X2, self.length_scales.transform())

    def dist(self, X, X2):
        """kernels.py:161-172."""
        return torch.sqrt(torch.clamp(self.squared_dist(X, X2), min=1e-40))


class Rbf(Stationary):
    """variance * exp(-r^2 / 2) (kernels.py:215-222)."""
    _kind = "Rbf"


SquaredExponential = Rbf


class Matern52(Stationary):
    """variance * (1 + sqrt5 r + 5/3 r^2) exp(-sqrt5 r) (kernels.py:204-212)."""
    _kind = "Matern52"


class Matern32(Stationary):
    """variance * (1 + sqrt3 r) exp(-sqrt3 r) (kernels.py:196-201)."""
    _kind = "Matern32"


class Exp(Stationary):
    """variance * exp(-r) (kernels.py:182-190)."""
    _kind = "Exp"


class Matern12(Exp):
    pass


# ---- combinators and input-independent kernels (thin glue) -----
class Periodic(Stationary):
    """variance * cos(r) (kernels.py:228-235)."""
    _kind = "Periodic"


class _LinearK(torch.autograd.Function):
    """K = (X * v) X2^T (kernels.py:258-262) as one NT fp64-MFMA contraction over the
    (zero-padded) input dimension; backward = three more contractions of the same form."""

    @staticmethod
    def forward(ctx, v, X, X2):
        n, d = X.shape
        other = X if X2 is None else X2
        m = other.shape[0]
        dp = _ops.round_up(d, 16)
        A = torch.zeros(_ops.round_up(n, 16), dp, dtype=torch_dtype, device=X.device)
        B = torch.zeros(_ops.round_up(m, 16), dp, dtype=torch_dtype, device=X.device)
        A[:n, :d] = X.detach() * v.detach()
        B[:m, :d] = other.detach()
        ctx.save_for_backward(v, X, other)
        ctx.symmetric = X2 is None
        return _ops.gemm_nt(A, B, n, m, dp)

    @staticmethod
    def backward(ctx, gK):
        v, X, other = ctx.saved_tensors
        n, d = X.shape
        m = other.shape[0]
        dev = X.device
        # G X2 [n, d] and G^T X [m, d]: NT contractions against the transposed point blocks
        mp, np_, dp = _ops.round_up(m, 16), _ops.round_up(n, 16), _ops.round_up(d, 16)
        G = torch.zeros(np_, mp, dtype=torch_dtype, device=dev)
        G[:n, :m] = gK
        Ot = torch.zeros(dp, mp, dtype=torch_dtype, device=dev)
        Ot[:d, :m] = other.detach().t()
        GX2 = _ops.gemm_nt(G, Ot, n, d, mp)                                # (G X2)[i, c]
        g_v = g_x = g_x2 = None
        if ctx.needs_input_grad[0]:
            g_v = (X.detach() * GX2).sum(0)
            if v.numel() == 1:
                g_v = g_v.sum().reshape(1)
        need_x2 = (not ctx.symmetric and ctx.needs_input_grad[2]) or (ctx.symmetric and ctx.needs_input_grad[1])
        if need_x2:
            Gt = torch.zeros(mp, np_, dtype=torch_dtype, device=dev)
            Gt[:m, :n] = gK.t()
            Xt = torch.zeros(dp, np_, dtype=torch_dtype, device=dev)
            Xt[:d, :n] = (X.detach() * v.detach()).t()
            GtX = _ops.gemm_nt(Gt, Xt, m, d, np_)                          # (G^T (X v))[j, c]
        if ctx.needs_input_grad[1]:
            g_x = GX2 * v.detach()
            if ctx.symmetric:
                g_x = g_x + GtX
        if not ctx.symmetric and ctx.needs_input_grad[2]:
            g_x2 = GtX
        return g_v, g_x, g_x2


class Linear(Kernel):
    """kernels.py:238-265: K = (X * variance) X2^T with one variance per input when ARD."""

    def __init__(self, input_dim, variance=1.0, ARD=None):
        super().__init__(input_dim)
        if ARD is None:
            ARD = np.asarray(variance).squeeze().shape != ()
        variance = variance * np.ones(input_dim)
        if variance.shape != (input_dim,):
            raise ValueError("shape of possibly-ARD param does not match input_dim")
        self.ARD = ARD
        self.variance = Param(torch.as_tensor(np.asarray(variance, dtype=np.float64)).clone(),
                              transform=DefaultPositiveTransform())

    def K(self, X, X2=None):
        if isinstance(X, np.ndarray):
            X = as_tensor(X).to(self.variance.device)
        if isinstance(X2, np.ndarray):
            X2 = as_tensor(X2).to(self.variance.device)
        return _LinearK.apply(self.variance.transform(), X, X2)

    def Kdiag(self, X):
        return torch.sum(X * X * self.variance.transform(), 1)


class Combination(Kernel):
    """kernels.py:268-283.  A tree of Sum / Product whose leaves are native terms (stationary kinds, Linear, Constant /
    Bias, White) is evaluated by ONE fused pass (csrc/kexpr.hip through gptorch_amd/_expr.py: the matrix is written
    once, the backward is one sweep per leaf); anything else -- a leaf without a native term, inputs that require
    gradients themselves -- composes the children's matrices the way the reference does."""

    def __init__(self, k1, k2):
        if not k1.input_dim == k2.input_dim:
            raise ValueError("Kernels must have same input dimension")
        super().__init__(k1.input_dim)
        self.kern1, self.kern2 = k1, k2

    def fused_program(self):
        """the expression program of this tree (gptorch_amd._expr.Program) or None; rebuilt on every call (cheap: a
        dozen Python objects) so that a tree edited after construction is never evaluated from a stale program."""
        from . import _expr
        return _expr.build(self)

    def _fused_K(self, X, X2):
        if isinstance(X, np.ndarray):
            X = as_tensor(X).to(self.kern1.variance.device)
        if isinstance(X2, np.ndarray):
            X2 = as_tensor(X2).to(self.kern1.variance.device)
        if not X.is_cuda or X.requires_grad or (X2 is not None and X2.requires_grad):
            return None
        from . import _expr
        prog = self.fused_program()
        if prog is None or not prog.grad_supported(X.shape[1]):
            return None
        return _expr.ExprK.apply(X, X2, prog, *prog.params())


class Sum(Combination):
    """kernels.py:286-295."""

    def K(self, X, X2=None):
        K = self._fused_K(X, X2)
        return K if K is not None else self.kern1.K(X, X2) + self.kern2.K(X, X2)

    def Kdiag(self, X):
        return self.kern1.Kdiag(X) + self.kern2.Kdiag(X)


class Product(Combination):
    """kernels.py:298-306."""

    def K(self, X, X2=None):
        K = self._fused_K(X, X2)
        return K if K is not None else self.kern1.K(X, X2) * self.kern2.K(X, X2)

    def Kdiag(self, X):
        return self.kern1.Kdiag(X) * self.kern2.Kdiag(X)


class Static(Kernel):
    """kernels.py:67-80."""

    def __init__(self, input_dim, variance=1.0):
        super().__init__(input_dim)
        self.variance = Param(torch.tensor([float(variance)], dtype=torch_dtype),
                              transform=DefaultPositiveTransform())

    def Kdiag(self, X):
        return self.variance.transform().expand(X.size(0))


class White(Static):
    """kernels.py:83-92."""

    def K(self, X, X2=None, presliced=False):
        if X2 is None:
            return self.variance.transform().expand(X.size(0)).diag()
        return torch.zeros(*_k_shape(X, X2), dtype=torch_dtype, device=X.device)


class Constant(Static):
    """kernels.py:95-101."""

    def K(self, X, X2=None, presliced=False):
        return self.variance.transform().expand(*_k_shape(X, X2))


class Bias(Constant):
    pass

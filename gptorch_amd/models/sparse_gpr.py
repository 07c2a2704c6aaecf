"""
Sparse GP regression: VFE (Titsias' collapsed bound), gptorch/models/sparse_gpr.py:22-195,
BASELINE config 5 / SURVEY 8(f)-1.  Forward (ELBO) and predict are composed from the
native primitives; everything N-sized lives in transposed storage so that every
contraction is the NT fp64-MFMA form:

    A^T = K(x, Z) L^-T            [N, M]   right-solve  (gpn_trsm_right_lt)
    A   = (A^T)^T                 [M, N]   HBM-bound transpose
    B   = A A^T / s2 + I          [M, M]   SYRK (lower) straight into a factor buffer
    c   = LB^-1 (A err) / s2               carried as the factor's extra rows

Status: evaluation and prediction only -- the hyper-parameter / inducing-point gradients
of the bound are not implemented natively yet, so the tensors returned here carry no
autograd graph (`optimize()` on a VFE model raises).  SVGP / FITC (sparse_gpr.py:76-90,
198-381) are out of scope.
"""
import math

import numpy as np
import torch

from .. import _ops
from ..mean_functions import Zero
from ..param import Param
from ..util import as_tensor
from .base import GPModel


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


class VFE(_InducingPointsGP):
    def __init__(self, *args, **kwargs):
        super().__init__(*args, **kwargs)
        assert isinstance(self.mean_function, Zero), "Mean functions not implemented for VFE yet."

    def _kind(self):
        from .. import kernels
        k = self.kernel
        if not isinstance(k, kernels.Stationary) or k._kind is None:
            raise NotImplementedError("gptorch_amd.VFE supports the native stationary kernels")
        return k

    def _common(self, x):
        """-> (f_uu, At [N, M] padded, fB (factor of B carrying c*s2), s2, trace(AAT))."""
        k = self._kind()
        with torch.no_grad():
            var, ls = k.variance.transform(), k.length_scales.transform()
            s2 = float(self.likelihood.variance.transform().item())
            Z, err = self.Z.detach(), self.Y          # sparse_gpr.py:125 quirk: err = self.Y
            n, dy = err.shape
            m = Z.shape[0]
            f_uu = _ops.kernel_factor(k._kind, Z, var, ls, None)                       # L = chol(K(Z)) (+ladder)
            At = _ops.padded_like_factor(f_uu, n)                                      # [N, M]
            _ops.kernel_matrix(k._kind, x, Z, var, ls, out=At, ldk=f_uu.ld)            # K(x, Z) = Kuf^T
            f_uu.solve_right_lt(At, n)                                                 # A^T = Kuf^T L^-T
            kp = _ops.round_up(n, 16)
            A = torch.zeros(_ops.round_up(m, 16), kp, dtype=torch.float64, device=x.device)
            _ops._native.check(_ops._native.lib().gpn_transpose(_ops._stream(x.device), _ops._ptr(At), n, m, At.stride(0),
                                                                 _ops._ptr(A), kp), "gpn_transpose")
            errT = torch.zeros(_ops.round_up(dy, 16), kp, dtype=torch.float64, device=x.device)
            errT[:dy, :n] = err.t()
            Aerr = _ops.gemm_nt(A, errT, m, dy, kp)                                    # A err  [M, dy]

            def attempt(jitter):
                fB = _ops.Factor(m, dy, x.device)
                _ops.gemm_nt(A, A, m, m, kp, alpha=1.0 / s2, C=fB.A, lower=True)       # AAT / s2 (lower)
                tr = fB.A.diagonal()[:m].sum()
                fB.A.diagonal()[:m].add_(1.0 if jitter is None else 1.0 + jitter)      # B = AAT + I
                fB.pack_rhs(Aerr)
                return fB, tr, fB.potrf()
            fB, tr, info = attempt(None)
            i = 0
            while info != 0:
                if i >= _ops.JITTER_TRIES:
                    raise RuntimeError("Max tries exceeded.")
                fB, tr, info = attempt(10.0 ** (-_ops.JITTER_TRIES + i))
                i += 1
        return f_uu, At, fB, s2, tr

    def log_likelihood(self, x=None, y=None):
        """variational lower bound, sparse_gpr.py:108-153 (0-dim tensor)."""
        x = x if x is not None else self.X
        y = y if y is not None else self.Y
        if not x.shape[0] == y.shape[0]:
            raise ValueError("X and Y must have same # data.")
        k = self._kind()
        f_uu, At, fB, s2, tr = self._common(x)
        n, d_out = self.Y.shape
        terms = fB.lml_terms()                       # [sum log LB_ii, || LB^-1 A err ||^2, ...]
        with torch.no_grad():
            elbo = -0.5 * d_out * n * math.log(2.0 * math.pi)
            elbo = elbo - d_out * terms[0]
            elbo = elbo - 0.5 * d_out * n * math.log(s2)
            elbo = elbo - 0.5 * (self.Y.pow(2).sum() + d_out * k.Kdiag(x).sum()) / s2
            elbo = elbo + 0.5 * terms[1] / (s2 * s2)          # c = LB^-1 (A err) / s2
            elbo = elbo + 0.5 * d_out * tr
        return elbo

    def _predict(self, x_new, diag=True, x=None):
        """sparse_gpr.py:155-195."""
        x = x if x is not None else self.X
        k = self._kind()
        f_uu, At, fB, s2, tr = self._common(x)
        with torch.no_grad():
            var, ls = k.variance.transform(), k.length_scales.transform()
            ns, m, dy = x_new.shape[0], self.Z.shape[0], self.Y.shape[1]
            T1 = _ops.padded_like_factor(f_uu, ns)                                    # tmp1^T = K(x*, Z) L^-T
            _ops.kernel_matrix(k._kind, x_new, self.Z.detach(), var, ls, out=T1, ldk=f_uu.ld)
            f_uu.solve_right_lt(T1, ns)
            T2 = T1.clone()
            fB.solve_right_lt(T2, ns)                                                 # tmp2^T = tmp1^T LB^-T
            kp = _ops.round_up(m, 16)
            mean = _ops.gemm_nt(T2, fB.A[m:], ns, dy, kp) / s2                        # tmp2^T c
            if diag:
                v = k.Kdiag(x_new).detach() - _ops.row_sumsq(T1, ns, m) + _ops.row_sumsq(T2, ns, m)
                return mean, v[:, None].expand_as(mean)
            cov = _ops.kernel_matrix(k._kind, x_new, None, var, ls)
            _ops.gemm_nt(T2, T2, ns, ns, kp, alpha=1.0, beta=1.0, C=cov)
            _ops.gemm_nt(T1, T1, ns, ns, kp, alpha=-1.0, beta=1.0, C=cov)
        return mean, cov

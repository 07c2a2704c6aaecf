"""
Exact GP regression (gptorch/models/gpr.py) over the native pipeline.

log_likelihood:  one autograd node = fused K(X)+sigma_n^2 I assembly into the
factor buffer -> blocked MFMA Cholesky with the residual (y - m)^T riding along
as extra rows (forward substitution for free) -> log-det / ||alpha||^2
reduction.  _predict re-uses the cached factor when neither the parameters nor
the inputs changed (the reference re-factorises on every call, gpr.py:104).
"""
import torch

from .. import _ops
from .. import kernels
from .base import GPModel


class GPR(GPModel):
    def __init__(self, x, y, kernel, mean_function=None, likelihood=None, name="gpr"):
        super().__init__(x, y, kernel, likelihood, mean_function, name)
        self._holder = {}        # reusable factor buffer for the training loop
        self._predict_cache = None

    def _stationary(self):
        k = self.kernel
        if not isinstance(k, kernels.Stationary) or k._kind is None:
            raise NotImplementedError(
                "gptorch_amd.GPR runs its fused native path for stationary kernels "
                "(Rbf, Matern52, Matern32, Exp); got %s" % type(k).__name__)
        return k

    def log_likelihood(self, x=None, y=None):
        """gpr.py:47-67; returns a tensor of shape (1,)."""
        x = x if x is not None else self.X
        y = y if y is not None else self.Y
        if not x.shape[0] == y.shape[0]:
            raise ValueError("X and Y must have same # data.")
        k = self._stationary()
        resid = y - self.mean_function(x)
        return _ops.GPRLogLik.apply(x, resid, k.variance.transform(), k.length_scales.transform(),
                                    self.likelihood.variance.transform(), k._kind, self._holder)

    def _compute_kyy(self, x=None):
        """K(x) + sigma_n^2 I as a dense tensor (gpr.py:69-86); API parity only --
        the training / predict paths never materialise it outside the factor buffer."""
        x = x if x is not None else self.X
        k = self._stationary()
        return _ops.kernel_matrix(k._kind, x, None, k.variance.transform(), k.length_scales.transform(),
                                  noise=self.likelihood.variance.transform())

    def _factor_for_predict(self, x):
        k = self._stationary()
        with torch.no_grad():
            var, ls, noise = k.variance.transform(), k.length_scales.transform(), self.likelihood.variance.transform()
            mean_x = self.mean_function(x)
            key = (x.data_ptr(), x._version, tuple(x.shape), self.Y.data_ptr(), self.Y._version,
                   var.cpu().numpy().tobytes(), ls.cpu().numpy().tobytes(), noise.cpu().numpy().tobytes(),
                   mean_x.sum().item(), k._kind)
            if self._predict_cache is None or self._predict_cache[0] != key:
                f = _ops.kernel_factor(k._kind, x, var, ls, noise, R=self.Y - mean_x)
                self._predict_cache = (key, f)
        return self._predict_cache[1], var, ls

    def _predict(self, x_new, diag=True, x=None):
        """p(F* | Y) (gpr.py:88-117): mean [n*, dy]; var [n*, dy] (diag) or cov [n*, n*]."""
        x = x if x is not None else self.X
        k = self._stationary()
        f, var, ls = self._factor_for_predict(x)
        with torch.no_grad():
            mean, v = _ops.gpr_predict(k._kind, x, x_new, var, ls, f, diag=diag)
            mean_f = mean + self.mean_function(x_new)
            var_f = v[:, None].expand_as(mean_f) if diag else v
        return mean_f, var_f

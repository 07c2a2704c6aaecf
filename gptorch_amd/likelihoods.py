"""Gaussian likelihood (gptorch/likelihoods.py:80-144), the only one on the GPR path."""
import math

import torch

from .model import Model
from .param import Param
from .settings import DefaultPositiveTransform
from .util import torch_dtype


class Likelihood(Model):
    def __init__(self):
        super().__init__()

    def forward(self):
        return None


class Gaussian(Likelihood):
    """(Spherical) Gaussian likelihood p(y|f) with variance Param (likelihoods.py:86-90)."""

    def __init__(self, variance=1.0):
        super().__init__()
        self.variance = Param(torch.tensor([float(variance)], dtype=torch_dtype),
                              transform=DefaultPositiveTransform())

    def logp(self, F, Y):
        """likelihoods.py:92-104."""
        return torch.distributions.Normal(F, torch.sqrt(self.variance.transform())).log_prob(Y)

    def predict_mean_variance(self, mean_f, var_f):
        """likelihoods.py:106-120."""
        return mean_f, var_f + self.variance.transform().expand_as(var_f)

    def predict_mean_covariance(self, mean_f, cov_f):
        """likelihoods.py:122-123."""
        return mean_f, cov_f + self.variance.transform().expand_as(cov_f).diag().diag()

    def propagate_log(self, qf, targets):
        """likelihoods.py:125-144."""
        if not isinstance(qf, (torch.distributions.Normal, torch.distributions.MultivariateNormal)):
            raise TypeError("Expect Gaussian q(f)")
        mu, s = qf.loc, qf.variance
        n = targets.nelement()
        if not mu.nelement() == n:
            raise ValueError("Targets (%i) and q(f) (%i) have mismatch in size" % (n, mu.nelement()))
        sigma_y = self.variance.transform()
        return -0.5 * (n * (math.log(2.0 * math.pi) + torch.log(sigma_y))
                       + (torch.sum((targets - mu) ** 2) + s.sum()) / sigma_y)

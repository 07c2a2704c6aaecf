"""
Param: an nn.Parameter that stores raw = transform.inv(value) and exposes the
constrained value through .transform(), plus an optional .prior -- behaviour of
gptorch/param.py:13-50 (gradients are therefore w.r.t. the raw, i.e. log, values).
"""
import torch
from torch.distributions.transforms import ComposeTransform


def _valid(transform):
    return ComposeTransform([]) if transform is None else transform


class Param(torch.nn.Parameter):
    def __new__(cls, data=None, requires_grad=True, transform=None, prior=None):
        raw = _valid(transform).inv(data)
        return super().__new__(cls, raw, requires_grad=requires_grad)

    def __init__(self, data, requires_grad=True, transform=None, prior=None):
        super().__init__()
        self._transform = _valid(transform)
        self.prior = prior

    def transform(self):
        return self._transform(self)

    def __repr__(self):
        return "Parameter containing:" + self.data.__repr__()

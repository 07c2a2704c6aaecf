"""Library-wide defaults.  Positive hyper-parameters (variances, length-scales) are optimised
through their logarithm, as in the reference (gptorch/settings.py:7)."""
import torch.distributions.transforms as _transforms


class DefaultPositiveTransform(_transforms.ExpTransform):
    """value = exp(raw): the constraint every positive Param is created with."""

"""Global defaults (gptorch/settings.py:7): positive parameters are stored as logs."""
from torch.distributions.transforms import ExpTransform

DefaultPositiveTransform = ExpTransform

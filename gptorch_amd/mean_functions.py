"""Constant / Zero mean functions (gptorch/mean_functions.py:15-49)."""
import torch

from .util import torch_dtype


class Constant(torch.nn.Module):
    def __init__(self, dy: int, val: torch.Tensor = None):
        super().__init__()
        if val is not None:
            if not val.shape[0] == dy:
                raise ValueError("Provided val doesn't match output dimension")
            val = val.clone()
        else:
            val = torch.zeros(dy, dtype=torch_dtype)
        self._dy = dy
        self.val = torch.nn.Parameter(val)

    def forward(self, x):
        return torch.zeros(x.shape[0], self._dy, dtype=torch_dtype, device=self.val.device) + self.val


class Zero(Constant):
    """Zero mean (default); `val` is frozen (mean_functions.py:42-49)."""

    def __init__(self, dy: int):
        super().__init__(dy)
        self.val.requires_grad_(False)

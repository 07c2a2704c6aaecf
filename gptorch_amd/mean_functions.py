"""Mean functions of the GP prior: one constant per output column (trainable) and the zero mean,
its frozen special case -- the behaviour of gptorch/mean_functions.py:15-49 (`val` [dy] parameter,
`ValueError` on a length mismatch, output [n, dy] on the parameter's device)."""
import torch

from .util import torch_dtype


def _initial_value(dy, val):
    if val is None:
        return torch.zeros(dy, dtype=torch_dtype)
    if val.shape[0] != dy:
        raise ValueError("Provided val doesn't match output dimension")
    return val.detach().clone()


class Constant(torch.nn.Module):
    """m(x)[i, :] = val for every row i of x."""

    trainable = True

    def __init__(self, dy: int, val: torch.Tensor = None):
        super().__init__()
        self._dy = int(dy)
        self.val = torch.nn.Parameter(_initial_value(self._dy, val), requires_grad=self.trainable)

    def forward(self, x):
        # broadcast view -> owned [n, dy] tensor; the backward of expand() sums over the rows,
        # which is the gradient the reference gets from `zeros + val`
        return self.val.unsqueeze(0).expand(x.shape[0], self._dy).clone()


class Zero(Constant):
    """The default mean: val = 0 and excluded from training (mean_functions.py:42-49)."""

    trainable = False

    def __init__(self, dy: int):
        super().__init__(dy)

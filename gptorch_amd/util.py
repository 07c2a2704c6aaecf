"""
fp64 type aliases and tensor conversion -- the shell counterpart of
gptorch/util.py:11-31 (TensorType, torch_dtype, as_tensor).
`squared_distance` (util.py:73-88) is fused into the native K-assembly kernel
and never materialised on the hot path; it is exposed here (forward only) for
API compatibility.
"""
import numpy as np
import torch

TensorType = torch.DoubleTensor
torch_dtype = torch.double


def as_tensor(x):
    """numpy / float / tensor -> fp64 tensor (util.py:15-31); keeps a tensor's device."""
    if isinstance(x, torch.Tensor):
        return x.to(torch_dtype)
    if isinstance(x, np.ndarray):
        return torch.as_tensor(x, dtype=torch_dtype).clone()
    elif isinstance(x, float):
        return torch.tensor([x], dtype=torch_dtype)
    else:
        raise TypeError("Unsupported type {}".format(type(x)))


def squared_distance(x1, x2=None):
    """[n1, n2] pairwise squared distances (util.py:73-88), computed by direct
    differences in the native kernel (so never negative).  Forward only."""
    from . import _ops
    one = torch.ones(1, dtype=torch_dtype, device=x1.device)
    return _ops.kernel_matrix("SqDist", x1, x2, one, one)

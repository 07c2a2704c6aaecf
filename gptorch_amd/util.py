"""
fp64 type aliases and tensor conversion -- the shell counterpart of
gptorch/util.py:11-31 (TensorType, torch_dtype, as_tensor).
`squared_distance` (util.py:73-88) is fused into the native K-assembly kernel
and never materialised on the hot path; as a public op it is the same kernel
with K = r^2 (kind SQDIST) behind an autograd node (kernels._SqDist).
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
    differences in the native kernel (so never negative); first and second derivatives w.r.t.
    the points as pinned by the reference's test/test_util.py:46-106."""
    from .kernels import _SqDist
    one = torch.ones(1, dtype=torch_dtype, device=x1.device)
    return _SqDist.apply(x1, x1 if x2 is None else x2, one)

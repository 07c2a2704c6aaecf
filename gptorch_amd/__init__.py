"""
gptorch_amd -- MI355X (gfx950) native exact-GP hot path behind the gptorch
v0.3.2 call surface (kernels.Rbf/Matern52 .K/.Kdiag, functions.cholesky/trtrs/
lt_log_determinant, models.GPR .loss/.log_likelihood/.predict_f/.predict_y/
.optimize).  All dense arithmetic runs in libgpnative.so (hand-written HIP);
PyTorch supplies device memory, streams and autograd plumbing only.
"""
__version__ = "0.1.0"

from . import util, settings, param, functions, kernels, likelihoods, mean_functions, model, models  # noqa: F401
from .param import Param  # noqa: F401

"""GP models (gptorch/models/__init__.py:20-21)."""
from .base import GPModel  # noqa: F401
from .gpr import (GPR, batched_factorise, batched_log_likelihood, batched_loss_and_grad, multi_start_optimize,  # noqa: F401
                  release_batch_buffers)
from .sparse_gpr import VFE  # noqa: F401
from .dist_gpr import DistGPR  # noqa: F401

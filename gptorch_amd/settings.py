"""Library-wide defaults.  Positive hyper-parameters (variances, length-scales) are optimised
through their logarithm, as in the reference (gptorch/settings.py:7)."""
import torch.distributions.transforms as _transforms


class DefaultPositiveTransform(_transforms.ExpTransform):
    """value = exp(raw): the constraint every positive Param is created with."""


# Opt-in automatic placement (round 5).  The reference builds and evaluates models on the CPU by default
# (gptorch/models/base.py:82-85, 392-416; its example runs without --cuda, examples/regression_1d.py:89-95); this package has
# no CPU arithmetic (a CPU tensor raises NativeError, and nothing ever falls back to the oracle).  With auto_device = True a
# CPU-constructed model is moved to the GPU ONCE -- data and parameters, exactly `model.cuda()` -- by its first loss() /
# optimize() / predict_*() call; predictions keep the reference's contract (numpy in -> numpy out, CPU tensor in -> CPU
# tensor out, base.py:21-55).  Default off: placement stays explicit unless the user asks.  GPTORCH_AMD_AUTO_DEVICE=1 turns
# it on for unmodified scripts.
import os as _os

auto_device = _os.environ.get("GPTORCH_AMD_AUTO_DEVICE", "0") not in ("", "0", "false", "False")

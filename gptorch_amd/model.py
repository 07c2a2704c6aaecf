"""
Model: nn.Module base of every GP object -- loss(), log_prior(), the flat
parameter vector glue for scipy optimisers (gptorch/model.py:33-217).
"""
import numpy as np
import torch

from .param import Param
from .util import torch_dtype


def _indent(text, n):
    lines = text.split("\n")
    if len(lines) == 1:
        return text
    return lines[0] + "\n" + "\n".join(" " * n + l for l in lines[1:])


class Model(torch.nn.Module):
    def forward(self):
        return None

    def __repr__(self):
        out = self.__class__.__name__ + " (\n"
        for name, p in self._parameters.items():
            shown = p.transform().data if hasattr(p, "transform") else p.data
            out += name + "\n" + str(shown) + "\n"
        for key, module in self._modules.items():
            out += "  (" + key + "): " + _indent(module.__repr__(), 2) + "\n"
        return out + ")\n"

    # ---- flat-parameter glue for scipy.optimize (model.py:56-133) -------------
    def _get_param_array(self):
        return np.concatenate([p.detach().cpu().numpy().flatten() for p in self.parameters() if p.requires_grad])

    def _set_parameters(self, param_array):
        at = 0
        for p in self.parameters():
            if p.requires_grad:
                nxt = at + p.numel()
                p.data = torch.as_tensor(np.reshape(param_array[at:nxt], p.shape), dtype=torch_dtype).to(p.device)
                at = nxt

    def _loss_and_grad(self, param_array):
        """f(x), g(x) for scipy; non-finite gradient entries -> 0 (model.py:123-133)."""
        self._set_parameters(param_array)
        for p in self.parameters():
            if p.grad is not None:
                p.grad.data.zero_()
        loss = self.loss()
        loss.backward()
        grad = np.concatenate([p.grad.cpu().numpy().flatten() for p in self.parameters() if p.requires_grad])
        print("loss: %s" % loss.item())
        finite = np.isfinite(grad)
        if np.all(finite):
            return float(loss.item()), grad.astype(np.float64)
        print("Warning: inf or nan in gradient: replacing with zeros")
        return loss.item(), np.where(finite, grad, 0.0).astype(np.float64)

    def extract_params(self):
        return tuple(self.parameters())

    def expand_params(self, *args):
        for arg, (_, p) in zip(args, self.named_parameters()):
            if isinstance(arg, Param):
                p.data = arg.data
            elif isinstance(arg, np.ndarray):
                raise NotImplementedError("Unresolved issues with expanding numpy arrays")

    def log_prior(self):
        """Sum of prior log-densities over parameters that carry one (model.py:158-177)."""
        total = 0.0
        for p in self.parameters():
            if getattr(p, "prior", None) is not None:
                val = p.transform() if hasattr(p, "transform") else p.data
                total = total + p.prior.log_prob(val).sum()
        return total

    def loss(self, *loss_args, params=None, **loss_kwargs):
        """model.py:179-197."""
        if params is not None:
            self.expand_params(*params)
        return self._loss(*loss_args, **loss_kwargs)

    compute_loss = loss  # pre-0.3 name used by BASELINE.json's north_star wording

    def gradcheck(self, eps=1e-6, atol=1e-5, rtol=1e-3, verbose=False):
        return torch.autograd.gradcheck(self.loss, self.extract_params(), eps=eps, atol=atol, rtol=rtol)

    def _loss(self, *args, **kwargs):
        raise NotImplementedError("Implement loss function")

"""
GPModel: data + kernel + likelihood + mean function, the optimiser loop and the
public predict API -- behaviour of gptorch/models/base.py (the shell around the
native hot path; no dense arithmetic happens here).
"""
import os
from time import time

import numpy as np
import torch
from scipy.optimize import minimize

from .. import likelihoods
from ..functions import cholesky
from ..mean_functions import Zero
from ..model import Model
from ..util import as_tensor, torch_dtype


def input_as_tensor(predict_func):
    """numpy in -> numpy out, tensor in -> tensor out on the caller's device
    (base.py:21-55)."""

    def predict(obj, input_new, *args, **kwargs):
        obj._auto_place()
        from_numpy = isinstance(input_new, np.ndarray)
        if from_numpy:
            input_new = torch.as_tensor(input_new, dtype=torch_dtype).to(obj.Y.device)
        else:
            outside_device = input_new.device
            input_new = input_new.to(obj.Y.device)
        out = predict_func(obj, input_new, *args, **kwargs)

        def back(o):
            return o.detach().cpu().numpy() if from_numpy else o.to(outside_device)

        if isinstance(out, torch.Tensor):
            return back(out)
        if isinstance(out, tuple):
            return tuple(back(o) for o in out)
        raise NotImplementedError("Unhandled output type {}".format(type(out)))

    return predict


_TORCH_DEFAULT_LR = {"SGD": 0.001, "Adam": 0.01, "LBFGS": 1.0, "Adadelta": 1.0, "Adagrad": 0.01,
                     "Adamax": 0.002, "ASGD": 0.01, "RMSprop": 0.01, "Rprop": 0.01}   # base.py:131-141
_SCIPY_METHODS = ["CG", "BFGS", "Newton-CG", "Nelder-Mead", "Powell", "L-BFGS-B", "TNC", "COBYLA", "SLSQP",
                  "dogleg", "trust-ncg"]                                            # base.py:203-215


FOREACH_ON_GPU = os.environ.get("GPTORCH_AMD_FOREACH", "1") != "0"     # _make_optimizer: multi-tensor optimiser kernels for GPU parameters


class GPModel(Model):
    def __init__(self, x, y, kernel, likelihood, mean_function, name="gp"):
        super().__init__()
        self.kernel = kernel
        self.likelihood = likelihood if likelihood is not None else GPModel._init_gaussian_likelihood(y)
        self.mean_function = mean_function if mean_function is not None else Zero(y.shape[1])
        x, y = as_tensor(x), as_tensor(y)
        x.requires_grad_(False)
        y.requires_grad_(False)
        self.X, self.Y = x, y
        self.__class__.__name__ = name

    def _auto_place(self):
        """settings.auto_device (opt-in): a CPU-resident model moves to the GPU once, as `model.cuda()` would
        (base.py:392-399), before its first native call.  Without a visible GPU nothing happens and the native call raises
        as always -- there is no CPU arithmetic to fall back to."""
        from .. import settings
        if settings.auto_device and not self.X.is_cuda and torch.cuda.is_available():
            self.cuda()

    @property
    def num_data(self):
        return self.Y.shape[0]

    @property
    def input_dimension(self):
        return self.X.shape[1]

    @property
    def output_dimension(self):
        return self.Y.shape[1]

    @staticmethod
    def _init_gaussian_likelihood(y) -> likelihoods.Gaussian:
        """0.001 * y.var() -- numpy (ddof=0) for arrays, torch (unbiased) for
        tensors, as base.py:101-109 behaves."""
        return likelihoods.Gaussian(variance=float(0.001 * y.var()))

    def _make_optimizer(self, method, parameters, learning_rate):
        """The nine torch optimisers with the reference's settings (base.py:144-200).  On the GPU the multi-tensor ("foreach")
        implementation is asked for explicitly: PyTorch picks it by itself for plain Parameters only, and would step `Param`
        tensors (a Parameter subclass, as in the reference) one small launch per tensor and operation."""
        lr = learning_rate
        o = torch.optim
        parameters = list(parameters)
        fe = {"foreach": True} if (FOREACH_ON_GPU and parameters and all(p.is_cuda for p in parameters)) else {}
        if method == "SGD":
            return o.SGD(parameters, lr=lr if lr is not None else 0.01, momentum=0.9, **fe)
        if method == "Adam":
            return o.Adam(parameters, lr=lr if lr is not None else 0.01, **fe)
        if method == "LBFGS":
            return o.LBFGS(parameters, lr=1.0 if lr is None else lr, max_iter=5, max_eval=None, tolerance_grad=1e-05,
                           tolerance_change=1e-09, history_size=50, line_search_fn=None)
        if method == "Adadelta":
            return o.Adadelta(parameters, lr=lr, rho=0.9, eps=1e-06, weight_decay=0.00001, **fe)
        if method == "Adagrad":
            return o.Adagrad(parameters, lr=lr, lr_decay=0, weight_decay=0, **fe)
        if method == "Adamax":
            return o.Adamax(parameters, lr=lr, betas=(0.9, 0.999), eps=1e-08, weight_decay=0, **fe)
        if method == "ASGD":
            return o.ASGD(parameters, lr=lr, lambd=0.0001, alpha=0.75, t0=1000000.0, weight_decay=0, **fe)
        if method == "RMSprop":
            return o.RMSprop(parameters, lr=lr, alpha=0.99, eps=1e-08, weight_decay=0.00, momentum=0.01, centered=False, **fe)
        if method == "Rprop":
            return o.Rprop(parameters, lr=lr, etas=(0.5, 1.2), step_sizes=(1e-06, 50), **fe)
        return None

    def optimize(self, method="Adam", max_iter=2000, verbose=True, learning_rate=None, capture=False):
        """Minimise loss() over the trainable parameters; returns (losses, seconds)
        for torch optimisers, the scipy result for scipy methods (base.py:111-296).
        capture=True (GPR over a native stationary kernel on the GPU, torch optimisers but LBFGS): the body of the reference's
        loop -- zero_grad(); loss(); backward(); step() (base.py:260-269) -- is captured into ONE hipGraph and replayed per
        iteration; the losses stay on the device and come back once at the end (the reference's `loss.item()` is a host
        synchronisation per iteration).  Same returned array, same printed lines (printed after the loop).  Anything else falls
        back to the ordinary loop."""
        self._auto_place()
        if capture and self._can_capture(method):
            return self._optimize_captured(method, max_iter, verbose, learning_rate)
        parameters = [p for p in self.parameters() if p.requires_grad]
        if learning_rate is None and method in _TORCH_DEFAULT_LR:
            learning_rate = _TORCH_DEFAULT_LR[method]
        if method in _SCIPY_METHODS:
            print("Scipy.optimize.minimize...")
            return self._optimize_scipy(method=method, maxiter=max_iter, disp=verbose)
        self.optimizer = self._make_optimizer(method, parameters, learning_rate)
        if self.optimizer is None:
            raise ValueError("Optimizer %s is not found. Supported: %s and scipy's %s"
                             % (method, ", ".join(_TORCH_DEFAULT_LR), ", ".join(_SCIPY_METHODS)))

        losses = np.zeros(max_iter)
        tic = time()
        print("{}: Start optimizing via {}".format(self.__class__.__name__, method))

        def closure():
            self.optimizer.zero_grad()
            loss = self.loss()
            loss.backward()
            return loss

        for idx in range(max_iter):
            if method == "LBFGS":
                loss = self.optimizer.step(closure)
                if isinstance(loss, float):  # converged
                    losses[idx] = loss
                    losses = losses[0: idx + 1]
                    break
                losses[idx] = loss.item()
            else:
                loss = closure()
                self.optimizer.step()
                losses[idx] = loss.item()
            if verbose or idx % 20 == 0:
                print("Iter: %d\tLoss: %s" % (idx, losses[idx]))
        t = time() - tic
        print("Optimization time taken: %s s" % t)
        print("Optimization method: %s" % str(self.optimizer))
        if len(losses) == max_iter:
            print("Optimization terminated by reaching the maximum iterations")
        else:
            print("Optimization terminated by getting below the tolerant error")
        return losses, t

    # ---- the optimiser step as one hipGraph replay (round 6) ---------------------------------------------------------------
    CAPTURE_WARMUP = 3          # eager iterations before the capture (torch's recipe: lazily created optimiser state, caches)
    CAPTURE_CHUNK = 25          # replays between two looks at the "a factorisation failed" flag

    def _can_capture(self, method):
        from .. import _ops
        stationary = getattr(self, "_stationary", None)
        if method not in _TORCH_DEFAULT_LR or method == "LBFGS" or not self.X.is_cuda or stationary is None or stationary() is None:
            return False
        if type(self).log_likelihood.__qualname__ != "GPR.log_likelihood":      # VFE / DistGPR / user subclasses: their own path
            return False
        return True

    def _optimize_captured(self, method, max_iter, verbose, learning_rate):
        """see optimize(capture=True).  Numerics: the optimiser is built with capturable=True and its step counters in fp64
        (torch creates them in the DEFAULT dtype, fp32, which would round the bias corrections to 1e-7): the trajectory
        agrees with the ordinary loop's to rounding (device pow instead of the host's for beta ** step), not bit for bit.
        A replay whose factorisation reports info != 0 cannot climb the jitter ladder inside the graph: the flag is read every
        CAPTURE_CHUNK replays, and a chunk that saw one is rolled back (parameters, optimiser state) and repeated eagerly."""
        import inspect
        from .. import _ops
        dev = self.X.device
        parameters = [p for p in self.parameters() if p.requires_grad]
        if learning_rate is None:
            learning_rate = _TORCH_DEFAULT_LR[method]
        prev_dtype = torch.get_default_dtype()
        torch.set_default_dtype(torch.float64)
        try:
            self.optimizer = self._make_optimizer(method, parameters, learning_rate)
            if "capturable" in inspect.signature(type(self.optimizer).__init__).parameters:
                for g in self.optimizer.param_groups:
                    g["capturable"] = True
            losses_dev = torch.zeros(max(1, max_iter), dtype=torch.float64, device=dev)
            counter = torch.zeros(1, dtype=torch.long, device=dev)
            tic = time()
            print("{}: Start optimizing via {}".format(self.__class__.__name__, method))

            def eager_step(idx):
                self.optimizer.zero_grad(set_to_none=True)
                loss = self.loss()
                loss.backward()
                self.optimizer.step()
                losses_dev[idx] = loss.detach().reshape(())

            done = 0
            side = torch.cuda.Stream(device=dev)
            side.wait_stream(torch.cuda.current_stream(dev))
            with torch.cuda.stream(side):
                while done < min(self.CAPTURE_WARMUP, max_iter):
                    eager_step(done)
                    done += 1
            torch.cuda.current_stream(dev).wait_stream(side)
            if done < max_iter:
                counter.fill_(done)
                self.optimizer.zero_grad(set_to_none=True)
                deferred = _ops.DeferredInfo(dev)
                graph = torch.cuda.CUDAGraph()
                with torch.cuda.graph(graph), deferred:
                    loss = self.loss()
                    loss.backward()
                    self.optimizer.step()
                    losses_dev.index_copy_(0, counter, loss.detach().reshape(1))
                    counter.add_(1)
                # (the capture itself executed nothing: iteration `done` is the first replay)

                def snapshot():
                    return ([p.detach().clone() for p in parameters],
                            [{k: (v.clone() if torch.is_tensor(v) else v) for k, v in self.optimizer.state[p].items()} for p in parameters])

                def restore(snap):
                    with torch.no_grad():
                        for p, v in zip(parameters, snap[0]):
                            p.copy_(v)
                        for p, st in zip(parameters, snap[1]):
                            for k, v in st.items():
                                if torch.is_tensor(v):
                                    self.optimizer.state[p][k].copy_(v)     # IN PLACE: the graph holds these addresses
                                else:
                                    self.optimizer.state[p][k] = v

                while done < max_iter:
                    chunk = min(self.CAPTURE_CHUNK, max_iter - done)
                    snap = snapshot()
                    deferred.flag.zero_()
                    for _ in range(chunk):
                        graph.replay()
                    if int(deferred.flag.item()) != 0:            # ONE read-back per chunk
                        restore(snap)
                        for k in range(chunk):
                            eager_step(done + k)
                        counter.fill_(done + chunk)
                    done += chunk
            losses = losses_dev[:max_iter].cpu().numpy().astype(np.float64)
        finally:
            torch.set_default_dtype(prev_dtype)
        for idx in range(max_iter):
            if verbose or idx % 20 == 0:
                print("Iter: %d\tLoss: %s" % (idx, losses[idx]))
        t = time() - tic
        print("Optimization time taken: %s s" % t)
        print("Optimization method: %s" % str(self.optimizer))
        print("Optimization terminated by reaching the maximum iterations")
        return losses, t

    def _optimize_scipy(self, method="L-BFGS-B", tol=None, callback=None, maxiter=1000, disp=True):
        """scipy.optimize.minimize on the flat raw-parameter vector (base.py:298-320)."""
        return minimize(fun=self._loss_and_grad, x0=self._get_param_array(), method=method, jac=True, tol=tol,
                        callback=callback, options=dict(disp=disp, maxiter=maxiter))

    def _predict(self, input_new, diag=True):
        raise NotImplementedError()

    @input_as_tensor
    def predict_f(self, input_new, diag=True, **kwargs):
        """mean and (diag) variance / full covariance of the latent function (base.py:338-346)."""
        return self._predict(input_new, diag=diag, **kwargs)

    @input_as_tensor
    def predict_y(self, input_new, diag=True, **kwargs):
        """... of the observations (base.py:348-360)."""
        mean_f, cov_f = self._predict(input_new, diag=diag, **kwargs)
        if diag:
            return self.likelihood.predict_mean_variance(mean_f, cov_f)
        return self.likelihood.predict_mean_covariance(mean_f, cov_f)

    def _samples(self, mu, sigma, n_samples, z=None):
        """mu + chol(sigma) z for z ~ N(0, I) of shape [n_samples, n_test, dy] (base.py:371-374 / 386-389).  The Cholesky is
        the native factorisation (functions.cholesky) and all n_samples * dy matrix-vector products go out as ONE native
        fp64 contraction  Z^T L^T  with the draws as rows (the reference issues a batched matmul).  z: the standard-normal
        draws, for callers (and tests) that bring their own."""
        chol_s = cholesky(sigma)
        if z is None:
            z = torch.randn(n_samples, *mu.shape, dtype=torch_dtype, device=chol_s.device)
        if not chol_s.is_cuda:
            return mu + chol_s[None, :, :] @ z
        from .. import _ops
        ns, nt, dy = z.shape
        zt = z.permute(0, 2, 1).reshape(ns * dy, nt).contiguous()            # one row per (sample, output) pair
        lz = _ops.matmul_nt(zt, chol_s.detach().contiguous())                 # [ns * dy, nt] = Z^T L^T
        return mu + lz.reshape(ns, dy, nt).permute(0, 2, 1)

    @input_as_tensor
    def predict_f_samples(self, input_new, n_samples=1, **kwargs):
        """[n_samp, n_test, dy] draws (base.py:362-375)."""
        mu, sigma = self.predict_f(input_new, diag=False, **kwargs)
        return self._samples(mu, sigma, n_samples)

    @input_as_tensor
    def predict_y_samples(self, input_new, n_samples=1, **kwargs):
        """base.py:377-390."""
        mu, sigma = self.predict_y(input_new, diag=False, **kwargs)
        return self._samples(mu, sigma, n_samples)

    def cuda(self):
        """Moves parameters AND data; returns None like base.py:392-399."""
        super().cuda()
        self.X, self.Y = self.X.cuda(), self.Y.cuda()

    def cpu(self):
        super().cpu()
        self.X, self.Y = self.X.cpu(), self.Y.cpu()

    def _loss(self, *args, **kwargs):
        self._auto_place()
        return -(self.log_likelihood(*args, **kwargs) + self.log_prior())

"""
Exact GP regression for N beyond one GPU's HBM with the call surface of gptorch.models.GPR
(gptorch/models/gpr.py:26-117): `loss()` / `log_likelihood()` with autograd, `optimize()`,
`predict_f` / `predict_y` -- over the 2-D block-cyclic engine of gptorch_amd/dist.py, one process per
GPU (`torch.distributed`, backend "nccl" = RCCL).  The reference has no counterpart
(models/base.py:33 "Assume single GPU"); every rank constructs the same model on the same data and
calls every method collectively, exactly as it would call GPR's on one GPU:

    dist.init_process_group("nccl", device_id=device)
    model = DistGPR(x, y, kernels.Rbf(d)); model.cuda()
    model.optimize(method="Adam", max_iter=50)          # identical parameter trajectories on all ranks
    mu, var = model.predict_y(x_test)

log_likelihood is one autograd node: forward = distributed assembly + factorisation (the residual
riding along), backward = the closed form on the same grid (U = L^-T carried as identity rows,
Kyy^-1 = U U^T with panels of U travelling like factorisation panels, per-rank sweeps, D + 2 scalars
all-reduced).  Prediction sends the test points through one more factorisation as extra residual
rows.  Kernels: the native stationary kinds, and (round 6) Sum / Product trees over native leaves through the fused
expression kernels (gptorch_amd/_expr.py: gpn_kernel_matrix_expr per tile block, gpn_kernel_expr_grad per leaf and tile
block, gpn_refine_resid_part_expr) -- the reference's example model Linear + Rbf + Constant
(examples/regression_1d.py:34-53) runs on the grid.  Trees with a White leaf are not taken (a rectangular tile block has
no diagonal to put it on).
"""
import torch

from .. import kernels
from .gpr import GPR


class _DistLogLik(torch.autograd.Function):
    @staticmethod
    def forward(ctx, variance, length_scales, noise, resid, engine):
        need = any(ctx.needs_input_grad[:4])
        if need:
            lml, g = engine.log_likelihood_and_grad(variance.detach(), length_scales.detach(), noise.detach(), resid.detach())
            nls = length_scales.numel()
            ctx.g_var, ctx.g_ls, ctx.g_noise = g[0:1].clone(), g[1:1 + nls].clone(), g[1 + nls:2 + nls].clone()
            ctx.g_resid = -engine.last_a.t().contiguous()          # dLML/d(y - m) = -a
            ctx.ls_shape = length_scales.shape
        else:
            lml = engine.log_likelihood(variance.detach(), length_scales.detach(), noise.detach(), resid.detach())
        return lml.reshape(1)

    @staticmethod
    def backward(ctx, grad_out):
        go = grad_out.reshape(())
        return (go * ctx.g_var, go * ctx.g_ls.reshape(ctx.ls_shape), go * ctx.g_noise,
                go * ctx.g_resid if ctx.needs_input_grad[3] else None, None)


class _DistExprLogLik(torch.autograd.Function):
    """_DistLogLik for a covariance expression: params = the leaves' constrained parameter tensors in the program's packing
    order; the engine sees them as ONE packed vector (its `variance` argument)."""

    @staticmethod
    def forward(ctx, noise, resid, engine, prog, *params):
        theta = prog.theta(params)
        empty = theta[:0]
        need = any(ctx.needs_input_grad[:2]) or any(ctx.needs_input_grad[4:])
        if need:
            lml, g = engine.log_likelihood_and_grad(theta, empty, noise.detach(), resid.detach())
            ctx.g_theta, ctx.g_noise = g[:prog.ntheta].clone(), g[prog.ntheta:prog.ntheta + 1].clone()
            ctx.g_resid = -engine.last_a.t().contiguous()
            ctx.shapes = [p.shape for p in params]
        else:
            lml = engine.log_likelihood(theta, empty, noise.detach(), resid.detach())
        return lml.reshape(1)

    @staticmethod
    def backward(ctx, grad_out):
        go = grad_out.reshape(())
        outs, off = [], 0
        for shp in ctx.shapes:
            cnt = int(torch.Size(shp).numel())
            outs.append(go * ctx.g_theta[off:off + cnt].reshape(shp))
            off += cnt
        return (go * ctx.g_noise, go * ctx.g_resid if ctx.needs_input_grad[1] else None, None, None) + tuple(outs)


class DistGPR(GPR):
    def __init__(self, x, y, kernel, mean_function=None, likelihood=None, name="dist_gpr", tile=2048, grid=None, tile_ops=None, schedule=None):
        super().__init__(x, y, kernel, mean_function=mean_function, likelihood=likelihood, name=name)
        self._prog = None
        if not (isinstance(kernel, kernels.Stationary) and kernel._kind is not None):
            from .. import _native
            prog = kernel.fused_program() if isinstance(kernel, kernels.Combination) else None
            if prog is None or any(t.type == _native.TERM_WHITE for t in prog.terms):
                raise NotImplementedError("DistGPR assembles its tiles with the native stationary kernels (Rbf, Matern52, ...) or with "
                                          "Sum / Product trees over native leaves (no White leaf)")
            self._prog = prog
        self._tile, self._grid, self._tile_ops = int(tile), grid, tile_ops
        self._schedule = schedule        # panel exchange: "bcast" | "mesh" (dist.BlockCyclicGP); None = GPN_DIST_SCHEDULE / "bcast"
        self._engine = None
        self._other = None               # (x, y, engine) for data passed to log_likelihood / _predict explicitly

    def _eng_for(self, x, y):
        """the engine for data other than the model's own (gpr.py:47-57 / 88-100 accept x and y): a second block-cyclic
        layout on the SAME grid, sharing the first engine's row / column sub-communicators; kept while the same tensors
        are passed again.  Collective like everything else: every rank passes the same data."""
        from .. import dist as gdist
        if x.shape[0] != y.shape[0]:
            raise ValueError("X and Y must have same # data.")
        base = self._eng()
        o = self._other
        if o is None or o[0] is not x or o[1] is not y or o[2].X.device != x.device:
            e = gdist.BlockCyclicGP(x, y, self._kspec(), tile=self._tile, grid=self._grid, ops=self._tile_ops,
                                    schedule=self._schedule, share=base)
            self._other = o = (x, y, e)
        return o[2]

    def _kspec(self):
        """what the engine's tile operations evaluate: the native kind's name, or the expression program"""
        return self._prog if self._prog is not None else self.kernel._kind

    def _eng(self):
        from .. import dist as gdist
        e = self._engine
        if e is None or e.X.device != self.X.device:
            e = gdist.BlockCyclicGP(self.X, self.Y, self._kspec(), tile=self._tile, grid=self._grid, ops=self._tile_ops,
                                    schedule=self._schedule)
            self._engine = e
        return e

    def log_likelihood(self, x=None, y=None):
        """gpr.py:47-67 on the grid; shape (1,).  x / y other than the training data (gpr.py:47-57 accepts them) get a
        block-cyclic layout of their own on the same grid (_eng_for)."""
        own = x is None and y is None
        x = x if x is not None else self.X
        y = y if y is not None else self.Y
        if not x.shape[0] == y.shape[0]:
            raise ValueError("X and Y must have same # data.")
        k = self.kernel
        resid = y - self.mean_function(x)
        if self._prog is not None:
            return _DistExprLogLik.apply(self.likelihood.variance.transform(), resid, self._eng() if own else self._eng_for(x, y),
                                         self._prog, *self._prog.params())
        return _DistLogLik.apply(k.variance.transform(), k.length_scales.transform(), self.likelihood.variance.transform(),
                                 resid, self._eng() if own else self._eng_for(x, y))

    def _predict(self, x_new, diag=True, x=None):
        """gpr.py:88-117 on the grid: mean [n*, dy]; var [n*, dy] (diag) or cov [n*, n*]."""
        k = self.kernel
        eng = self._eng() if x is None else self._eng_for(x, self.Y)
        x = x if x is not None else self.X
        with torch.no_grad():
            resid = self.Y - self.mean_function(x)
            if self._prog is not None:
                theta = self._prog.theta(self._prog.params())
                mean, v = eng.predict(theta, theta[:0], self.likelihood.variance.transform(), resid, x_new, diag=diag)
            else:
                mean, v = eng.predict(k.variance.transform(), k.length_scales.transform(), self.likelihood.variance.transform(),
                                      resid, x_new, diag=diag)
            mean_f = mean + self.mean_function(x_new)
            var_f = v[:, None].expand_as(mean_f) if diag else v
        return mean_f, var_f

"""
Composite covariance functions on the native path (csrc/kexpr.hip).

gptorch composes kernels with `+` and `*` (kernels.py:286-306 Sum / Product); its own example model is
`Linear + Rbf + Constant` (examples/regression_1d.py:34-53).  The reference evaluates such a tree with
elementwise torch ops on dense N x M matrices.  Here a tree whose leaves are native terms -- a stationary
kernel with a native kind (kernels.py:108-235), Linear (238-265), Constant / Bias (95-105), White (83-92) --
is expanded into a SUM OF PRODUCTS of leaf instances and handed to gpn_kernel_matrix_expr, which writes the
matrix once (for GPR: straight into the factor buffer); gradients come from gpn_kernel_expr_grad, one sweep per
leaf instance, re-computing everything from the points.

  Program          the expanded expression + the packing of all leaf parameters into one device vector
  kernel_matrix    K(X, X2) of a Program (forward only)
  ExprK            Kernel.K of a composite as ONE autograd node (parameter gradients; not w.r.t. the points)
  ExprLogLik       GPR.log_likelihood (gpr.py:47-67) over a composite kernel as one autograd node: fused
                   assembly -> native factorisation -> closed-form backward with the expression sweeps
"""
import ctypes

import torch

from . import _native, _ops
from ._ops import _c, _ptr, _stream


class Program:
    """sum-of-products form of a kernel tree.  `leaves`: the distinct leaf kernel objects in first-visit order;
    `groups`: list of lists of leaf indices (one list per product).  Parameters are packed leaf by leaf:
    [variance(s), length-scale(s)] in their CONSTRAINED values."""

    def __init__(self, leaves, groups):
        self.leaves, self.groups = leaves, groups
        if sum(len(g) for g in groups) > _native.EXPR_MAX_TERMS or len(groups) > _native.EXPR_MAX_GROUPS:
            raise ValueError("expression too large for the fused evaluation")
        self.offsets = []            # per leaf: (var_off, nvar, ls_off, nls)
        off = 0
        for k in leaves:
            nvar = k.variance.numel()
            nls = k.length_scales.numel() if hasattr(k, "length_scales") else 0
            self.offsets.append((off, nvar, off + nvar, nls))
            off += nvar + nls
        self.ntheta = off
        inst = [li for g in groups for li in g]                     # term instances, group by group
        self.instances = inst
        self.terms = (_native.ExprTerm * len(inst))()
        for i, li in enumerate(inst):
            k = leaves[li]
            var_off, nvar, ls_off, nls = self.offsets[li]
            self.terms[i] = _native.ExprTerm(_leaf_type(k), _ops.KINDS.get(getattr(k, "_kind", None) or "Rbf", 0), var_off, ls_off,
                                             max(nls, 1), max(nvar, 1))
        starts, pos = [0], 0
        for g in groups:
            pos += len(g)
            starts.append(pos)
        self.gstart = (ctypes.c_int * len(starts))(*starts)
        self.ngroups = len(groups)

    def signature(self):
        """hashable structure of the expression (leaf types / kinds / parameter counts per instance, grouping): two kernels
        with equal signatures run through the same kernels with the same launch shapes (lock-step groups)."""
        return (tuple((t.type, t.kind, t.var_off, t.ls_off, t.nls, t.nvar) for t in self.terms), tuple(self.gstart), self.ntheta)

    def params(self):
        """the leaves' parameter tensors in packing order (constrained values, autograd-connected)."""
        out = []
        for k in self.leaves:
            out.append(k.variance.transform())
            if hasattr(k, "length_scales"):
                out.append(k.length_scales.transform())
        return out

    def theta(self, params):
        return torch.cat([p.detach().reshape(-1) for p in params]).contiguous()

    def grad_supported(self, d):
        """per-dimension parameters keep one accumulator per input in registers: d <= 16 (kexpr.hip)."""
        return all(d <= 16 or (nvar == 1 and nls <= 1) for (_, nvar, _, nls) in self.offsets)

    def scatter(self, per_instance, params):
        """per-instance gradient vectors -> one gradient per parameter tensor (instances of one leaf add up)."""
        flat = torch.zeros(self.ntheta, dtype=torch.float64, device=params[0].device)
        for i, li in enumerate(self.instances):
            var_off, nvar, ls_off, nls = self.offsets[li]
            g = per_instance[i]
            flat[var_off:var_off + nvar] += g[:nvar]
            if nls:
                flat[ls_off:ls_off + nls] += g[nvar:nvar + nls]
        out, off = [], 0
        for p in params:
            out.append(flat[off:off + p.numel()].reshape(p.shape))
            off += p.numel()
        return out


def flat_grad(prog, per_instance):
    """per-instance gradient vectors (_sweeps) -> ONE flat vector [ntheta] in packing order (instances of one leaf add up)."""
    flat = torch.zeros(prog.ntheta, dtype=torch.float64, device=per_instance[0].device)
    for i, li in enumerate(prog.instances):
        var_off, nvar, ls_off, nls = prog.offsets[li]
        g = per_instance[i]
        flat[var_off:var_off + nvar] += g[:nvar]
        if nls:
            flat[ls_off:ls_off + nls] += g[nvar:nvar + nls]
    return flat


def _leaf_type(k):
    from . import kernels
    if isinstance(k, kernels.Stationary):
        return _native.TERM_STATIONARY
    if isinstance(k, kernels.Linear):
        return _native.TERM_LINEAR
    if isinstance(k, kernels.White):
        return _native.TERM_WHITE
    return _native.TERM_CONSTANT


def build(kernel):
    """-> Program for a tree of Sum / Product over native leaves, or None (some leaf has no native term, the
    expansion is too large, or the kernel is a single stationary leaf, which has its own fused path)."""
    from . import kernels

    def expand(k):                      # -> list of products, each a list of leaf objects
        if isinstance(k, kernels.Sum):
            a, b = expand(k.kern1), expand(k.kern2)
            return None if a is None or b is None else a + b
        if isinstance(k, kernels.Product):
            a, b = expand(k.kern1), expand(k.kern2)
            if a is None or b is None or len(a) * len(b) > _native.EXPR_MAX_GROUPS:
                return None
            return [x + y for x in a for y in b]
        native_stationary = isinstance(k, kernels.Stationary) and type(k).K is kernels.Stationary.K and k._kind in _ops.KINDS \
            and k._kind != "SqDist"
        simple = type(k) in (kernels.Linear, kernels.White, kernels.Constant, kernels.Bias)
        return [[k]] if (native_stationary or simple) else None

    sop = expand(kernel)
    if sop is None:
        return None
    leaves = []
    for prod in sop:
        for k in prod:
            if not any(k is q for q in leaves):
                leaves.append(k)
    groups = [[next(i for i, q in enumerate(leaves) if q is k) for k in prod] for prod in sop]
    try:
        return Program(leaves, groups)
    except ValueError:
        return None


def kernel_matrix(prog, theta, X, X2=None, noise=None, out=None, ldk=None, lower=False):
    """K(X, X2) of the expression as a new [n, m] tensor, or written into `out` (leading dimension ldk; lower: only the
    tiles on / below the diagonal; noise: added on the diagonal of K(X))."""
    _ops._req(X, X2, theta, noise)
    X = _c(X.detach())
    n, d = X.shape
    if X2 is not None:
        X2 = _c(X2.detach())
        m = X2.shape[0]
        if X2.shape[1] != d:
            raise ValueError("X and X2 must have the same input dimension")
    else:
        m = n
    if out is None:
        out = torch.empty(n, m, dtype=torch.float64, device=X.device)
        ldk = m
    noise = None if noise is None else _c(noise.detach())
    st = _native.lib().gpn_kernel_matrix_expr(_stream(X.device), prog.terms, len(prog.instances), prog.gstart, prog.ngroups, _ptr(theta),
                                              _ptr(X), n, _ptr(X2), m, d, _ptr(noise), _ops.GPN_LOWER if lower else _ops.GPN_FULL,
                                              _ptr(out), ldk)
    _native.check(st, "gpn_kernel_matrix_expr")
    return out


def _sweeps(prog, theta, X, X2, G, ldg, at=None, ldat=0, dy=0):
    """one gpn_kernel_expr_grad per leaf instance -> (list of per-instance gradient vectors, trace(W) or None)."""
    lib = _native.lib()
    X = _c(X.detach())
    n, d = X.shape
    X2c = None if X2 is None else _c(X2.detach())
    m = n if X2c is None else X2c.shape[0]
    lml = at is not None
    work = torch.empty(max(1, int(lib.gpn_kernel_expr_grad_work_bytes(n, m, d, 1 if lml else 0)) // 8), dtype=torch.float64, device=X.device)
    outs, trace = [], None
    for i, li in enumerate(prog.instances):
        _, nvar, _, nls = prog.offsets[li]
        want_trace = 1 if (lml and i == 0) else 0
        out = torch.empty(nvar + nls + want_trace, dtype=torch.float64, device=X.device)
        st = lib.gpn_kernel_expr_grad(_stream(X.device), prog.terms, len(prog.instances), prog.gstart, prog.ngroups, _ptr(theta), i,
                                      _ptr(X), n, _ptr(X2c), m, d, _ptr(G), ldg, _ptr(at), ldat, dy, want_trace, _ptr(work), _ptr(out))
        _native.check(st, "gpn_kernel_expr_grad")
        if want_trace:
            trace = out[nvar + nls:nvar + nls + 1]
        outs.append(out)
    return outs, trace


class ExprK(torch.autograd.Function):
    """Kernel.K(X, X2) of a composite kernel: one fused assembly; backward = one sweep per leaf instance against the
    incoming dense gradient.  Gradients w.r.t. the parameters only (callers that differentiate w.r.t. the points use the
    composed path)."""

    @staticmethod
    def forward(ctx, X, X2, prog, *params):
        theta = prog.theta(params)
        ctx.prog, ctx.has_x2 = prog, X2 is not None
        ctx.save_for_backward(X, X2 if X2 is not None else X, theta, *params)
        return kernel_matrix(prog, theta, X, X2)

    @staticmethod
    def backward(ctx, gK):
        from . import _backward
        X, X2, theta, *params = ctx.saved_tensors
        g = _backward._rowmajor(gK)
        outs, _ = _sweeps(ctx.prog, theta, X, X2 if ctx.has_x2 else None, g, g.stride(0))
        return (None, None, None) + tuple(ctx.prog.scatter(outs, params))


class ExprLogLik(torch.autograd.Function):
    """GPR.log_likelihood (gpr.py:47-67) for a composite kernel as ONE autograd node: the expression is assembled straight
    into the factor buffer (one N x N write, noise on the diagonal), the factorisation carries the residual as extra
    rows, and the backward is the closed form  dLML/dtheta = sum G o dKyy/dtheta  with  G = 1/2 (a a^T - dy Kyy^-1)
    formed on the fly inside the expression sweeps -- no dense N x N tensor passes through autograd."""

    @staticmethod
    def forward(ctx, X, R, noise, prog, holder, *params):
        theta = prog.theta(params)
        n, e = R.shape
        f = holder.get("factor")
        if f is None or f.n != n or f.e != e or f.device != X.device:
            f = _ops.Factor(n, e, X.device)
        holder["factor"] = f
        terms = _factor_single(prog, theta, X, R, _c(noise.detach()), f)
        ctx.prog, ctx.factor, ctx.generation = prog, f, f.generation
        ctx.save_for_backward(X, R, noise, theta, *params)
        return terms[2:3].clone()

    @staticmethod
    def backward(ctx, grad_out):
        from . import _backward
        X, R, noise, theta, *params = ctx.saved_tensors
        f, prog = ctx.factor, ctx.prog
        if f.generation != ctx.generation:           # the reusable buffer was refactorised since: rebuild privately
            f = _ops.Factor(R.shape[0], R.shape[1], X.device)

            def attempt(jitter):
                nz = _c(noise.detach())
                kernel_matrix(prog, theta, X, None, noise=nz if jitter is None else nz + jitter, out=f.A, ldk=f.ld, lower=True)
                f.pack_rhs(R)
                return f.potrf()
            _ops._ladder(attempt)
        n, dy = f.n, f.e
        U = _backward._upper_inverse(f)
        Kinv = _backward._kinv_lower(f, U)
        at = _ops.gemm_nt(f.A[n:], U, dy, n, _ops.round_up(n, 16), tri=_ops.TRI_B_UPPER)       # a^T = alpha^T U^T  [dy, n]
        outs, trace = _sweeps(prog, theta, X, None, Kinv, Kinv.stride(0), at=at, ldat=at.stride(0), dy=dy)
        go = grad_out.reshape(())
        g_params = [go * g for g in prog.scatter(outs, params)]
        return (None, -go * at.t() if ctx.needs_input_grad[1] else None, go * trace if ctx.needs_input_grad[2] else None,
                None, None) + tuple(g_params)


def _factor_single(prog, theta, X, R, nz0, f):
    """assembly -> factorisation (+ jitter ladder) -> terms (+ refinement from refine_min_n(expression=True) rows on) of ONE model
    into the factor buffer f: the body of ExprLogLik.forward, shared with the replay of a model that failed inside a lock-step batch."""
    n, e = R.shape

    def attempt(jitter):
        kernel_matrix(prog, theta, X, None, noise=nz0 if jitter is None else nz0 + jitter, out=f.A, ldk=f.ld, lower=True)
        f.pack_rhs(R)
        return f.potrf()

    f.jitter_rung = _ops._ladder(attempt)
    terms = f.lml_terms()
    f.refined = False
    if n >= _ops.refine_min_n(expression=True):
        lib = _native.lib()
        nz = nz0 if f.jitter_rung < 0 else nz0 + 10.0 ** (-_ops.JITTER_TRIES + f.jitter_rung)
        if f._refine_work is None:
            f._refine_work = torch.empty(max(1, int(lib.gpn_lml_refine_work_bytes(n, e)) // 8), dtype=torch.float64, device=X.device)
        Xc, Rc = _c(X.detach()), _c(R.detach())
        st = lib.gpn_lml_refine_expr(_stream(X.device), prog.terms, len(prog.instances), prog.gstart, prog.ngroups, _ptr(theta),
                                     _ptr(Xc), n, Xc.shape[1], _ptr(Rc), None, e, _ptr(nz), _ptr(f.A), f.ld, _ptr(f.winv),
                                     _ptr(f._refine_work), _ptr(terms))
        _native.check(st, "gpn_lml_refine_expr")
        f.refined = True
    return terms


class BatchedExprLogLik(torch.autograd.Function):
    """ExprLogLik for `batch` independent models with composite kernels of ONE structure (equal Program.signature()) in LOCK
    STEP -- the reference's own example model, Linear + Rbf + Constant (examples/regression_1d.py:34-53), in a multi-start
    search.  The assembly and the gradient sweeps are the expression's, model by model (their parameters differ); everything
    kernel-independent goes out once over all models: the factorisation (gpn_potrf_lower_batched), the reductions
    (gpn_lml_reduce_batched) and the backward's U = L^-T, Kyy^-1 = U U^T, a^T = alpha^T U^T (gpn_lml_kinv_batched).  A model whose
    factorisation reports info != 0 is replayed alone through the jitter ladder into a private factor.  Per model, values and
    gradients are bit-identical to ExprLogLik.
    Inputs: X [n, d] shared or [batch, n, d]; R [n, dy] shared or [batch, n, dy]; noise [batch]; progs: one Program per model;
    params: the models' constrained leaf parameters, model after model (len(progs[0].params()) each)."""

    @staticmethod
    def forward(ctx, X, R, noise, progs, holder, *params):
        batch = len(progs)
        npar = len(params) // batch
        n, e = R.shape[-2], R.shape[-1]
        lib = _native.lib()
        fb = holder.get("fb")
        if fb is None or fb.batch != batch or fb.n != n or fb.e != e or fb.A.device != X.device:
            fb = _ops.FactorBatch(batch, n, e, X.device)
        holder["fb"] = fb
        fb.generation += 1
        thetas = [progs[b].theta(params[b * npar:(b + 1) * npar]) for b in range(batch)]
        nz = _c(noise.detach().reshape(batch))
        Xb = lambda b: X if X.dim() == 2 else X[b]
        Rb = lambda b: R if R.dim() == 2 else R[b]
        factors = [fb.factor(b) for b in range(batch)]
        # ONE assembly launch over the models (their programs have one structure: the term table of the first, every model its own
        # row of parameter values), the right-hand sides as one copy (Factor.pack_rhs: (y - m)^T below the matrix, its corner cleared)
        theta_all = torch.stack(thetas)
        thetas = [theta_all[b] for b in range(batch)]
        Xc, d = _c(X.detach()), X.shape[-1]
        st = lib.gpn_kernel_matrix_expr_batched(_stream(X.device), progs[0].terms, len(progs[0].instances), progs[0].gstart, progs[0].ngroups, batch,
                                                _ptr(theta_all), theta_all.shape[1], _ptr(Xc), 0 if X.dim() == 2 else n * d, n, d, _ptr(nz),
                                                _ptr(fb.A), fb.ld, fb.sA)
        _native.check(st, "gpn_kernel_matrix_expr_batched")
        if e:
            A3 = fb.A.view(batch, fb.rows, fb.ld)
            A3[:, n:n + e, n:] = 0.0
            A3[:, n:n + e, :n] = R.detach().transpose(-1, -2)
        fb.info.zero_()
        st = lib.gpn_potrf_lower_batched(_stream(X.device), _ptr(fb.A), n, e, fb.ld, fb.sA, _ptr(fb.winv), fb.sW, _ptr(fb.info), batch)
        _native.check(st, "gpn_potrf_lower_batched")
        st = lib.gpn_lml_reduce_batched(_stream(X.device), _ptr(fb.A), n, e, fb.ld, fb.sA, _ptr(fb.out), batch)
        _native.check(st, "gpn_lml_reduce_batched")
        if n >= _ops.refine_min_n(expression=True):
            if getattr(fb, "_refine_work", None) is None:
                fb._refine_work = torch.empty(max(1, int(lib.gpn_lml_refine_work_bytes(n, e)) // 8), dtype=torch.float64, device=X.device)
            for b in range(batch):
                Xc, Rc = _c(Xb(b).detach()), _c(Rb(b).detach())
                st = lib.gpn_lml_refine_expr(_stream(X.device), progs[b].terms, len(progs[b].instances), progs[b].gstart, progs[b].ngroups,
                                             _ptr(thetas[b]), _ptr(Xc), n, Xc.shape[1], _ptr(Rc), None, e, _ptr(nz[b:b + 1]), _ptr(factors[b].A),
                                             fb.ld, _ptr(factors[b].winv), _ptr(fb._refine_work), _ptr(fb.out[b]))
                _native.check(st, "gpn_lml_refine_expr")
        out = fb.out[:, 2].clone()
        info = fb.info.cpu()                         # one read-back for the batch
        replayed = {}
        for b in (torch.nonzero(info).reshape(-1).tolist() if bool(info.any()) else ()):
            if int(info[b]) < 0:
                raise _ops.NativeError("factorisation reported the internal status %d (not a property of the matrix)" % int(info[b]))
            f = _ops.Factor(n, e, X.device)
            out[b] = _factor_single(progs[b], thetas[b], Xb(b), Rb(b), nz[b:b + 1], f)[2]
            replayed[b] = f
        ctx.progs, ctx.fb, ctx.generation, ctx.replayed, ctx.npar = progs, fb, fb.generation, replayed, npar
        ctx.save_for_backward(X, R, noise, *thetas, *params)
        return out

    @staticmethod
    def backward(ctx, grad_out):
        from . import _backward
        X, R, noise, *rest = ctx.saved_tensors
        progs, npar = ctx.progs, ctx.npar
        batch = len(progs)
        thetas, params = rest[:batch], rest[batch:]
        n, dy = R.shape[-2], R.shape[-1]
        Xb = lambda b: X if X.dim() == 2 else X[b]
        Rb = lambda b: R if R.dim() == 2 else R[b]
        lib = _native.lib()
        fb = ctx.fb
        if fb.generation != ctx.generation:
            # the shared buffers were refactorised by a later forward: rebuild this node's factors privately (one by one)
            fb = None
        lay = (ctypes.c_int64 * 4)()
        _native.check(lib.gpn_lml_kinv_layout(n, dy, lay), "gpn_lml_kinv_layout")
        ld, koff, aoff, stride = (int(v) for v in lay)
        kinvs, ats = {}, {}
        if fb is not None:
            need = max(1, int(lib.gpn_lml_kinv_batched_work_bytes(n, dy, batch)) // 8)
            if fb._backward_work is None or fb._backward_work.numel() < need:
                fb._backward_work = torch.empty(need, dtype=torch.float64, device=X.device)
            st = lib.gpn_lml_kinv_batched(_stream(X.device), batch, n, _ptr(fb.A), fb.ld, fb.sA, _ptr(fb.winv), fb.sW, dy, _ptr(fb._backward_work))
            _native.check(st, "gpn_lml_kinv_batched")
            rows = _ops.round_up(max(n, 1), 64)
            for b in range(batch):
                blk = fb._backward_work[b * stride:(b + 1) * stride]
                kinvs[b] = blk[koff:koff + rows * ld].view(rows, ld)
                ats[b] = blk[aoff:aoff + dy * ld].view(dy, ld)
        nz = noise.detach().reshape(batch)
        for b in range(batch):
            if fb is None or b in ctx.replayed:
                f = ctx.replayed.get(b)
                if f is None:
                    f = _ops.Factor(n, dy, X.device)
                    _factor_single(progs[b], thetas[b], Xb(b), Rb(b), _c(nz[b:b + 1]), f)
                U = _backward._upper_inverse(f)
                kinvs[b] = _backward._kinv_lower(f, U)
                ats[b] = _ops.gemm_nt(f.A[n:], U, dy, n, _ops.round_up(n, 16), tri=_ops.TRI_B_UPPER)
        go = grad_out.reshape(batch)
        g_params, g_noise, g_R = [], [], []
        swept = None
        if fb is not None and not ctx.replayed and batch > 1:
            # the gradient sweeps in lock step: per leaf instance ONE sweep + one reduction launch over all models, reading every
            # model's Kyy^-1 and a^T where gpn_lml_kinv_batched left them
            prog0 = progs[0]
            theta_all = torch.stack(list(thetas))
            Xc, d = _c(X.detach()), X.shape[-1]
            wk = fb._backward_work
            work = torch.empty(max(1, batch * int(lib.gpn_kernel_expr_grad_work_bytes(n, n, d, 1)) // 8), dtype=torch.float64, device=X.device)
            per_instance = []
            for i, li in enumerate(prog0.instances):
                _, nvar, _, nls = prog0.offsets[li]
                want_trace = 1 if i == 0 else 0
                o = torch.empty(batch, nvar + nls + want_trace, dtype=torch.float64, device=X.device)
                st = lib.gpn_kernel_expr_grad_batched(_stream(X.device), prog0.terms, len(prog0.instances), prog0.gstart, prog0.ngroups, batch,
                                                      _ptr(theta_all), theta_all.shape[1], i, _ptr(Xc), 0 if X.dim() == 2 else n * d, n, d,
                                                      wk.data_ptr() + 8 * koff, ld, stride, wk.data_ptr() + 8 * aoff, ld, stride, dy, want_trace,
                                                      _ptr(work), _ptr(o))
                _native.check(st, "gpn_kernel_expr_grad_batched")
                per_instance.append((o, nvar + nls))
            swept = per_instance
        for b in range(batch):
            pb = params[b * npar:(b + 1) * npar]
            if swept is not None:
                outs = [o[b] for o, _ in swept]
                trace = swept[0][0][b, swept[0][1]:swept[0][1] + 1]
            else:
                outs, trace = _sweeps(progs[b], thetas[b], Xb(b), None, kinvs[b], kinvs[b].stride(0), at=ats[b], ldat=ats[b].stride(0), dy=dy)
            g_params += [go[b] * g for g in progs[b].scatter(outs, pb)]
            g_noise.append(go[b] * trace)
            if ctx.needs_input_grad[1]:
                g_R.append(-go[b] * ats[b][:, :n].t())
        g_resid = None
        if ctx.needs_input_grad[1]:
            g_resid = torch.stack(g_R) if R.dim() == 3 else torch.stack(g_R).sum(0)
        return (None, g_resid, torch.cat(g_noise).reshape(noise.shape) if ctx.needs_input_grad[2] else None, None, None) + tuple(g_params)

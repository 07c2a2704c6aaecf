"""
Exact GP regression (gptorch/models/gpr.py) over the native pipeline.

log_likelihood:  one autograd node = fused K(X)+sigma_n^2 I assembly into the
factor buffer -> blocked MFMA Cholesky with the residual (y - m)^T riding along
as extra rows (forward substitution for free) -> log-det / ||alpha||^2
reduction.  _predict re-uses the cached factor when neither the parameters nor
the inputs changed (the reference re-factorises on every call, gpr.py:104).
"""
import collections

import torch

from .. import _expr, _ops, mean_functions
from .. import kernels
from .base import GPModel


INVERSE_AFTER_CALLS = 3      # predictions with one cached factor before L^-1 is formed (see _predict); N < 4096 only
BLOCKED_AFTER_CALLS = 1      # ... and before the 1024 x 1024 diagonal blocks are inverted (N >= 4096)


class GPR(GPModel):
    def __init__(self, x, y, kernel, mean_function=None, likelihood=None, name="gpr"):
        super().__init__(x, y, kernel, likelihood, mean_function, name)
        self._holder = {}        # reusable factor buffer for the training loop
        self._predict_cache = None
        self._predict_calls = 0

    def _stationary(self):
        """the kernel if it is one of the native stationary kinds (fused assembly -> factor
        path), else None (composite / linear / static kernels: dense-K path)."""
        k = self.kernel
        return k if isinstance(k, kernels.Stationary) and k._kind is not None else None

    def _expression(self, x):
        """the fused-expression program of a composite kernel (gptorch_amd._expr) or None (dense-K path)."""
        k = self.kernel
        if not isinstance(k, kernels.Combination) or not x.is_cuda or x.requires_grad:
            return None
        prog = k.fused_program()
        return prog if prog is not None and prog.grad_supported(x.shape[1]) else None

    def log_likelihood(self, x=None, y=None):
        """gpr.py:47-67; returns a tensor of shape (1,)."""
        x = x if x is not None else self.X
        y = y if y is not None else self.Y
        if not x.shape[0] == y.shape[0]:
            raise ValueError("X and Y must have same # data.")
        k = self._stationary()
        resid = y - self.mean_function(x)
        if k is None:
            prog = self._expression(x)
            if prog is not None:
                # composite kernel with native leaves (e.g. the reference's example Linear + Rbf + Constant,
                # examples/regression_1d.py:34-53): fused assembly into the factor buffer, expression sweeps in the backward
                from .. import _expr
                return _expr.ExprLogLik.apply(x, resid, self.likelihood.variance.transform(), prog, self._holder, *prog.params())
            return _ops.DenseLogLik.apply(self.kernel.K(x), resid, self.likelihood.variance.transform())
        return _ops.GPRLogLik.apply(x, resid, k.variance.transform(), k.length_scales.transform(),
                                    self.likelihood.variance.transform(), k._kind, self._holder)

    def _compute_kyy(self, x=None):
        """K(x) + sigma_n^2 I as a dense tensor (gpr.py:69-86); API parity only --
        the training / predict paths never materialise it outside the factor buffer."""
        x = x if x is not None else self.X
        k = self._stationary()
        if k is None:
            Kyy = self.kernel.K(x).clone()
            Kyy.diagonal().add_(self.likelihood.variance.transform()[0])
            return Kyy
        return _ops.kernel_matrix(k._kind, x, None, k.variance.transform(), k.length_scales.transform(),
                                  noise=self.likelihood.variance.transform())

    def _factor_for_predict(self, x):
        k = self._stationary()
        with torch.no_grad():
            var, ls, noise = k.variance.transform(), k.length_scales.transform(), self.likelihood.variance.transform()
            # the factor depends on the inputs and on EVERY model parameter, mean function included.  Inputs: the cache HOLDS
            # the tensors it was built from and compares identity + version counter (large, never edited through .data) -- a
            # held tensor cannot be freed, so a temporary `x=` re-allocated at the same address with equal shape and version
            # can never pass for the cached one (round-4 review: the key used data_ptr()).  Parameters are compared by
            # value on the device (edits through `.data` do not bump a version counter): one
            # concatenation + one torch.equal = a single host sync per prediction
            key = (x._version, tuple(x.shape), self.Y._version, k._kind)
            params = torch.cat([p.detach().reshape(-1) for p in self.parameters()])
            c = self._predict_cache
            if c is None or c[0] != key or c[3] is not x or c[4] is not self.Y or c[2].shape != params.shape \
                    or not torch.equal(c[2], params):
                f = _ops.kernel_factor(k._kind, x, var, ls, noise, R=self.Y - self.mean_function(x))
                self._predict_cache = (key, f, params, x, self.Y)
                self._predict_calls = 0
            self._predict_calls += 1
        return self._predict_cache[1], var, ls

    def _predict_dense(self, x_new, diag, x):
        """gpr.py:88-117 for a kernel without a native kind: the kernel's own K() calls, the
        native factorisation / right-solves / contractions."""
        with torch.no_grad():
            n, ns = x.shape[0], x_new.shape[0]
            # the factor is kept between predictions exactly as for the native kinds (_factor_for_predict): data by identity +
            # version, every parameter by value
            key = (x._version, tuple(x.shape), self.Y._version, "dense")
            params = torch.cat([p.detach().reshape(-1) for p in self.parameters()])
            c = self._predict_cache
            if c is None or c[0] != key or c[3] is not x or c[4] is not self.Y or c[2].shape != params.shape \
                    or not torch.equal(c[2], params):
                f = _ops.cholesky_factor(self._compute_kyy(x), rhs=self.Y - self.mean_function(x))
                self._predict_cache = (key, f, params, x, self.Y)
                self._predict_calls = 0
            self._predict_calls += 1
            f = self._predict_cache[1]
            Bt = _ops.padded_like_factor(f, ns)
            Bt[:ns, :n] = self.kernel.K(x_new, x)
            f.solve_right_lt(Bt, ns)                                        # A^T = K(x*, x) L^-T
            mean = _ops.gemm_nt(Bt, f.A[n:], ns, f.e, _ops.round_up(n, 16)) + self.mean_function(x_new)
            if diag:
                v = self.kernel.Kdiag(x_new) - _ops.row_sumsq(Bt, ns, n)
                return mean, v[:, None].expand_as(mean)
            cov = self.kernel.K(x_new).clone()
            _ops.gemm_nt(Bt, Bt, ns, ns, _ops.round_up(n, 16), alpha=-1.0, beta=1.0, C=cov)
            return mean, cov

    def _predict(self, x_new, diag=True, x=None):
        """p(F* | Y) (gpr.py:88-117): mean [n*, dy]; var [n*, dy] (diag) or cov [n*, n*]."""
        x = x if x is not None else self.X
        k = self._stationary()
        if k is None:
            return self._predict_dense(x_new, diag, x)
        f, var, ls = self._factor_for_predict(x)
        with torch.no_grad():
            # The first prediction with a factor walks the right-solve recursion down to the 128-wide leaf inverses (~2 n / 128
            # small launches).  From the second on (N >= 4096): the inverses of the 1024 x 1024 diagonal blocks are formed once
            # (n 1024^2 / 3 flops) and the solve is n / 1024 steps of two large contractions (2.3 -> 1.3 ms at N = 8192, 1024
            # test points).  Below 4096 rows: from the third prediction on, one contraction with the explicit inverse.
            big = x.shape[0] >= _ops.BLOCKED_PREDICT_MIN_N
            mean_f, v = _ops.gpr_predict(k._kind, x, x_new, var, ls, f, diag=diag,
                                         use_inverse=(not big) and self._predict_calls >= INVERSE_AFTER_CALLS,
                                         mean_new=self.mean_function(x_new), blocked=big and self._predict_calls > BLOCKED_AFTER_CALLS)
            var_f = v[:, None].expand_as(mean_f) if diag else v
        return mean_f, var_f


_LANES = {}          # device -> two HIP streams shared by every batched call (streams= placement only)

# Lock-step buffers reused between calls (a search calls batched_log_likelihood / batched_loss_and_grad once per optimiser
# step): keyed by the FULL group key + batch size -- two groups of one call can never share an entry (round-4 advice: keyed
# by (device, batch, n, dy) only, two groups of equal count / N / dy but different kernel kind or ARD shared one buffer and
# the first group read the second group's results) -- least recently used first out, bounded in entries and bytes;
# release_batch_buffers() drops them all.
# scipy methods whose `minimize` holds a module-wide lock while it runs (one restart at a time whatever we do)
_SCIPY_SERIAL_METHODS = ("COBYLA",)
_MULTI_START_STALL_S = 2.0       # a round of the scipy multi-start stops waiting for silent restarts after this long
_MULTI_START_JOIN_S = 10.0


class _MultiStartAborted(RuntimeError):
    """raised inside a restart's objective when the multi-start search it belongs to has ended (interrupt, error)."""


_BATCH_BUFFERS = collections.OrderedDict()    # (group key, batch) -> holder dict {"fb": _ops.FactorBatch}
BATCH_BUFFER_MAX_ENTRIES = 4
BATCH_BUFFER_MAX_BYTES = 48 << 30


def _batch_holder(key):
    h = _BATCH_BUFFERS.pop(key, None)
    if h is None:
        h = {}
    _BATCH_BUFFERS[key] = h                     # most recently used last
    def total():
        return sum(v["fb"].nbytes() for v in _BATCH_BUFFERS.values() if "fb" in v)
    while len(_BATCH_BUFFERS) > 1 and (len(_BATCH_BUFFERS) > BATCH_BUFFER_MAX_ENTRIES or total() > BATCH_BUFFER_MAX_BYTES):
        _BATCH_BUFFERS.popitem(last=False)
    return h


def release_batch_buffers():
    """free the factor buffers batched_log_likelihood / batched_loss_and_grad keep between calls."""
    _BATCH_BUFFERS.clear()


def _stacked_values(params):
    """constrained values of same-shaped Params as one [B, ...] tensor: one stack + ONE transform when they share it."""
    t0 = params[0]._transform
    if all(p._transform == t0 for p in params):
        return t0(torch.stack([p.data for p in params]))
    return torch.stack([p.transform() for p in params])


def _place_all(models):
    """settings.auto_device: CPU-constructed models move to the GPU (once) BEFORE they are grouped -- the grouping looks at
    m.X.is_cuda, and a model that is still on the CPU would silently fall out of every lock-step group."""
    for m in models:
        place = getattr(m, "_auto_place", None)
        if place is not None:
            place()


def _group_key(m):
    k = m._stationary()
    return (k._kind, tuple(m.X.shape), m.Y.shape[1], int(k.length_scales.numel()), m.X.device)


RAGGED_MIN_FRACTION = 0.75      # a ragged group's smallest model has at least this fraction of its largest model's rows
RAGGED_POOL_BELOW = 8            # equal-size groups of fewer models than this may merge with neighbouring sizes into one ragged group


def _panel_regime(n):
    """models whose sizes select the same panel levels of the factorisation (gpn_potrf_panel_levels) -- and none of which is refined --
    can be padded into one ragged lock-step group; None: no ragged group for this size"""
    if n <= 2 * _ops.LEAF or n >= _ops.refine_min_n():
        return None
    return 0 if n <= 2048 else 1 if n < 20480 else 2


def _lockstep_groups(models, for_grad=False):
    """[(key, indices)] of the models that can share one lock-step call, grouped by (kernel kind, n, d, dy, ARD, device):
    GPR over a native stationary kernel.  for_grad (the stacked-parameter
    optimiser loop of multi_start_optimize): also no priors (loss() = -(LML + log prior), model.py:158-197, is formed per model).
    Models left alone by that (cross-validation folds of unequal length, learning curves) form RAGGED groups: same kind / d / dy / ARD,
    zero mean, different n within one panel regime, each padded to the group's largest model with identity rows
    (_ops.lml_forward_batched(n_of=...)); their key carries the sizes as a sixth entry."""
    groups = {}
    for i, m in enumerate(models):
        # GPR's own log_likelihood only: other GPModels (VFE), and subclasses that evaluate differently (DistGPR: collective, on the
        # process grid), take their own path
        if not isinstance(m, GPR) or type(m).log_likelihood is not GPR.log_likelihood:
            continue
        k = m._stationary()
        if k is None or not m.X.is_cuda or m.X.shape[0] == 0:
            continue
        if for_grad and any(getattr(p, "prior", None) is not None for p in m.parameters()):
            continue
        groups.setdefault(_group_key(m), []).append(i)
    out, pool = [], {}
    for key, g in groups.items():
        # Small equal-size groups and singletons of ragged-eligible models are POOLED: folds of n and n - 1 rows make one ragged group
        # of all of them rather than two small groups.  (Groups of RAGGED_POOL_BELOW models or more stay as they are -- multi-start
        # restarts on one data set share its tensors --, and so does everything the stacked optimiser loop asks for.)
        eligible = not for_grad and len(g) < RAGGED_POOL_BELOW and _panel_regime(key[1][0]) is not None and \
            all(type(models[i].mean_function) is mean_functions.Zero for i in g)
        if eligible:
            pool.setdefault((key[0], key[1][1], key[2], key[3], key[4], _panel_regime(key[1][0])), []).extend(g)
            continue
        # a lock-step group holds B factor buffers AND (with gradients) a backward workspace of two more N x N matrices per model at
        # once: groups that would not fit the device's free memory are split into chunks that do (singletons fall to the sequential path)
        cap = _lockstep_capacity(key, models[g[0]].X.device)
        for at in range(0, len(g), cap):
            chunk = g[at:at + cap]
            if len(chunk) >= 2:
                out.append((key, chunk))
    for (kind, d, dy, nls, dev, _regime), g in pool.items():
        g = sorted(g, key=lambda i: (-models[i].X.shape[0], i))
        at = 0
        while at < len(g):
            nmax = models[g[at]].X.shape[0]
            end = at + 1
            cap = _lockstep_capacity((kind, (nmax, d)), dev)
            while end < len(g) and end - at < cap and models[g[end]].X.shape[0] >= RAGGED_MIN_FRACTION * nmax:
                end += 1
            if end - at >= 2:
                chunk = sorted(g[at:end])
                sizes = tuple(models[i].X.shape[0] for i in chunk)
                if len(set(sizes)) == 1:                     # all of one size after all: the ordinary equal-size group
                    out.append((_group_key(models[chunk[0]]), chunk))
                else:
                    out.append(((kind, (nmax, d), dy, nls, dev, sizes), chunk))
            at = end
    return out


LOCKSTEP_MEMORY_FRACTION = 0.7   # of the device's free memory (+ what the lock-step cache already holds) a group may take


def _lockstep_capacity(key, device):
    """how many models of this group key fit one lock-step call: 3 padded N x N matrices per model (factor + the backward's two)."""
    n = int(key[1][0])
    ld = -(-n // 128) * 128 + 128
    per_model = 3 * 8 * (ld + 16) * ld
    try:
        free, _total = torch.cuda.mem_get_info(device)
    except Exception:
        return 1 << 30
    held = sum(v["fb"].nbytes() for v in _BATCH_BUFFERS.values() if "fb" in v)
    return max(2, int(LOCKSTEP_MEMORY_FRACTION * (free + held)) // per_model)


def _expression_groups(models):
    """[(key, indices, programs)] of the GPR models over COMPOSITE kernels of one structure (equal _expr.Program.signature(),
    n, d, dy, device) that can share the kernel-independent launches of an evaluation (_expr.BatchedExprLogLik)."""
    groups = {}
    for i, m in enumerate(models):
        if not isinstance(m, GPR) or type(m).log_likelihood is not GPR.log_likelihood or m._stationary() is not None:
            continue
        if not m.X.is_cuda or m.X.shape[0] == 0:
            continue
        prog = m._expression(m.X)
        if prog is None:
            continue
        key = ("expr", prog.signature(), tuple(m.X.shape), m.Y.shape[1], m.X.device)
        groups.setdefault(key, ([], []))
        groups[key][0].append(i)
        groups[key][1].append(prog)
    return [(key, g, progs) for key, (g, progs) in groups.items() if len(g) >= 2]


def _vfe_groups(models):
    """[(key, indices)] of the VFE models (sparse_gpr.py:108-153) that can share one lock-step evaluation
    (_vfe_lockstep.BatchedVFEBound): one native stationary kind, equal (N, D, dy, M, ARD), on one device, in the single-chunk regime
    (_vfe_lockstep.supported); split into chunks that fit the device's free memory."""
    from . import _vfe_lockstep
    from .sparse_gpr import VFE
    groups = {}
    for i, m in enumerate(models):
        if not isinstance(m, VFE) or type(m).log_likelihood is not VFE.log_likelihood or type(m)._bound is not VFE._bound:
            continue
        k = m._native_kernel()
        if k is None or not m.X.is_cuda or not _vfe_lockstep.supported(m.X.shape[0], m.Z.shape[0]):
            continue
        key = ("vfe", k._kind, tuple(m.X.shape), m.Y.shape[1], int(k.length_scales.numel()), tuple(m.Z.shape), m.X.device)
        groups.setdefault(key, []).append(i)
    out = []
    for key, g in groups.items():
        try:
            free, _total = torch.cuda.mem_get_info(key[-1])
            cap = max(2, int(LOCKSTEP_MEMORY_FRACTION * free) // _vfe_lockstep.per_model_bytes(key[2][0], key[5][0], key[3]))
        except Exception:
            cap = 1 << 30
        for at in range(0, len(g), cap):
            chunk = g[at:at + cap]
            if len(chunk) >= 2:
                out.append((key, chunk))
    return out


def _vfe_group_bound(ms, key, differentiable):
    """the lock-step bounds [B] of one _vfe_groups group (autograd-connected to every model's Params when differentiable)"""
    from . import _vfe_lockstep
    B = len(ms)
    m0 = ms[0]
    same_x = all(m.X.data_ptr() == m0.X.data_ptr() for m in ms)
    same_y = same_x and all(m.Y.data_ptr() == m0.Y.data_ptr() for m in ms)
    X = m0.X if same_x else torch.stack([m.X for m in ms])
    Y = m0.Y if same_y else torch.stack([m.Y for m in ms])               # sparse_gpr.py:125 quirk: err = Y (Zero mean only)
    plists = ([m.kernel.variance for m in ms], [m.kernel.length_scales for m in ms], [m.likelihood.variance for m in ms])
    stacks = []
    for plist in plists:
        t0 = _shared_transform(plist)
        if not differentiable:
            stacks.append(_stacked_values(plist))
        elif t0 is not None:
            stacks.append(t0(torch.stack(list(plist))))
        else:
            stacks.append(torch.stack([p.transform() for p in plist]))
    Z = torch.stack([m.Z for m in ms]) if differentiable else torch.stack([m.Z.data for m in ms])
    return _vfe_lockstep.BatchedVFEBound.apply(stacks[0].reshape(B), stacks[1].reshape(B, -1), stacks[2].reshape(B), Z, key[1], X, Y)


def _group_data(ms, differentiable=False, key=None):
    """(X, R, n_of) of a lock-step group: shared [n, d] / [n, dy] when every model holds the same tensors (restarts on one data
    set), else stacked [B, ...].  differentiable: R keeps the autograd graph of trainable mean functions.
    A ragged group (key with a sixth entry: the sizes): X, R padded to the largest model, n_of = the sizes on the device (int32);
    otherwise n_of is None."""
    if key is not None and len(key) > 5:
        # (data and zero-mean right-hand sides do not change between the iterations of a search: the padded stacks are kept with
        #  the group's lock-step buffers and rebuilt when a model's tensors are replaced or edited in place)
        holder = _batch_holder((key, len(ms)))
        stamp = tuple((id(m.X), m.X._version, id(m.Y), m.Y._version) for m in ms)
        cached = holder.get("ragged_data")
        if cached is not None and cached[0] == stamp:
            return cached[1], cached[2], cached[3]
        nmax, B = key[1][0], len(ms)
        X = torch.zeros(B, nmax, key[1][1], dtype=torch.float64, device=key[4])
        R = torch.zeros(B, nmax, key[2], dtype=torch.float64, device=key[4])
        for b, m in enumerate(ms):
            X[b, :m.X.shape[0]] = m.X
            R[b, :m.X.shape[0]] = m.Y
        n_of = torch.tensor(key[5], dtype=torch.int32, device=key[4])
        holder["ragged_data"] = (stamp, X, R, n_of, [(m.X, m.Y) for m in ms])     # (holds the tensors: an id() cannot be reused)
        return X, R, n_of
    m0 = ms[0]
    same_x = all(m.X.data_ptr() == m0.X.data_ptr() for m in ms)
    zero_mean = all(type(m.mean_function) is mean_functions.Zero for m in ms)
    same_r = same_x and zero_mean and all(m.Y.data_ptr() == m0.Y.data_ptr() for m in ms)
    X = m0.X if same_x else torch.stack([m.X for m in ms])
    if same_r:
        R = m0.Y
    elif zero_mean:
        R = torch.stack([m.Y for m in ms])
    elif differentiable:
        R = torch.stack([m.Y - m.mean_function(m.X) for m in ms])
    else:
        with torch.no_grad():
            R = torch.stack([m.Y - m.mean_function(m.X) for m in ms])
    return X, R, None


def batched_log_likelihood(models, streams=None):
    """log_likelihood() of several INDEPENDENT GPR models (multi-start hyper-parameter search: one model per restart; the
    reference evaluates them one per optimiser step, gptorch/models/base.py:260-269).  No gradients (batched_loss_and_grad
    has them); returns a list of (1,) tensors, each BIT-IDENTICAL to that model's own log_likelihood().

    streams=None (default): models of one shape (kernel kind, N, D, dy) run in LOCK STEP through ONE
    gpn_lml_forward_batched call -- one assembly launch, the 128x128 leaf as a grid of B workgroups, every column pass and
    contraction as a strided-batch launch -- and the `info` words are read once at the end; a model whose factorisation
    reports info != 0 is re-evaluated through the sequential path (jitter ladder of functions.py:20-43); from
    refine_min_n() rows on every model's quadratic form is refined as log_likelihood() refines it.  Dense-K / composite
    kernels and singletons take the sequential path.

    streams = a list of HIP streams (one per model): the round-3 placement instead -- whole evaluations alternating
    over the given streams (the current stream itself gives back-to-back execution)."""
    if streams is not None:
        return _batched_on_streams(models, streams)
    _place_all(models)
    out = [None] * len(models)
    with torch.no_grad():
        pending = []
        for key, g in _lockstep_groups(models):
            ms = [models[i] for i in g]
            # host side: a handful of launches per GROUP, none per model (a per-model exp / subtraction / comparison costs
            # more than the model's share of the batch at N = 512)
            X, R, n_of = _group_data(ms, key=key)
            var = _stacked_values([m._stationary().variance for m in ms]).reshape(len(ms))
            ls = _stacked_values([m._stationary().length_scales for m in ms]).reshape(len(ms), -1)
            nz = _stacked_values([m.likelihood.variance for m in ms]).reshape(len(ms))
            holder = _batch_holder((key, len(ms)))
            fb, terms = _ops.lml_forward_batched(key[0], X, R, var, ls, nz, fb=holder.get("fb"),
                                                 refine=n_of is None and ms[0].X.shape[0] >= _ops.refine_min_n(), n_of=n_of)
            holder["fb"] = fb
            pending.append((g, fb, terms))
        for g, fb, terms in pending:
            info = fb.info.cpu().tolist()          # one read-back per group (synchronises the stream)
            vals = terms[:, 2:3].clone()           # ONE copy out of the shared buffers; every model gets its row of it
            for b, i in enumerate(g):
                if info[b] == 0:
                    out[i] = vals[b]
                    # (the per-model factor cache is NOT pointed at the shared buffer: the next batched call overwrites it)
        for key, g, progs in _expression_groups(models):
            ms = [models[i] for i in g]
            X, R, _ = _group_data(ms)
            nz = _stacked_values([m.likelihood.variance for m in ms]).reshape(len(ms))
            flat = [p for prog in progs for p in prog.params()]
            lml = _expr.BatchedExprLogLik.apply(X, R, nz, progs, _batch_holder((key, len(ms))), *flat)
            for b, i in enumerate(g):
                out[i] = lml[b:b + 1].clone()
        for key, g in _vfe_groups(models):
            elbo = _vfe_group_bound([models[i] for i in g], key, differentiable=False)
            for b, i in enumerate(g):
                out[i] = elbo[b]                                             # (VFE.log_likelihood returns a 0-dim tensor)
        for i, m in enumerate(models):
            if out[i] is None:
                out[i] = m.log_likelihood()
    return out


def batched_factorise(models):
    """The factorisations the predictions of several models start from -- chol(Kyy) with L^-1 (y - m) riding along, gpr.py:104-106 --
    in LOCK STEP (cross-validation scoring: k fitted folds, each about to predict its held-out rows; the reference re-factorises
    inside every _predict call, one model at a time).  Models of one shape (kind, N, D, dy, ARD) share ONE lock-step forward into
    buffers of their own; every model's factor cache is then seeded with its slice, so its next predict_f / predict_y /
    predict_*_samples goes straight to the solve -- with the factor, and therefore the predictions, bit-identical to what the model
    would have computed alone.  A model whose factorisation needs the jitter ladder, and everything no group takes, is left to
    its own _predict.  Returns the number of models seeded."""
    _place_all(models)
    seeded = 0
    with torch.no_grad():
        for key, g in _lockstep_groups(models):
            if len(key) > 5:                         # ragged groups: a padded factor is not the layout _predict's entry points take
                continue
            ms = [models[i] for i in g]
            X, R, _ = _group_data(ms)
            var = _stacked_values([m._stationary().variance for m in ms]).reshape(len(ms))
            ls = _stacked_values([m._stationary().length_scales for m in ms]).reshape(len(ms), -1)
            nz = _stacked_values([m.likelihood.variance for m in ms]).reshape(len(ms))
            fb, _terms = _ops.lml_forward_batched(key[0], X, R, var, ls, nz, fb=None)      # buffers of their own: the caches keep them
            info = fb.info.cpu()
            for b, m in enumerate(ms):
                if int(info[b]) != 0:
                    continue
                k = m._stationary()
                ckey = (m.X._version, tuple(m.X.shape), m.Y._version, k._kind)
                params = torch.cat([p.detach().reshape(-1) for p in m.parameters()])
                m._predict_cache = (ckey, fb.factor(b), params, m.X, m.Y)
                m._predict_calls = 0
                seeded += 1
    return seeded


def _group_param_lists(ms):
    return ([m._stationary().variance for m in ms], [m._stationary().length_scales for m in ms],
            [m.likelihood.variance for m in ms])


def _shared_transform(params):
    t0 = params[0]._transform
    return t0 if all(p._transform == t0 for p in params) else None


def _has_priors(ms):
    return any(getattr(p, "prior", None) is not None for m in ms for p in m.parameters())


def _plan_groups(models):
    """the grouping of batched_loss_and_grad for a list of models: [(lock-step groups), (expression groups), (sparse groups)] with each
    group's "has priors" flag.  Shapes, kernels and priors do not change while a search runs: multi_start_optimize plans ONCE and
    hands the plan to every iteration (the grouping walks every model's parameters and, for composite kernels, rebuilds their
    expression programs: host time that a small-N iteration would spend several times over)."""
    _place_all(models)
    return ([(key, g, _has_priors([models[i] for i in g])) for key, g in _lockstep_groups(models)],
            [(key, g, progs, _has_priors([models[i] for i in g])) for key, g, progs in _expression_groups(models)],
            [(key, g, _has_priors([models[i] for i in g])) for key, g in _vfe_groups(models)])


def batched_loss_and_grad(models, _plan=None):
    """`loss = m.loss(); loss.backward()` for several INDEPENDENT GPR models -- the body of the reference's optimiser step
    (gptorch/models/base.py:260-269: `closure()`), which the reference can only run one model at a time.  Gradients are
    ACCUMULATED into every trainable parameter's `.grad` exactly as backward() does; returns the list of detached (1,) loss
    tensors.

    Models of one shape (kernel kind, N, D, dy, ARD) run in LOCK STEP: one gpn_lml_forward_batched + one
    gpn_lml_backward_batched call per group (_ops.BatchedGPRLogLik), the hyper-parameters of the group stacked so that the
    transforms and their chain rule are one small launch per parameter kind.  Each model's loss AND gradients are
    BIT-IDENTICAL to its own `loss(); backward()`; a model whose factorisation fails is replayed alone through the jitter
    ladder; parameters with priors add their model's own log_prior() (model.py:158-197); sizes that refine the quadratic
    form refine it per model.  Composite / dense-K kernels and singletons take the sequential path."""
    plan = _plan if _plan is not None else _plan_groups(models)
    out = [None] * len(models)
    for key, g, priors in plan[0]:
        ms = [models[i] for i in g]
        B = len(ms)
        X, R, n_of = _group_data(ms, differentiable=True, key=key)
        stacks = []
        for plist in _group_param_lists(ms):
            t0 = _shared_transform(plist)
            if t0 is not None:
                stacks.append(t0(torch.stack(list(plist))))          # StackBackward hands every Param its own gradient row
            else:
                stacks.append(torch.stack([p.transform() for p in plist]))
        var, ls, nz = stacks[0].reshape(B), stacks[1].reshape(B, -1), stacks[2].reshape(B)
        holder = _batch_holder((key, B))
        if n_of is not None:
            holder["sizes"] = key[5]
        lml = _ops.BatchedGPRLogLik.apply(X, R, var, ls, nz, key[0], holder, n_of)
        if priors:
            # parameters with priors (model.py:158-197: loss = -(LML + log prior)): each model's own log_prior(), added to its
            # entry of the lock-step LML exactly as Model._loss adds it
            loss = torch.cat([-(lml[b:b + 1] + m.log_prior()) for b, m in enumerate(ms)])
        else:
            loss = -(lml + 0.0)                                      # model.py:_loss with an empty log prior
        if loss.requires_grad:
            loss.sum().backward()
        ld = loss.detach()
        for b, i in enumerate(g):
            out[i] = ld[b:b + 1]
    for key, g, progs, priors in plan[1]:
        # composite kernels of one structure (the reference's example model Linear + Rbf + Constant in a multi-start search):
        # the expression's assembly and sweeps per model, everything kernel-independent once over the group
        ms = [models[i] for i in g]
        B = len(ms)
        X, R, _ = _group_data(ms, differentiable=True)
        plist = [m.likelihood.variance for m in ms]
        t0 = _shared_transform(plist)
        nz = (t0(torch.stack(list(plist))) if t0 is not None else torch.stack([p.transform() for p in plist])).reshape(B)
        flat = [p for prog in progs for p in prog.params()]
        lml = _expr.BatchedExprLogLik.apply(X, R, nz, progs, _batch_holder((key, B)), *flat)
        if priors:
            loss = torch.cat([-(lml[b:b + 1] + m.log_prior()) for b, m in enumerate(ms)])
        else:
            loss = -(lml + 0.0)
        if loss.requires_grad:
            loss.sum().backward()
        ld = loss.detach()
        for b, i in enumerate(g):
            out[i] = ld[b:b + 1]
    for key, g, priors in plan[2]:
        # sparse models of one shape (sparse_gpr.py:108-153 in a multi-start search over inducing points / hyper-parameters)
        ms = [models[i] for i in g]
        elbo = _vfe_group_bound(ms, key, differentiable=True)
        # Model.loss (model.py:158-197): -(bound + log prior), model by model as the sequential code forms it
        if priors:
            loss = torch.stack([-(elbo[b] + m.log_prior()) for b, m in enumerate(ms)])
        else:
            loss = -(elbo + 0.0)
        if loss.requires_grad:
            loss.sum().backward()
        ld = loss.detach()
        for b, i in enumerate(g):
            out[i] = ld[b]
    for i, m in enumerate(models):
        if out[i] is None:
            loss = m.loss()
            if loss.requires_grad:
                loss.backward()
            out[i] = loss.detach()
    return out


def _multi_start_scipy(models, method, max_iter, verbose):
    """scipy.optimize.minimize for every model AT ONCE (base.py:298-320; what examples/regression_1d.py:53 and the
    reference's notebooks run is L-BFGS-B): each restart's `minimize` runs in its own host thread and only ever waits -- its
    `fun(x)` posts the parameter vector it wants evaluated and sleeps; the calling thread collects one request per
    still-running restart, evaluates ALL of them in one batched_loss_and_grad call (lock-step groups + sequential rest),
    and hands every restart its (loss, gradient).  Line searches make the restarts ask for different numbers of
    evaluations, and restarts finish at different iterations: a round simply covers whoever is still running.  Each
    restart sees exactly the values Model._loss_and_grad (model.py:123-133) would have given it -- bit for bit -- so its
    iterates, its result and its printed losses are those of its own optimize(); only the order in which the restarts'
    "loss: ..." lines interleave differs.  -> list of scipy OptimizeResult."""
    import threading
    import numpy as np
    from scipy.optimize import minimize
    B = len(models)
    if method in _SCIPY_SERIAL_METHODS:
        # scipy runs these under a module-wide lock (COBYLA: scipy.optimize._cobyla_py._module_lock): a second restart's
        # `minimize` cannot even start while the first one sits in its objective, so the restarts cannot post requests together.
        # One after the other through each model's own optimize() -- what the reference does (base.py:298-320).
        return [m.optimize(method=method, max_iter=max_iter, verbose=verbose) for m in models]
    cond = threading.Condition()
    pending, answers = {}, {}
    done = [False] * B
    results = [None] * B
    state = {"abort": None}
    x0 = [m._get_param_array() for m in models]

    def make_fun(i):
        def fun(x):
            with cond:
                if state["abort"] is not None:
                    raise _MultiStartAborted(state["abort"])
                pending[i] = np.array(x, dtype=np.float64, copy=True)
                cond.notify_all()
                while i not in answers and state["abort"] is None:
                    cond.wait()
                if i not in answers:
                    pending.pop(i, None)
                    raise _MultiStartAborted(state["abort"])
                ans = answers.pop(i)
            if isinstance(ans, BaseException):
                raise ans
            return ans
        return fun

    def worker(i):
        try:
            results[i] = minimize(fun=make_fun(i), x0=x0[i], method=method, jac=True, tol=None, callback=None,
                                  options=dict(disp=verbose, maxiter=max_iter))
        except BaseException as exc:             # delivered to the caller after every restart has finished
            results[i] = exc
        finally:
            with cond:
                done[i] = True
                cond.notify_all()

    threads = [threading.Thread(target=worker, args=(i,), daemon=True) for i in range(B)]
    for t in threads:
        t.start()
    try:
        while True:
            with cond:
                stalled = False
                while True:
                    active = [i for i in range(B) if not done[i]]
                    if not active or all(i in pending for i in active):
                        break
                    # a restart that neither finishes nor posts (a method that serialises inside scipy, a callback that blocks):
                    # after the stall time-out the round covers whoever HAS posted -- never a dead wait
                    if not cond.wait(timeout=_MULTI_START_STALL_S) and pending:
                        stalled = True
                        break
                if not active:
                    break
                batch = {i: pending.pop(i) for i in (active if not stalled else sorted(pending))}
            idx = sorted(batch)
            out = {}
            try:
                # Model._loss_and_grad (model.py:123-133) for all requests of the round at once.  The requested vectors travel to
                # the device as ONE copy and every parameter becomes a slice of it (model.py:66-76 makes one tensor per parameter:
                # 3 small copies per model and round); the gradients come back as ONE copy.
                dev = models[idx[0]].X.device
                flat = torch.as_tensor(np.concatenate([batch[i] for i in idx]), dtype=torch.float64).to(dev)
                at = 0
                for i in idx:
                    for p in models[i].parameters():
                        if p.requires_grad:
                            nxt = at + p.numel()
                            p.data = flat[at:nxt].reshape(p.shape)
                            at = nxt
                        p.grad = None                    # (a fresh gradient: what zeroing + accumulating gives)
                losses = batched_loss_and_grad([models[i] for i in idx])
                trainable = [[p for p in models[i].parameters() if p.requires_grad] for i in idx]
                allg = torch.cat([p.grad.reshape(-1) for ps in trainable for p in ps] + [l.reshape(-1) for l in losses]).cpu().numpy()
                lvals = allg[len(allg) - len(idx):]
                at = 0
                staged = {}
                for k, i in enumerate(idx):
                    cnt = sum(p.numel() for p in trainable[k])
                    staged[i] = (float(lvals[k]), np.array(allg[at:at + cnt]))
                    at += cnt
                for i in idx:                            # (nothing is printed before the whole round has its values)
                    value, grad = staged[i]
                    print("loss: %s" % value)
                    finite = np.isfinite(grad)
                    if np.all(finite):
                        out[i] = (value, grad.astype(np.float64))
                    else:
                        print("Warning: inf or nan in gradient: replacing with zeros")
                        out[i] = (value, np.where(finite, grad, 0.0).astype(np.float64))
            except Exception:
                # one request of the round failed (e.g. the jitter ladder ran out for one model): evaluate them one by one so
                # that only the restart it belongs to sees the exception.  (KeyboardInterrupt / SystemExit are not caught here:
                # they end the whole search through the `finally` below.)
                out = {}
                for i in idx:
                    try:
                        out[i] = models[i]._loss_and_grad(batch[i])
                    except Exception as exc:
                        out[i] = exc
            with cond:
                answers.update(out)
                cond.notify_all()
    except BaseException as exc:
        with cond:
            state["abort"] = exc
        raise
    finally:
        # whatever ended the collecting loop, no worker stays behind waiting for an answer: every pending and every future
        # request of a restart that is still running is answered with _MultiStartAborted, its `minimize` unwinds, its thread ends
        with cond:
            if state["abort"] is None and not all(done):
                state["abort"] = RuntimeError("multi-start search ended early")
            cond.notify_all()
        for t in threads:
            t.join(timeout=_MULTI_START_JOIN_S)
    for r in results:
        if isinstance(r, BaseException):
            raise r
    return results


def _captured_lockstep_loop(model, method, trainable, learning_rate, step, dev_losses, max_iter):
    """multi_start_optimize(capture=True) for one stacked group: `step(optimizer, idx)` (zero_grad, transforms, lock-step loss + backward, optimiser
    step, loss row idx) captured into ONE hipGraph after GPModel.CAPTURE_WARMUP eager steps and replayed; see GPModel._optimize_captured
    (the single-model form: same warm-up on a side stream, fp64 step counters, device-side loss index, chunked info flag with rollback)."""
    import inspect
    dev = dev_losses.device
    prev_dtype = torch.get_default_dtype()
    torch.set_default_dtype(torch.float64)
    try:
        optimizer = model._make_optimizer(method, trainable, learning_rate)
        if "capturable" in inspect.signature(type(optimizer).__init__).parameters:
            for g in optimizer.param_groups:
                g["capturable"] = True

        def eager(idx):
            step(optimizer, idx, set_to_none=True)
        counter = torch.zeros(1, dtype=torch.long, device=dev)
        done = 0
        side = torch.cuda.Stream(device=dev)
        side.wait_stream(torch.cuda.current_stream(dev))
        with torch.cuda.stream(side):
            while done < min(model.CAPTURE_WARMUP, max_iter):
                eager(done)
                done += 1
        torch.cuda.current_stream(dev).wait_stream(side)
        if done < max_iter:
            counter.fill_(done)
            optimizer.zero_grad(set_to_none=True)
            deferred = _ops.DeferredInfo(dev)
            graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(graph), deferred:
                loss = step(optimizer, None, set_to_none=True)
                dev_losses.index_copy_(0, counter, loss.detach().reshape(1, -1))
                counter.add_(1)

            def snapshot():
                return ([p.detach().clone() for p in trainable],
                        [{k: (v.clone() if torch.is_tensor(v) else v) for k, v in optimizer.state[p].items()} for p in trainable])

            def restore(snap):
                with torch.no_grad():
                    for p, v in zip(trainable, snap[0]):
                        p.copy_(v)
                    for p, st in zip(trainable, snap[1]):
                        for k, v in st.items():
                            if torch.is_tensor(v):
                                optimizer.state[p][k].copy_(v)          # IN PLACE: the graph holds these addresses
                            else:
                                optimizer.state[p][k] = v
            while done < max_iter:
                chunk = min(model.CAPTURE_CHUNK, max_iter - done)
                snap = snapshot()
                deferred.flag.zero_()
                for _ in range(chunk):
                    graph.replay()
                if int(deferred.flag.item()) != 0:                      # ONE read-back per chunk
                    restore(snap)
                    for k in range(chunk):
                        eager(done + k)
                    counter.fill_(done + chunk)
                done += chunk
    finally:
        torch.set_default_dtype(prev_dtype)
    return optimizer


STACKED_MAX_N = 2048      # multi_start_optimize(stacked=None): stacked parameter tensors below this many rows, one optimiser per model from it on


def multi_start_optimize(models, method="Adam", max_iter=2000, verbose=False, learning_rate=None, stacked=None, capture=False):
    """GPModel.optimize (gptorch/models/base.py:111-296) for several INDEPENDENT restarts at once: every iteration is ONE
    lock-step loss + backward over each group of equally shaped models (see batched_loss_and_grad) and ONE optimiser step
    on the group's STACKED raw parameters -- the torch optimisers the reference offers are elementwise (all but LBFGS), so
    every restart follows the trajectory its own `optimize()` would.  Given equal parameters the losses and gradients are
    bit-identical to the sequential ones; the optimiser step itself is PyTorch's multi-tensor kernel, which rounds
    `p + value * (a / b)` with or without a fused multiply-add depending on a tensor's size and alignment (measured: 1 ulp
    on a parameter after 2 Adam steps for [4, 1] against [1]), so trajectories agree to ~1e-12 relative, not bit for bit.
    stacked=False: no stacked parameter tensors at all -- every model keeps its own optimiser and only the evaluation is shared:
    trajectories BIT-IDENTICAL to each model's own optimize(), at one optimiser step per model and iteration of host work
    (C1 x 64: 6 ms per iteration instead of 1; immaterial from N = 2048 on).
    stacked=None (the default): the bitwise mode wherever it is free -- groups of models with at least STACKED_MAX_N (2048) rows
    keep one optimiser per model, smaller ones are stacked.
    capture=True: the stacked groups' iteration -- transforms, lock-step evaluation, closed-form backward, the optimiser's
    `capturable` step, the loss row -- is captured into ONE hipGraph after three eager steps and replayed (GPModel.optimize(
    capture=True) for B restarts at once: base.py:260-269 without a host round trip per iteration); `info != 0` is OR-ed into a
    device flag read every 25 replays, and a chunk that saw one is rolled back and repeated eagerly through the jitter ladder.
    Trajectories agree with the uncaptured stacked loop to rounding (the optimiser's bias corrections are formed on the device).
    Returns (losses [len(models), max_iter] numpy, seconds).  The models' Params hold the final values afterwards.

    Stacked groups: as batched_loss_and_grad's stationary groups, and additionally every model of the group trains the same
    subset of (variance, length_scales, noise) with a shared transform, no priors and no trainable mean function.  Everything
    else that batched_loss_and_grad can still evaluate together (composite kernels of one structure, models with priors or
    trainable means) keeps ONE OPTIMISER PER MODEL and shares only the evaluation: those trajectories are bit-identical to
    each model's own optimize().  method="LBFGS" (a closure-driven line search per model) and models nothing can be shared
    with are optimised one after the other by their own optimize().

    scipy methods ("L-BFGS-B", "CG", "BFGS" ...: base.py:203-215, 298-320): every restart's scipy.optimize.minimize runs at
    once and each round of function evaluations is ONE batched_loss_and_grad call (_multi_start_scipy); returns
    (list of scipy results, seconds) -- each bit-identical to the model's own optimize(method=...)."""
    import time
    import numpy as np
    from .base import _TORCH_DEFAULT_LR, _SCIPY_METHODS
    _place_all(models)
    if method in _SCIPY_METHODS:
        print("Scipy.optimize.minimize...")
        tic = time.time()
        try:
            return _multi_start_scipy(models, method, max_iter, verbose), time.time() - tic
        finally:
            release_batch_buffers()
    if learning_rate is None and method in _TORCH_DEFAULT_LR:
        learning_rate = _TORCH_DEFAULT_LR[method]
    losses = np.zeros((len(models), max_iter))
    done = [False] * len(models)
    tic = time.time()
    groups = _lockstep_groups(models, for_grad=True) if (stacked is not False and method in _TORCH_DEFAULT_LR and method != "LBFGS") else []
    if stacked is None:
        groups = [(key, g) for key, g in groups if key[1][0] < STACKED_MAX_N]      # key[1] = X's shape
    for key, g in groups:
        ms = [models[i] for i in g]
        B = len(ms)
        plists = _group_param_lists(ms)
        transforms = [_shared_transform(pl) for pl in plists]
        flags = [{bool(p.requires_grad) for p in pl} for pl in plists]
        mean_trainable = any(p.requires_grad for m in ms for p in m.mean_function.parameters())
        if any(t is None for t in transforms) or any(len(f) != 1 for f in flags) or mean_trainable:
            continue
        X, R, _ = _group_data(ms)
        raws = [torch.nn.Parameter(torch.stack([p.data for p in pl]), requires_grad=f.pop()) for pl, f in zip(plists, flags)]
        trainable = [r for r in raws if r.requires_grad]
        if not trainable:
            continue                     # nothing to optimise in lock step: each model's own optimize() reports as the reference does
        holder = {}
        dev_losses = torch.empty(max_iter, B, dtype=torch.float64, device=X.device)
        print("multi_start_optimize: %d x %s in lock step via %s" % (B, ms[0].__class__.__name__, method))

        def step(optimizer, idx, set_to_none=False):
            optimizer.zero_grad(set_to_none=set_to_none)
            var, ls, nz = (t(r) for t, r in zip(transforms, raws))
            lml = _ops.BatchedGPRLogLik.apply(X, R, var.reshape(B), ls.reshape(B, -1), nz.reshape(B), key[0], holder)
            loss = -(lml + 0.0)
            loss.sum().backward()
            optimizer.step()
            if idx is not None:
                dev_losses[idx] = loss.detach()
            return loss
        if capture:
            optimizer = _captured_lockstep_loop(ms[0], method, trainable, learning_rate, step, dev_losses, max_iter)
            if verbose:                                # (the replays print nothing: the lines of the ordinary loop, afterwards)
                for idx, row in enumerate(dev_losses.tolist()):
                    print("Iter: %d\tLoss: %s" % (idx, row))
        else:
            optimizer = ms[0]._make_optimizer(method, trainable, learning_rate)
            for idx in range(max_iter):
                step(optimizer, idx)
                if verbose:
                    print("Iter: %d\tLoss: %s" % (idx, dev_losses[idx].tolist()))
        losses[g, :] = dev_losses.t().cpu().numpy()
        with torch.no_grad():
            for pl, r in zip(plists, raws):
                for b, p in enumerate(pl):
                    p.data = r.data[b].clone()
        for i in g:
            done[i] = True
    rest = [i for i in range(len(models)) if not done[i]]
    rest_models = [models[i] for i in rest]
    if method in _TORCH_DEFAULT_LR and method != "LBFGS" and len(rest) >= 2 and \
            (_lockstep_groups(rest_models) or _expression_groups(rest_models) or _vfe_groups(rest_models)):
        # What cannot share a stacked parameter tensor (composite kernels, priors, trainable mean functions, mixed frozen
        # parameters) still shares the EVALUATION: every model keeps its own optimiser over its own parameters -- exactly the
        # objects and tensor layouts of its own optimize(), so its trajectory is bit-identical -- and each iteration is one
        # batched_loss_and_grad over all of them (base.py:260-269: zero_grad, loss, backward, step).
        opts = []
        for m in rest_models:
            m._auto_place()
        plist = [p for m in rest_models for p in m.parameters() if p.requires_grad]
        if len({id(p) for p in plist}) == len(plist):
            # ONE optimiser object over every model's own parameter tensors: the torch optimisers are elementwise per tensor and
            # their multi-tensor kernels treat every tensor of the list by itself, so each model's update is what its own
            # optimiser would do -- bit for bit -- at one step() call per iteration instead of one per model
            shared = rest_models[0]._make_optimizer(method, plist, learning_rate)
            for m in rest_models:
                m.optimizer = shared
            opts.append(shared)
        else:                                   # models that share Param objects: every model's own optimiser, as optimize() would
            for m in rest_models:
                m.optimizer = m._make_optimizer(method, [p for p in m.parameters() if p.requires_grad], learning_rate)
                opts.append(m.optimizer)
        print("multi_start_optimize: %d models, one lock-step evaluation per iteration, via %s" % (len(rest), method))
        plan = _plan_groups(rest_models)              # (shapes, kernels and priors are fixed while the search runs)
        for idx in range(max_iter):
            for o in opts:
                o.zero_grad()
            out = batched_loss_and_grad(rest_models, _plan=plan)
            for o in opts:
                o.step()
            vals = torch.cat([l.reshape(-1) for l in out]).cpu().numpy()
            losses[rest, idx] = vals
            if verbose:
                print("Iter: %d\tLoss: %s" % (idx, vals.tolist()))
        for i in rest:
            done[i] = True
    for i, m in enumerate(models):
        if not done[i]:
            res = m.optimize(method=method, max_iter=max_iter, verbose=verbose, learning_rate=learning_rate)
            if isinstance(res, tuple):
                losses[i, :len(res[0])] = res[0]
    release_batch_buffers()          # the search is over: its lock-step buffers (B factors + backward workspaces) go back to the allocator
    return losses, time.time() - tic


def _batched_on_streams(models, streams):
    """whole evaluations placed on the caller's streams (see batched_log_likelihood)."""
    dev = models[0].X.device
    cur = torch.cuda.current_stream(dev)
    side = any(st is not cur for st in streams)
    if side:
        # host-side fork/join: event waits between a created stream and the legacy default stream
        # cost ~7 ms per evaluation on this runtime (tools/stream_kind_test.py waits), a host sync of an
        # idle stream costs nothing
        cur.synchronize()
    pending = []
    with torch.no_grad():
        for m, st in zip(models, streams):
            k = m._stationary()
            if k is None or m.X.shape[0] >= _ops.refine_min_n():
                # dense-K / composite kernels, and sizes at which log_likelihood() refines the quadratic form
                # (the value must not depend on which entry point computed it): sequential path
                pending.append(None)
                continue
            with torch.cuda.stream(st):
                resid = m.Y - m.mean_function(m.X)
                f = _ops.kernel_factor_async(k._kind, m.X, k.variance.transform(), k.length_scales.transform(),
                                             m.likelihood.variance.transform(), R=resid,
                                             factor=m._holder.get("factor"))
                m._holder["factor"] = f
                pending.append((f, f.lml_terms(), st))
        out = []
        for m, p in zip(models, pending):
            ok = False
            if p is not None:
                with torch.cuda.stream(p[2]):
                    ok = int(p[0].info.item()) == 0        # synchronises that model's stream
            out.append(p[1][2:3] if ok else m.log_likelihood())
    return out


def two_lane_streams(models):
    """the round-3 default placement of batched_log_likelihood(streams=...): the models alternate between two internal streams."""
    dev = models[0].X.device
    if dev not in _LANES:
        _LANES[dev] = [torch.cuda.Stream(device=dev), torch.cuda.Stream(device=dev)]
    return [_LANES[dev][i % 2] for i in range(len(models))]

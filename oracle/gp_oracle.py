"""
TEST INFRASTRUCTURE ONLY -- CPU oracle for the exact-GP hot path.

This file is a CPU restatement of the *reference's* algorithm (cics-nd/gptorch
v0.3.2) for the path  K(X,X) assembly -> +sigma_n^2 I -> Cholesky -> triangular
solve -> log-det -> LML (+ autograd backward) and the predict variant.  It is
the checker for the HIP path; it is never imported by the product package
`gptorch_amd` -- only `tests/`, `__graft_entry__.smoke()` and the
`cpu_baseline` leg of `bench.py` may use it.

Why torch-CPU ops and not plain C/numpy: the reference itself is pure Python
over PyTorch ops; its arithmetic lives in a third-party dependency that is NOT
under /root/reference -- PyTorch (unpinned: `requirements.txt:13` "torch>=1",
`setup.py:21`), resolving in this image to torch 2.10.0 (ATen CPU -> oneMKL
dgemm/dpotrf/dtrsm).  Restating the same op sequence against the same ATen
kernels makes this oracle *the reference path itself* up to Python glue, and
lets autograd supply the reference's backward.  An independent plain-C
restatement (unblocked textbook algorithms) lives beside it in `gp_oracle.c`.

Parity pin: every function here is checked against the imported reference
(`tests/golden/make_golden.py`, run in the build container where
/root/reference exists) and against the reference's own golden fixtures
(`test/data/kernels/*.npy` values, `test/test_util.py:38-106`), and the
resulting vectors are committed under `tests/golden/`.  GPR.log_likelihood,
its gradients and GPR._predict are *unpinned by the reference's own tests*
(`test/test_models/test_gpr.py` checks shapes only) -- those goldens come from
running the reference itself here.

All citations are relative to /root/reference/.
"""
import math

import numpy as np
import torch

DTYPE = torch.float64  # gptorch/util.py:11-12 (TensorType = DoubleTensor)


# ----------------------------------------------------------------------------
# L1 numerical primitives
# ----------------------------------------------------------------------------
def squared_distance(x1, x2=None):
    """gptorch/util.py:73-88 -- Gram-trick pairwise squared distance; negatives
    clamped in value but not in gradient (the .detach() at util.py:88)."""
    if x2 is None:
        return squared_distance(x1, x1)
    x1s = x1.pow(2).sum(1, keepdim=True)
    x2s = x2.pow(2).sum(1, keepdim=True)
    r2 = x1s + x2s.t() - 2.0 * x1 @ x2.t()
    return r2 - (torch.clamp(r2, max=0.0)).detach()


def jit_op(op, x, max_tries=10):
    """gptorch/functions.py:20-43 -- try plain, then absolute jitter
    10^(-10+i) I for i = 0..9, then RuntimeError("Max tries exceeded.").
    Returns (result, rung) with rung = -1 for the un-jittered success."""
    try:
        return op(x), -1
    except Exception:
        pass
    for i in range(max_tries):
        try:
            this_jitter = 10.0 ** (-max_tries + i) * torch.eye(*x.shape, dtype=x.dtype)
            return op(x + this_jitter), i
        except RuntimeError:
            pass
    raise RuntimeError("Max tries exceeded.")


def cholesky(x):
    """gptorch/functions.py:46-47 (torch.cholesky == torch.linalg.cholesky, lower)."""
    return jit_op(torch.linalg.cholesky, x)[0]


def cholesky_rung(x):
    return jit_op(torch.linalg.cholesky, x)


def trtrs(b, a, lower=True):
    """gptorch/functions.py:71-76 -- solve a x = b, triangular a."""
    return torch.linalg.solve_triangular(a, b, upper=not lower)


def lt_log_determinant(L):
    """gptorch/functions.py:61-68."""
    return L.diag().log().sum()


# ----------------------------------------------------------------------------
# L2 kernels (stationary family; hyper-parameters given in constrained space)
# ----------------------------------------------------------------------------
def scaled_squared_dist(X, X2, length_scales):
    """gptorch/kernels.py:149-159."""
    if X2 is None:
        return squared_distance(X / length_scales)
    return squared_distance(X / length_scales, X2 / length_scales)


def scaled_dist(X, X2, length_scales):
    """gptorch/kernels.py:161-172 -- sqrt(clamp(r2, min=1e-40))."""
    return torch.sqrt(torch.clamp(scaled_squared_dist(X, X2, length_scales), min=1e-40))


def kernel_K(kind, X, X2, variance, length_scales):
    """Rbf: kernels.py:215-222; Matern52: 204-212; Matern32: 196-201;
    Exp/Matern12: 182-193; Periodic: 228-235."""
    if kind == "Rbf":
        r2 = scaled_squared_dist(X, X2, length_scales)
        return variance * torch.exp(-r2 / 2.0)
    r = scaled_dist(X, X2, length_scales)
    if kind == "Matern52":
        s5 = torch.tensor([math.sqrt(5.0)], dtype=DTYPE).to(r.device)   # kernels.py:207
        return variance * (1.0 + s5 * r + 5.0 / 3.0 * r * r) * torch.exp(-s5 * r)
    if kind == "Matern32":
        r3 = torch.tensor([math.sqrt(3.0)], dtype=DTYPE).to(r.device) * r   # kernels.py:199-200
        return variance * (1.0 + r3) * torch.exp(-r3)
    if kind in ("Exp", "Matern12"):
        return variance * torch.exp(-r)
    if kind == "Periodic":
        return variance * torch.cos(r)                                   # kernels.py:228-235
    raise ValueError(kind)


def linear_K(X, X2, variance):
    """kernels.py:258-262 (variance: one per input dimension)."""
    return torch.mm(X * variance, (X if X2 is None else X2).t())


def dense_lml(Kyy, resid):
    """gpr.py:61-67 on a given Kyy = K(x) + sigma_n^2 I (any kernel)."""
    L = cholesky(Kyy)
    alpha = trtrs(resid, L)
    n, dy = resid.shape
    return -0.5 * alpha.pow(2).sum() - dy * lt_log_determinant(L) - 0.5 * dy * n * math.log(2.0 * math.pi)


def dense_predict(Kyy, Ksx, Kss_or_diag, resid, diag=True):
    """gpr.py:102-115 for given Kyy, K(x, x*) and K(x*) (or its diagonal)."""
    L = cholesky(Kyy)
    A = trtrs(Ksx, L)
    V = trtrs(resid, L)
    mean = A.t() @ V
    if diag:
        return mean, (Kss_or_diag - A.pow(2).sum(0))[:, None].expand_as(mean)
    return mean, Kss_or_diag - A.t() @ A


def linear_Kdiag(X, variance):
    """kernels.py:264-265."""
    return torch.sum(X * X * variance, 1)


def kernel_Kdiag(X, variance):
    """gptorch/kernels.py:174-179."""
    return variance.expand(X.size(0))


# ----------------------------------------------------------------------------
# L3 GPR (gptorch/models/gpr.py)
# ----------------------------------------------------------------------------
class GPROracle:
    """Exact GP regression with a stationary kernel, Gaussian likelihood and a
    constant mean, parameterised like the reference: raw = log(value)
    (param.py:13-50 with settings.py:7 ExpTransform)."""

    def __init__(self, x, y, kind="Rbf", variance=1.0, length_scales=1.0,
                 noise=None, ARD=False, mean=None):
        self.X = torch.as_tensor(np.asarray(x), dtype=DTYPE)
        self.Y = torch.as_tensor(np.asarray(y), dtype=DTYPE)
        self.kind = kind
        d = self.X.shape[1]
        ls = np.asarray(length_scales, dtype=np.float64) * (np.ones(d) if ARD else np.ones(1))
        if noise is None:  # models/base.py:101-109 (numpy var -> ddof=0)
            noise = 0.001 * np.asarray(y).var()
        self.raw_variance = torch.tensor([math.log(variance)], dtype=DTYPE, requires_grad=True)
        self.raw_length_scales = torch.tensor(np.log(ls), dtype=DTYPE, requires_grad=True)
        self.raw_noise = torch.tensor([math.log(noise)], dtype=DTYPE, requires_grad=True)
        dy = self.Y.shape[1]
        self.mean_val = torch.zeros(dy, dtype=DTYPE) if mean is None else \
            torch.as_tensor(np.asarray(mean), dtype=DTYPE)

    def parameters(self):
        return [self.raw_variance, self.raw_length_scales, self.raw_noise]

    # -- pieces -------------------------------------------------------------
    def _mean(self, x):
        """mean_functions.py:28-32."""
        return torch.zeros(x.shape[0], self.Y.shape[1], dtype=DTYPE) + self.mean_val

    def K(self, X, X2=None):
        return kernel_K(self.kind, X, X2, self.raw_variance.exp(), self.raw_length_scales.exp())

    def compute_kyy(self, x=None):
        """gpr.py:69-86."""
        x = self.X if x is None else x
        n = x.shape[0]
        return self.K(x) + self.raw_noise.exp().expand(n, n).diag().diag()

    def log_likelihood(self, x=None, y=None):
        """gpr.py:47-67 -- GPML Alg. 2.1; result shape (1,)."""
        x = self.X if x is None else x
        y = self.Y if y is None else y
        if not x.shape[0] == y.shape[0]:
            raise ValueError("X and Y must have same # data.")
        num_input, dim_output = y.shape
        L = cholesky(self.compute_kyy(x))
        alpha = trtrs(y - self._mean(x), L)
        const = torch.tensor([-0.5 * dim_output * num_input * np.log(2 * np.pi)], dtype=DTYPE)
        return -0.5 * alpha.pow(2).sum() - dim_output * lt_log_determinant(L) + const

    def loss(self):
        """models/base.py:418-419 with no priors (model.py:158-177 -> 0.0)."""
        return -(self.log_likelihood() + 0.0)

    def predict_f(self, x_new, diag=True):
        """gpr.py:88-117."""
        x_new = torch.as_tensor(np.asarray(x_new), dtype=DTYPE)
        x = self.X
        k_ys = self.K(x, x_new)
        L = cholesky(self.compute_kyy(x))
        A = trtrs(k_ys, L)
        V = trtrs(self.Y - self._mean(x), L)
        mean_f = A.t() @ V + self._mean(x_new)
        if diag:
            var_f = (kernel_Kdiag(x_new, self.raw_variance.exp()) - (A * A).sum(0))[:, None].expand_as(mean_f)
        else:
            var_f = self.K(x_new) - A.t() @ A
        return mean_f, var_f

    def predict_y(self, x_new, diag=True):
        """models/base.py:348-360 + likelihoods.py:106-123."""
        mean_f, cov_f = self.predict_f(x_new, diag=diag)
        s = self.raw_noise.exp()
        if diag:
            return mean_f, cov_f + s.expand_as(cov_f)
        return mean_f, cov_f + s.expand_as(cov_f).diag().diag()

    def loss_and_grads(self):
        for p in self.parameters():
            p.grad = None
        loss = self.loss()
        loss.backward()
        return loss.detach().clone(), [p.grad.detach().clone() for p in self.parameters()]

    def optimize_adam(self, max_iter=50, learning_rate=0.01):
        """models/base.py:149-151, 260-269 -- Adam(lr=0.01) on the raw (log) params."""
        opt = torch.optim.Adam(self.parameters(), lr=learning_rate)
        losses = np.zeros(max_iter)
        for idx in range(max_iter):
            opt.zero_grad()
            loss = self.loss()
            loss.backward()
            opt.step()
            losses[idx] = loss.item()
        return losses


# ----------------------------------------------------------------------------
# Closed-form gradient -- used to cross-check the HIP
# backward independently of autograd.
# ----------------------------------------------------------------------------
def lml_closed_form_grads(kind, X, Y, variance, length_scales, noise, mean_val=None):
    """Returns (lml, dLML/dlog variance, dLML/dlog length_scales[D or 1], dLML/dlog noise)
    from  G = 1/2 (a a^T - dy K^-1),  a = Kyy^-1 (y - m)."""
    X = torch.as_tensor(X, dtype=DTYPE)
    Y = torch.as_tensor(Y, dtype=DTYPE)
    n, dy = Y.shape
    v = torch.tensor([variance], dtype=DTYPE)
    ls = torch.as_tensor(np.atleast_1d(length_scales), dtype=DTYPE)
    Kf = kernel_K(kind, X, None, v, ls)
    Kyy = Kf + noise * torch.eye(n, dtype=DTYPE)
    L = torch.linalg.cholesky(Kyy)
    R = Y if mean_val is None else Y - torch.as_tensor(mean_val, dtype=DTYPE)
    alpha = torch.linalg.solve_triangular(L, R, upper=False)
    lml = -0.5 * alpha.pow(2).sum() - dy * L.diag().log().sum() - 0.5 * dy * n * math.log(2 * math.pi)
    a = torch.cholesky_solve(R, L)
    Kinv = torch.cholesky_inverse(L)
    G = 0.5 * (a @ a.t() - dy * Kinv)
    g_var = (G * Kf).sum()
    g_noise = noise * G.diag().sum()
    Xs = X / ls
    if kind == "Rbf":
        base = Kf
    elif kind == "Matern52":
        r = torch.sqrt(torch.clamp(squared_distance(Xs), min=1e-40))
        s5 = math.sqrt(5.0)
        base = variance * (5.0 / 3.0) * (1.0 + s5 * r) * torch.exp(-s5 * r)
    else:
        raise ValueError(kind)
    GB = G * base
    g_ls = []
    for d in range(X.shape[1]):
        diff = Xs[:, d:d + 1] - Xs[:, d:d + 1].t()
        g_ls.append((GB * diff * diff).sum())
    g_ls = torch.stack(g_ls)
    if ls.numel() == 1:
        g_ls = g_ls.sum().reshape(1)
    return lml, g_var, g_ls, g_noise


# ----------------------------------------------------------------------------
# VFE (gptorch/models/sparse_gpr.py:92-195) -- BASELINE config 5, SURVEY 8(f)-1
# ----------------------------------------------------------------------------
class VFEOracle:
    """Titsias' collapsed bound exactly as sparse_gpr.py:108-153 evaluates it (zero mean;
    note the reference's quirk `err = self.Y`, sparse_gpr.py:125)."""

    def __init__(self, x, y, z, kind="Matern32", variance=1.0, length_scales=1.0, noise=1.0):
        self.X = torch.as_tensor(np.asarray(x), dtype=DTYPE)
        self.Y = torch.as_tensor(np.asarray(y), dtype=DTYPE)
        self.Z = torch.as_tensor(np.asarray(z), dtype=DTYPE)
        self.kind = kind
        self.variance = torch.tensor([float(variance)], dtype=DTYPE)
        self.ls = torch.as_tensor(np.atleast_1d(np.asarray(length_scales, dtype=np.float64)))
        self.noise = torch.tensor([float(noise)], dtype=DTYPE)

    def K(self, a, b=None):
        return kernel_K(self.kind, a, b, self.variance, self.ls)

    def _common(self, x):
        m = self.Z.shape[0]
        Kuf = self.K(self.Z, x)
        L = cholesky(self.K(self.Z))
        A = trtrs(Kuf, L)
        AAT = A @ A.t() / self.noise
        B = AAT + torch.eye(m, dtype=DTYPE)
        LB = cholesky(B)
        c = trtrs(A @ self.Y, LB) / self.noise
        return L, A, AAT, LB, c

    def log_likelihood(self):
        """sparse_gpr.py:108-153."""
        x = self.X
        n, d_out = self.Y.shape
        L, A, AAT, LB, c = self._common(x)
        elbo = -0.5 * d_out * n * math.log(2 * math.pi)
        elbo = elbo - d_out * LB.diag().log().sum()
        elbo = elbo - 0.5 * d_out * n * self.noise.log()
        elbo = elbo - 0.5 * (self.Y.pow(2).sum() + d_out * kernel_Kdiag(x, self.variance).sum()) / self.noise
        elbo = elbo + 0.5 * c.pow(2).sum()
        elbo = elbo + 0.5 * d_out * AAT.diag().sum()
        return elbo[0]

    def predict_f(self, x_new, diag=True):
        """sparse_gpr.py:155-195."""
        x_new = torch.as_tensor(np.asarray(x_new), dtype=DTYPE)
        L, A, AAT, LB, c = self._common(self.X)
        Kus = self.K(self.Z, x_new)
        tmp1 = trtrs(Kus, L)
        tmp2 = trtrs(tmp1, LB)
        mean = tmp2.t() @ c
        if diag:
            var = (kernel_Kdiag(x_new, self.variance) - tmp1.pow(2).sum(0).squeeze()
                   + tmp2.pow(2).sum(0).squeeze())[:, None].expand_as(mean)
        else:
            var = self.K(x_new) + tmp2.t() @ tmp2 - tmp1.t() @ tmp1
        return mean, var


def vfe_grads_autograd(o):
    """d(elbo)/d(variance, length_scales, noise, Z) of a VFEOracle by autograd through the
    reference's op chain (what `loss().backward()` gives the reference, up to the sign and the
    chain through the log-parameterisation)."""
    leaves = [o.variance, o.ls, o.noise, o.Z]
    for t in leaves:
        t.requires_grad_(True)
        t.grad = None
    o.log_likelihood().backward()
    out = [t.grad.clone() for t in leaves]
    for t in leaves:
        t.requires_grad_(False)
        t.grad = None
    return out


def vfe_closed_form_grads(o):
    """The closed form the native backward implements (gptorch_amd/models/sparse_gpr.py):
    dF/dKuu, dF/dKuf as explicit M x M / M x N matrices, then the kernel chain rule by
    autograd on K alone.  Returns the same four tensors as vfe_grads_autograd."""
    with torch.no_grad():
        m, (n, p) = o.Z.shape[0], o.Y.shape
        s = o.noise.item()
        Kuu, Kuf = o.K(o.Z), o.K(o.Z, o.X)
        L = cholesky(Kuu)
        A = trtrs(Kuf, L)
        eye = torch.eye(m, dtype=DTYPE)
        B = A @ A.t() / s + eye
        LB = cholesky(B)
        v = A @ o.Y
        c = trtrs(v, LB) / s
        beta = torch.linalg.solve_triangular(LB.t(), c, upper=True)
        Binv = torch.cholesky_inverse(LB)
        Li = torch.linalg.solve_triangular(L, eye, upper=False)
        gamma = Li.t() @ beta
        bbT = beta @ beta.t()
        Guu = Li.t() @ (0.5 * p * (2 * eye - Binv - B) - 0.5 * bbT) @ Li
        P = Li.t() @ (p * (eye - Binv) - bbT) @ Li
        Guf = (P @ Kuf + gamma @ o.Y.t()) / s
        g_noise = (0.5 * p / s) * (m - Binv.diagonal().sum()) - c.pow(2).sum() / s \
            + 0.5 * (beta * ((B - eye) @ beta)).sum() / s - 0.5 * p * (B - eye).diagonal().sum() / s \
            - 0.5 * p * n / s + 0.5 * (o.Y.pow(2).sum() + p * n * o.variance[0]) / (s * s)
    leaves = [o.variance, o.ls, o.Z]
    for t in leaves:
        t.requires_grad_(True)
        t.grad = None
    ((Guu * o.K(o.Z)).sum() + (Guf * o.K(o.Z, o.X)).sum() - 0.5 * p * n / s * o.variance.sum()).backward()
    g_var, g_ls, g_Z = [t.grad.clone() for t in leaves]
    for t in leaves:
        t.requires_grad_(False)
        t.grad = None
    return [g_var, g_ls, g_noise.reshape(1), g_Z]

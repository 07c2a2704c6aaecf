import os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np, torch
from gptorch_amd import kernels, likelihoods, rng
from gptorch_amd.models import GPR
from oracle import gp_oracle as orc
for (n, d, dy, kind, ard) in [(600, 3, 200, "Rbf", False), (700, 64, 2, "Matern52", True), (1300, 40, 1, "Rbf", True), (300, 70, 1, "Rbf", False), (400, 100, 2, "Matern52", True)]:
    x, y = rng.make_regression(n, d, dy, seed=4)
    ls = (0.7 * np.sqrt(d) * (0.5 + rng.uniform(9, d))) if ard else 0.7 * np.sqrt(d)
    k = getattr(kernels, kind)(d, variance=1.2, length_scales=ls, ARD=ard)
    m = GPR(x, y, k, likelihood=likelihoods.Gaussian(variance=0.05)); m.cuda()
    o = orc.GPROracle(x, y, kind=kind, variance=1.2, length_scales=ls, noise=0.05, ARD=ard)
    lo = o.loss(); lo.backward()
    try:
        l = m.loss(); l.backward()
    except Exception as e:
        print(n, d, dy, kind, "forward ok?" , "EXC", type(e).__name__, str(e)[:80]); continue
    gv = abs(m.kernel.variance.grad.item() - o.raw_variance.grad.item()) / abs(o.raw_variance.grad.item())
    gl = (m.kernel.length_scales.grad.cpu() - o.raw_length_scales.grad).abs().max().item() / o.raw_length_scales.grad.abs().max().item()
    print(n, d, dy, kind, "loss rel err %.2e  g_var %.2e  g_ls %.2e" % (abs(l.item() - lo.item()) / abs(lo.item()), gv, gl))
    xs = rng.normal(77, (33, d))
    mu, var = m.predict_f(xs); omu, ovar = o.predict_f(xs)
    print("   predict mean %.2e var %.2e" % (np.abs(mu - omu.detach().numpy()).max(), np.abs(var - ovar.detach().numpy()).max()))

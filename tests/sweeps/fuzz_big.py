import os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np, torch
from gptorch_amd import kernels, likelihoods, rng
from gptorch_amd.models import GPR
from oracle import gp_oracle as orc
torch.set_num_threads(64)
rs = np.random.RandomState(3)
worst = [0, 0, 0]
for it, n in enumerate([3071, 3072, 3073, 4097, 5000, 6143, 7169, 8191, 8193, 9215]):
    d = int(rs.choice([3, 8, 16])); dy = int(rs.choice([1, 2])); kind = str(rs.choice(["Rbf", "Matern52"]))
    x, y = rng.make_regression(n, d, dy, seed=7000 + it)
    ls = float(np.sqrt(d) * (0.6 + rs.rand()))
    m = GPR(x, y, getattr(kernels, kind)(d, variance=1.2, length_scales=ls), likelihood=likelihoods.Gaussian(variance=0.02)); m.cuda()
    o = orc.GPROracle(x, y, kind=kind, variance=1.2, length_scales=ls, noise=0.02)
    lo = o.loss(); lo.backward()
    l = m.loss(); l.backward()
    e_l = abs(l.item() - lo.item()) / abs(lo.item())
    e_g = max((a.cpu() - b).abs().max().item() / max(1.0, b.abs().max().item()) for a, b in
              [(m.kernel.variance.grad, o.raw_variance.grad), (m.kernel.length_scales.grad, o.raw_length_scales.grad), (m.likelihood.variance.grad, o.raw_noise.grad)])
    xs = rng.normal(1, (9, d)); mu, var = m.predict_f(xs)
    with torch.no_grad(): omu, ovar = o.predict_f(xs)
    e_p = max(np.abs(mu - omu.numpy()).max(), np.abs(var - ovar.numpy()).max())
    worst = [max(worst[0], e_l), max(worst[1], e_g), max(worst[2], e_p)]
    print(n, d, dy, kind, "lml %.2e grad %.2e pred %.2e" % (e_l, e_g, e_p), flush=True)
print("worst", worst)

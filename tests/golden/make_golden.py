#!/usr/bin/env python3
"""
Generate the golden fixtures under tests/golden/ by RUNNING THE REFERENCE
(cics-nd/gptorch v0.3.2, imported read-only from /root/reference) in the build
container, and check the CPU oracle (oracle/gp_oracle.py) against it.

    python tests/golden/make_golden.py            # everything up to N=8192
    python tests/golden/make_golden.py --big      # also the N=32768 forward (needs ~40 GB, minutes)

The reference never travels to the GPU box; only the vectors written here do.
Inputs come from gptorch_amd.rng (deterministic, regenerated bit-for-bit from
(seed, N, D)), so big cases store scalars + a checksum of the inputs only.
"""
import argparse
import contextlib
import io
import json
import os
import sys
import time

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.abspath(os.path.join(HERE, "..", ".."))
REF = "/root/reference"
sys.path.insert(0, ROOT)
sys.path.insert(0, REF)

import gptorch  # noqa: E402  (the reference)
from gptorch import kernels as rk  # noqa: E402
from gptorch.models import GPR as RefGPR  # noqa: E402
from gptorch import likelihoods as rl  # noqa: E402
from gptorch import functions as rf  # noqa: E402
from gptorch import util as ru  # noqa: E402
from gptorch import mean_functions as rm  # noqa: E402

from gptorch_amd import rng  # noqa: E402
from oracle import gp_oracle as orc  # noqa: E402

assert gptorch.__file__.startswith(REF), gptorch.__file__

KERNELS = {"Rbf": rk.Rbf, "Matern52": rk.Matern52, "Matern32": rk.Matern32, "Exp": rk.Exp, "Periodic": rk.Periodic}
STATIC_KERNELS = ["White", "Constant", "Bias", "Linear", "Matern12"]   # fixtures only (no ARD set)


def quiet():
    return contextlib.redirect_stdout(io.StringIO())


def ref_model(case, x, y):
    d = x.shape[1]
    ls = case["length_scales"]
    if case["ARD"]:
        ls = np.asarray(ls, dtype=np.float64) * np.ones(d)
    kern = KERNELS[case["kind"]](d, variance=case["variance"], length_scales=ls, ARD=case["ARD"])
    lik = rl.Gaussian(variance=case["noise"])
    mean = None
    if case.get("mean") is not None:
        mean = rm.Constant(y.shape[1], val=torch.tensor(case["mean"], dtype=torch.float64))
        mean.val.requires_grad_(False)
    return RefGPR(x, y, kern, likelihood=lik, mean_function=mean)


def oracle_model(case, x, y):
    return orc.GPROracle(x, y, kind=case["kind"], variance=case["variance"],
                         length_scales=case["length_scales"], noise=case["noise"],
                         ARD=case["ARD"], mean=case.get("mean"))


def rel(a, b):
    a, b = np.asarray(a, dtype=np.float64), np.asarray(b, dtype=np.float64)
    return float(np.max(np.abs(a - b) / np.maximum(1.0, np.abs(b))))


# ----------------------------------------------------------------------------
def gen_ref_kernel_fixtures(out):
    """Copy the VALUES of the reference's own kernel fixtures (test/data/kernels)
    for the stationary kernels on the path, and check reference + oracle on them
    exactly as test/test_kernels.py:59-127 does."""
    ddir = os.path.join(REF, "test", "data", "kernels")
    pack = {}
    for name in ["x1", "x2", "ard_length_scales"]:
        pack[name] = np.load(os.path.join(ddir, name + ".npy"))
    for k in KERNELS:
        for suffix in ["kx", "kx2", "kdiag", "kx_ard", "kx2_ard", "kdiag_ard"]:
            pack[f"{k}_{suffix}"] = np.load(os.path.join(ddir, f"{k}_{suffix}.npy"))
    for k in STATIC_KERNELS:
        for suffix in ["kx", "kx2", "kdiag"]:
            pack[f"{k}_{suffix}"] = np.load(os.path.join(ddir, f"{k}_{suffix}.npy"))
    x1, x2 = torch.tensor(pack["x1"]), torch.tensor(pack["x2"])
    one = torch.ones(1, dtype=torch.float64)
    ard = torch.tensor(pack["ard_length_scales"])
    assert np.allclose(orc.linear_K(x1, None, torch.ones(3, dtype=torch.float64)).numpy(), pack["Linear_kx"])
    assert np.allclose(orc.linear_K(x1, x2, torch.ones(3, dtype=torch.float64)).numpy(), pack["Linear_kx2"])
    assert np.allclose(orc.linear_Kdiag(x1, torch.ones(3, dtype=torch.float64)).numpy(), pack["Linear_kdiag"])
    for k in KERNELS:
        assert np.allclose(orc.kernel_K(k, x1, None, one, one).numpy(), pack[f"{k}_kx"])
        assert np.allclose(orc.kernel_K(k, x1, x2, one, one).numpy(), pack[f"{k}_kx2"])
        assert np.allclose(orc.kernel_K(k, x1, None, one, ard).numpy(), pack[f"{k}_kx_ard"])
        assert np.allclose(orc.kernel_K(k, x1, x2, one, ard).numpy(), pack[f"{k}_kx2_ard"])
        assert np.allclose(orc.kernel_Kdiag(x1, one).numpy(), pack[f"{k}_kdiag"])
    np.savez(os.path.join(out, "ref_kernel_fixtures.npz"), **pack)
    print("ref kernel fixtures: oracle matches", len(pack), "arrays")


def gen_kernel_cases(out):
    """K(X), K(X,X2), Kdiag from the reference on rng inputs: full matrices at
    small sizes, 64 sampled entries + Frobenius norm + trace at C1/C2 sizes."""
    cases = []
    small = {}
    rs = np.random.RandomState(7)
    for kind in ["Rbf", "Matern52", "Matern32", "Exp", "Periodic"]:
        for (n, m, d, ard) in [(33, 17, 3, False), (70, 129, 5, True), (128, 64, 8, True), (1, 1, 1, False), (200, 1, 2, False)]:
            x = rng.normal(100 + n, (n, d))
            x2 = rng.normal(200 + m, (m, d))
            ls = (0.5 + rng.uniform(300 + d, d)) if ard else np.array([0.8])
            var = 1.7
            kern = KERNELS[kind](d, variance=var, length_scales=ls if ard else float(ls[0]), ARD=ard)
            tx, tx2 = torch.tensor(x), torch.tensor(x2)
            with torch.no_grad():
                kx = kern.K(tx).numpy()
                kx2 = kern.K(tx, tx2).numpy()
                kd = kern.Kdiag(tx).numpy()
                okx = orc.kernel_K(kind, tx, None, torch.tensor([var], dtype=torch.float64), torch.tensor(ls)).numpy()
            assert np.max(np.abs(kx - okx)) < 1e-14
            key = f"{kind}_{n}_{m}_{d}_{int(ard)}"
            small[key + "_kx"], small[key + "_kx2"], small[key + "_kdiag"] = kx, kx2, kd
            cases.append(dict(key=key, kind=kind, n=n, m=m, d=d, ARD=ard, variance=var,
                              length_scales=ls.tolist(), seed_x=100 + n, seed_x2=200 + m))
    for (n, m, d) in [(33, 17, 3), (70, 129, 20)]:
        x, x2 = rng.normal(100 + n, (n, d)), rng.normal(200 + m, (m, d))
        v = 0.5 + rng.uniform(400 + d, d)
        kern = rk.Linear(d, variance=v)
        with torch.no_grad():
            small[f"Linear_{n}_{m}_{d}_kx"] = kern.K(torch.tensor(x)).numpy()
            small[f"Linear_{n}_{m}_{d}_kx2"] = kern.K(torch.tensor(x), torch.tensor(x2)).numpy()
            small[f"Linear_{n}_{m}_{d}_kdiag"] = kern.Kdiag(torch.tensor(x)).numpy()
    np.savez_compressed(os.path.join(out, "kernel_small.npz"), **small)
    sampled = []
    for (kind, n, d, ls) in [("Rbf", 512, 2, 1.0), ("Rbf", 8192, 8, np.sqrt(8.0)), ("Matern52", 4096, 16, 4.0)]:
        x = rng.normal(0, (n, d))
        kern = KERNELS[kind](d, variance=1.0, length_scales=float(ls))
        with torch.no_grad():
            kx = kern.K(torch.tensor(x)).numpy()
        ii = rs.randint(0, n, 64)
        jj = rs.randint(0, n, 64)
        ii[:4] = jj[:4]  # a few diagonal entries
        sampled.append(dict(kind=kind, n=n, d=d, length_scales=float(ls), variance=1.0, seed_x=0,
                            x_checksum=rng.checksum(x), i=ii.tolist(), j=jj.tolist(),
                            values=kx[ii, jj].tolist(), frobenius=float(np.linalg.norm(kx)),
                            trace=float(np.trace(kx)), sum=float(kx.sum())))
    with open(os.path.join(out, "kernel_cases.json"), "w") as f:
        json.dump(dict(small=cases, sampled=sampled), f, indent=1)
    print("kernel cases:", len(cases), "small,", len(sampled), "sampled")


LML_CASES = [
    # name, kind, n, d, dy, variance, length_scales, ARD, noise, mean
    dict(name="C1_rbf_512_2", kind="Rbf", n=512, d=2, dy=1, variance=1.0, length_scales=1.0, ARD=False, noise=1e-2),
    dict(name="rbf_300_3_dy2_ard", kind="Rbf", n=300, d=3, dy=2, variance=1.3, length_scales=[0.7, 1.1, 1.9], ARD=True, noise=5e-2),
    dict(name="m52_300_3_dy2_ard", kind="Matern52", n=300, d=3, dy=2, variance=0.8, length_scales=[0.7, 1.1, 1.9], ARD=True, noise=5e-2),
    dict(name="rbf_77_1", kind="Rbf", n=77, d=1, dy=1, variance=2.0, length_scales=0.5, ARD=False, noise=1e-3),
    dict(name="m52_130_4_mean", kind="Matern52", n=130, d=4, dy=3, variance=1.0, length_scales=2.0, ARD=False, noise=1e-1, mean=[0.3, -0.2, 1.0]),
    dict(name="rbf_1024_8", kind="Rbf", n=1024, d=8, dy=1, variance=1.0, length_scales=float(np.sqrt(8.0)), ARD=False, noise=1e-2),
    dict(name="rbf_1000_8_ls1", kind="Rbf", n=1000, d=8, dy=1, variance=1.0, length_scales=1.0, ARD=False, noise=1e-2),
    dict(name="rbf_2048_8", kind="Rbf", n=2048, d=8, dy=1, variance=1.0, length_scales=float(np.sqrt(8.0)), ARD=False, noise=1e-2),
    dict(name="rbf_4096_8_n1e-4", kind="Rbf", n=4096, d=8, dy=1, variance=1.0, length_scales=float(np.sqrt(8.0)), ARD=False, noise=1e-4),
    dict(name="m52_1024_16", kind="Matern52", n=1024, d=16, dy=1, variance=1.0, length_scales=4.0, ARD=False, noise=1e-2),
    dict(name="m52_2048_16_ard", kind="Matern52", n=2048, d=16, dy=2, variance=1.5, length_scales=(3.0 + np.arange(16) / 8.0).tolist(), ARD=True, noise=1e-2),
    dict(name="m52_4096_16", kind="Matern52", n=4096, d=16, dy=1, variance=1.0, length_scales=4.0, ARD=False, noise=1e-2),
]
C2_CASE = dict(name="C2_rbf_8192_8", kind="Rbf", n=8192, d=8, dy=1, variance=1.0, length_scales=float(np.sqrt(8.0)), ARD=False, noise=1e-2)
C3_CASE = dict(name="C3_m52_32768_16", kind="Matern52", n=32768, d=16, dy=1, variance=1.0, length_scales=4.0, ARD=False, noise=1e-2)


def gen_lml(out, big):
    res = []
    for case in LML_CASES + [C2_CASE]:
        x, y = rng.make_regression(case["n"], case["d"], case["dy"], seed=0)
        t0 = time.time()
        m = ref_model(case, x, y)
        loss = m.loss()
        assert loss.shape == (1,)
        grads_wanted = case["n"] <= 4096
        entry = dict(case)
        entry["x_checksum"], entry["y_checksum"] = rng.checksum(x), rng.checksum(y)
        entry["lml"] = float(-loss.item())
        if grads_wanted:
            loss.backward()
            entry["grad_loss"] = {
                "kernel.variance": m.kernel.variance.grad.numpy().tolist(),
                "kernel.length_scales": m.kernel.length_scales.grad.numpy().tolist(),
                "likelihood.variance": m.likelihood.variance.grad.numpy().tolist(),
            }
        t_ref = time.time() - t0
        # oracle vs reference
        o = oracle_model(case, x, y)
        if grads_wanted:
            ol, og = o.loss_and_grads()
            assert rel(ol.numpy(), loss.detach().numpy()) < 1e-12, (case["name"], ol, loss)
            for a, b in zip(og, [m.kernel.variance.grad, m.kernel.length_scales.grad, m.likelihood.variance.grad]):
                assert rel(a.numpy(), b.numpy()) < 1e-10, (case["name"], a, b)
            if case["kind"] in ("Rbf", "Matern52") and case["n"] <= 1024:
                lml, gv, gl, gn = orc.lml_closed_form_grads(case["kind"], x, y, case["variance"],
                                                            case["length_scales"], case["noise"], case.get("mean"))
                assert rel(-gv.numpy(), m.kernel.variance.grad.numpy()) < 1e-8
                assert rel(-gl.numpy(), m.kernel.length_scales.grad.numpy()) < 1e-8
                assert rel(-gn.numpy(), m.likelihood.variance.grad.numpy()) < 1e-8
        else:
            with torch.no_grad():
                ol = o.loss()
            assert rel(ol.numpy(), loss.detach().numpy()) < 1e-12
        # predictions at 16 test points (diag + full) -- gpr.py:88-117, base.py:338-360
        xs = rng.normal(4242, (16, case["d"]))
        with torch.no_grad():
            mf, vf = m.predict_f(xs)
            my, vy = m.predict_y(xs)
            mfc, cf = m.predict_f(xs, diag=False)
            myc, cy = m.predict_y(xs, diag=False)
            omf, ovf = o.predict_f(xs)
            omy, ocy = o.predict_y(xs, diag=False)
        assert rel(omf.numpy(), mf) < 1e-10 and rel(ovf.numpy(), vf) < 1e-10
        assert rel(ocy.numpy(), cy) < 1e-10
        entry["predict"] = dict(seed_xs=4242, mean_f=mf.tolist(), var_f=vf.tolist(), var_y=vy.tolist(),
                                cov_f=cf.tolist(), cov_y_diag=np.diag(cy).tolist())
        res.append(entry)
        print(f"lml {case['name']}: lml={entry['lml']:.10f} ref_time={t_ref:.2f}s")
    if big:
        case = C3_CASE
        x, y = rng.make_regression(case["n"], case["d"], case["dy"], seed=0)
        t0 = time.time()
        with torch.no_grad():
            m = ref_model(case, x, y)
            loss = m.loss()
        entry = dict(case)
        entry["x_checksum"], entry["y_checksum"] = rng.checksum(x), rng.checksum(y)
        entry["lml"] = float(-loss.item())
        entry["ref_seconds"] = time.time() - t0
        print(f"lml {case['name']}: lml={entry['lml']:.10f} ref_time={entry['ref_seconds']:.1f}s")
        with open(os.path.join(out, "lml_c3.json"), "w") as f:
            json.dump(entry, f, indent=1)
    with open(os.path.join(out, "lml_cases.json"), "w") as f:
        json.dump(res, f, indent=1)


MID_CASE = dict(name="mid_rbf_12000_8", kind="Rbf", n=12000, d=8, dy=1, variance=1.0, length_scales=float(np.sqrt(8.0)), ARD=False, noise=1e-2)


def gen_mid(out):
    """one golden on the UNREFINED side of the refinement threshold (12288 rows): the reference's LML at N = 12000
    (round-3 review: 8193 <= N < 12288 was neither refined nor covered by any golden), plus predictions at 16 points."""
    case = MID_CASE
    x, y = rng.make_regression(case["n"], case["d"], case["dy"], seed=0)
    t0 = time.time()
    with torch.no_grad():
        m = ref_model(case, x, y)
        loss = m.loss()
        xs = rng.normal(4242, (16, case["d"]))
        mf, vf = m.predict_f(xs)
    entry = dict(case)
    entry["x_checksum"], entry["y_checksum"] = rng.checksum(x), rng.checksum(y)
    entry["lml"] = float(-loss.item())
    entry["predict"] = dict(seed_xs=4242, mean_f=mf.tolist(), var_f=vf.tolist())
    entry["ref_seconds"] = time.time() - t0
    with torch.no_grad():
        ol = oracle_model(case, x, y).loss()
    entry["oracle_abs_diff"] = float(abs(ol.item() - loss.item()))
    print(f"lml {case['name']}: lml={entry['lml']:.10f} ref_time={entry['ref_seconds']:.1f}s oracle diff {entry['oracle_abs_diff']:.2e}")
    with open(os.path.join(out, "lml_mid_12000.json"), "w") as f:
        json.dump(entry, f, indent=1)


def gen_adam(out):
    """50-step Adam trajectories (base.py:149-151, 260-269)."""
    res = []
    for case in [dict(name="adam_C1_rbf_512_2", kind="Rbf", n=512, d=2, dy=1, variance=1.0, length_scales=1.0, ARD=False, noise=1e-2),
                 dict(name="adam_m52_1024_16_ard", kind="Matern52", n=1024, d=16, dy=1, variance=1.0, length_scales=4.0, ARD=True, noise=1e-2)]:
        x, y = rng.make_regression(case["n"], case["d"], case["dy"], seed=0)
        m = ref_model(case, x, y)
        with quiet():
            losses, _ = m.optimize(method="Adam", max_iter=50, verbose=False)
        o = oracle_model(case, x, y)
        ol = o.optimize_adam(50, 0.01)
        assert rel(ol, losses) < 1e-9, (ol - losses)
        entry = dict(case)
        entry["losses"] = losses.tolist()
        entry["final"] = {
            "kernel.variance": m.kernel.variance.detach().numpy().tolist(),
            "kernel.length_scales": m.kernel.length_scales.detach().numpy().tolist(),
            "likelihood.variance": m.likelihood.variance.detach().numpy().tolist(),
        }
        res.append(entry)
        print(f"adam {case['name']}: {losses[0]:.6f} -> {losses[-1]:.6f}")
    with open(os.path.join(out, "adam_cases.json"), "w") as f:
        json.dump(res, f, indent=1)


ADAM_MID_CASE = dict(name="adam_mid_m52_12288_16", kind="Matern52", n=12288, d=16, dy=1, variance=1.0, length_scales=4.0, ARD=False, noise=1e-2)


def gen_adam_mid(out, steps=10):
    """config 3's training loop ABOVE the refinement threshold (12288 rows): a 10-step Adam trajectory of the reference
    (base.py:149-151, 260-269) on C3's model shape at N = 12288 -- the refined LML, its backward and the optimiser pinned
    jointly (round-4 review item 2; adam_cases.json stops at N = 1024).  ~1-2 min per step on 8 threads, ~25 GB."""
    case = ADAM_MID_CASE
    x, y = rng.make_regression(case["n"], case["d"], case["dy"], seed=0)
    m = ref_model(case, x, y)
    t0 = time.time()
    with quiet():
        losses, _ = m.optimize(method="Adam", max_iter=steps, verbose=False)
    entry = dict(case)
    entry["steps"] = steps
    entry["learning_rate"] = 0.01
    entry["x_checksum"], entry["y_checksum"] = rng.checksum(x), rng.checksum(y)
    entry["losses"] = [float(v) for v in losses]
    entry["final_raw"] = {
        "kernel.variance": m.kernel.variance.detach().numpy().tolist(),
        "kernel.length_scales": m.kernel.length_scales.detach().numpy().tolist(),
        "likelihood.variance": m.likelihood.variance.detach().numpy().tolist(),
    }
    entry["ref_seconds"] = time.time() - t0
    print(f"adam mid: {losses[0]:.8f} -> {losses[-1]:.8f} in {entry['ref_seconds']:.0f}s")
    with open(os.path.join(out, "adam_mid_case.json"), "w") as f:
        json.dump(entry, f, indent=1)
    del m
    o = oracle_model(case, x, y)
    ol = o.optimize_adam(steps, 0.01)
    entry["oracle_rel_diff"] = rel(ol, losses)
    print("adam mid: oracle rel diff", entry["oracle_rel_diff"])
    with open(os.path.join(out, "adam_mid_case.json"), "w") as f:
        json.dump(entry, f, indent=1)
    assert entry["oracle_rel_diff"] < 1e-9


def gen_functions(out):
    """functions.cholesky / trtrs / lt_log_determinant direct (unpinned by the
    reference's tests -- test_functions.py only imports) + jitter ladder."""
    res = {}
    n, k = 96, 5
    a = rng.normal(11, (n, n))
    spd = a @ a.T / n + 0.5 * np.eye(n)
    b = rng.normal(12, (n, k))
    L = rf.cholesky(torch.tensor(spd))
    X = rf.trtrs(torch.tensor(b), L)
    res["spd_seed"], res["b_seed"], res["n"], res["k"] = 11, 12, n, k
    res["chol_frobenius"] = float(L.norm())
    res["chol_diag"] = L.diag().tolist()
    res["logdet"] = float(rf.lt_log_determinant(L))
    res["trtrs"] = X.tolist()
    assert rel(orc.cholesky(torch.tensor(spd)).numpy(), L.numpy()) < 1e-14
    ladder = []
    dup_x = np.repeat(rng.normal(13, (8, 2)), 2, axis=0)  # duplicate rows -> singular K, noise=0
    kdup = rk.Rbf(2).K(torch.tensor(dup_x)).detach()
    for name, mat in [("ones2", torch.tensor([[1.0, 1.0], [1.0, 1.0]], dtype=torch.float64)),
                      ("indef2", torch.tensor([[1.0, 2.0], [2.0, 1.0]], dtype=torch.float64)),
                      ("dup_rows_rbf", kdup)]:
        try:
            Lr = rf.cholesky(mat)
            _, rung = orc.cholesky_rung(mat)
            ladder.append(dict(name=name, ok=True, rung=rung, diag=Lr.diag().tolist()))
        except RuntimeError as e:
            ladder.append(dict(name=name, ok=False, error=str(e)))
    res["ladder"] = ladder
    res["dup_seed"] = 13
    with open(os.path.join(out, "functions_cases.json"), "w") as f:
        json.dump(res, f, indent=1)
    print("functions:", [(l["name"], l.get("rung", l.get("error"))) for l in ladder])


def gen_sparse(out):
    """VFE: the reference's OWN known answers (test/test_models/test_sparse_gpr.py:81-142:
    loss == 8.842242323920674 on test/data/models/sparse_gpr/*.dat, Matern32, all
    hyper-parameters 1) + the same data files' predictions; plus a medium case from rng."""
    from gptorch.models.sparse_gpr import VFE
    ddir = os.path.join(REF, "test", "data", "models", "sparse_gpr")
    pack = {k: np.atleast_2d(np.loadtxt(os.path.join(ddir, k + ".dat"))) for k in
            ["x", "y", "z", "x_test", "vfe_y_mean", "vfe_y_cov"]}
    for k in ["x", "y", "z", "x_test", "vfe_y_mean"]:
        if pack[k].shape[0] == 1:
            pack[k] = pack[k].T
    kern = rk.Matern32(1)
    kern.length_scales.data = torch.zeros(1, dtype=torch.float64)
    kern.variance.data = torch.zeros(1, dtype=torch.float64)
    m = VFE(pack["x"], pack["y"], kern, inducing_points=pack["z"], likelihood=rl.Gaussian(variance=1.0),
            mean_function=rm.Zero(1))
    loss = m.loss().item()
    # pytest.approx default (rel 1e-6), as test_sparse_gpr.py:101; with today's torch the reference
    # itself evaluates to 8.8422395...: sqrt at (near-)coincident x/z amplifies Gram-trick rounding
    assert abs(loss - 8.842242323920674) < 1e-6 * 8.842242323920674, loss
    pack["vfe_loss_reference_run"] = np.array([loss])
    o = orc.VFEOracle(pack["x"], pack["y"], pack["z"], "Matern32", 1.0, 1.0, 1.0)
    assert abs(-o.log_likelihood().item() - loss) < 1e-12
    mu, s = m._predict(torch.tensor(pack["x_test"]), diag=False)
    assert np.allclose(mu.detach().numpy().ravel(), pack["vfe_y_mean"].ravel())
    assert np.allclose(s.detach().numpy(), pack["vfe_y_cov"])
    omu, os_ = o.predict_f(pack["x_test"], diag=False)
    assert np.allclose(omu.numpy().ravel(), pack["vfe_y_mean"].ravel()) and np.allclose(os_.numpy(), pack["vfe_y_cov"])
    np.savez(os.path.join(out, "ref_sparse_gpr_fixtures.npz"), **pack)
    # medium case
    case = dict(n=3000, d=4, dy=2, m=200, kind="Matern52", variance=1.3, length_scales=1.6, noise=0.05)
    x, y = rng.make_regression(case["n"], case["d"], case["dy"], seed=0)
    z = rng.normal(55, (case["m"], case["d"]))
    mm = VFE(x, y, rk.Matern52(case["d"], variance=case["variance"], length_scales=case["length_scales"]),
             inducing_points=z, likelihood=rl.Gaussian(variance=case["noise"]), mean_function=rm.Zero(case["dy"]))
    with torch.no_grad():
        case["elbo"] = float(mm.log_likelihood().item())
        xs = rng.normal(56, (16, case["d"]))
        mu, var = mm._predict(torch.tensor(xs))
        _, cov = mm._predict(torch.tensor(xs), diag=False)
    oo = orc.VFEOracle(x, y, z, "Matern52", case["variance"], case["length_scales"], case["noise"])
    assert abs(oo.log_likelihood().item() - case["elbo"]) < 1e-9 * abs(case["elbo"])
    case.update(seed_z=55, seed_xs=56, mean=mu.tolist(), var=var.tolist(), cov=cov.tolist())

    def ref_grads(model):
        """gradients of the reference's loss w.r.t. its RAW parameters (autograd through
        sparse_gpr.py:108-153), checked against the oracle's autograd."""
        model.zero_grad()
        model.loss().backward()
        return dict(g_variance=model.kernel.variance.grad.tolist(),
                    g_length_scales=model.kernel.length_scales.grad.tolist(),
                    g_noise=model.likelihood.variance.grad.tolist(), g_Z=model.Z.grad.tolist())
    mm = VFE(x, y, rk.Matern52(case["d"], variance=case["variance"], length_scales=case["length_scales"]),
             inducing_points=z, likelihood=rl.Gaussian(variance=case["noise"]), mean_function=rm.Zero(case["dy"]))
    case.update(ref_grads(mm))        # fresh model: _predict freezes Z (sparse_gpr.py:165)
    cases = [case]
    # ARD Rbf, ragged sizes, dy = 1, + a 5-step Adam trajectory with Z trainable
    c2 = dict(n=2500, d=3, dy=1, m=150, kind="Rbf", variance=0.8, length_scales=[0.5, 0.7, 0.9], noise=0.1,
              seed_z=57, seed_xs=58)
    x, y = rng.make_regression(c2["n"], c2["d"], c2["dy"], seed=0)
    z = rng.normal(57, (c2["m"], c2["d"]))

    def build():
        return VFE(x, y, rk.Rbf(c2["d"], variance=c2["variance"], length_scales=np.array(c2["length_scales"]),
                                ARD=True),
                   inducing_points=z.copy(), likelihood=rl.Gaussian(variance=c2["noise"]), mean_function=rm.Zero(1))
    m2 = build()
    with torch.no_grad():
        c2["elbo"] = float(m2.log_likelihood().item())
        xs = rng.normal(58, (16, c2["d"]))
        mu, var = m2._predict(torch.tensor(xs))
        _, cov = m2._predict(torch.tensor(xs), diag=False)
    c2.update(mean=mu.tolist(), var=var.tolist(), cov=cov.tolist())
    m2 = build()          # _predict froze Z (sparse_gpr.py:165); start again for the gradients
    c2.update(ref_grads(m2))
    m3 = build()
    losses, _ = m3.optimize(method="Adam", max_iter=5, verbose=False, learning_rate=0.01)
    c2["adam_losses"] = [float(v) for v in losses]
    c2["adam_final_Z_sum"] = float(m3.Z.detach().sum().item())
    cases.append(c2)
    with open(os.path.join(out, "vfe_cases.json"), "w") as f:
        json.dump(cases, f, indent=1)
    print("sparse: VFE known answer reproduced (8.842242323920674); medium elbo %.8f" % case["elbo"])


def gen_composite(out):
    """GPR with kernels that have no single native kind (Sum / Product / Linear / White,
    kernels.py:238-306): LML, raw-parameter gradients and predictions from the reference."""
    cases = []
    x, y = rng.make_regression(400, 3, 2, seed=0)
    xs = rng.normal(71, (16, 3))
    specs = {"rbf_plus_linear": lambda: rk.Rbf(3, variance=1.2, length_scales=1.5) + rk.Linear(3, variance=np.array([0.3, 0.5, 0.7])),
             "m32_times_rbf": lambda: rk.Matern32(3, variance=0.9, length_scales=2.0) * rk.Rbf(3, variance=1.1, length_scales=np.array([1.0, 2.0, 3.0]), ARD=True),
             "m52_plus_white": lambda: rk.Matern52(3, variance=1.0, length_scales=1.3) + rk.White(3, variance=0.05)}
    for name, mk in specs.items():
        m = RefGPR(x, y, mk(), likelihood=rl.Gaussian(variance=0.05))
        m.zero_grad()
        loss = m.loss()
        loss.backward()
        grads = {n: p.grad.tolist() for n, p in m.named_parameters() if p.grad is not None}
        with torch.no_grad():
            mu, var = m._predict(torch.tensor(xs))
            _, cov = m._predict(torch.tensor(xs), diag=False)
        cases.append(dict(name=name, n=400, d=3, dy=2, noise=0.05, seed_xs=71, loss=float(loss.item()), grads=grads,
                          mean=mu.tolist(), var=var.tolist(), cov=cov.tolist()))
    with open(os.path.join(out, "composite_cases.json"), "w") as f:
        json.dump(cases, f, indent=1)
    print("composite:", [(c["name"], c["loss"], sorted(c["grads"])) for c in cases])


def gen_composite_big(out):
    """the reference's own example model (examples/regression_1d.py:34-53: Linear + Rbf + Constant) at BASELINE configs[1]'s size
    (N = 8192, D = 8): loss, raw-parameter gradients and predictions from the reference -- the fused expression path at a size
    where its tiling and the 1536-column panels are in play (composite_cases.json stops at N = 400)."""
    n, d = 8192, 8
    x, y = rng.make_regression(n, d, 1, seed=0)
    xs = rng.normal(71, (16, d))
    t0 = time.time()
    m = RefGPR(x, y, rk.Linear(d, variance=0.3) + rk.Rbf(d, variance=1.2, length_scales=float(np.sqrt(d))) + rk.Constant(d, variance=0.4),
               likelihood=rl.Gaussian(variance=0.01))
    m.zero_grad()
    loss = m.loss()
    loss.backward()
    grads = {nm: p.grad.tolist() for nm, p in m.named_parameters() if p.grad is not None}
    with torch.no_grad():
        mu, var = m._predict(torch.tensor(xs))
    case = dict(name="linear_plus_rbf_plus_constant_8192_8", n=n, d=d, dy=1, noise=0.01, seed_xs=71, loss=float(loss.item()), grads=grads,
                mean=mu.tolist(), var=var.tolist(), ref_seconds=time.time() - t0)
    with open(os.path.join(out, "composite_big_case.json"), "w") as f:
        json.dump(case, f, indent=1)
    print("composite big:", case["loss"], {k: v for k, v in grads.items()}, "%.1f s" % case["ref_seconds"])


def gen_composite_16k(out):
    """the example model once more at N = 16384 -- above the size from which the native paths refine the quadratic form:
    loss and raw-parameter gradients from the reference (about 30 GB of host memory for its autograd)."""
    n, d = 16384, 8
    x, y = rng.make_regression(n, d, 1, seed=0)
    t0 = time.time()
    m = RefGPR(x, y, rk.Linear(d, variance=0.3) + rk.Rbf(d, variance=1.2, length_scales=float(np.sqrt(d))) + rk.Constant(d, variance=0.4),
               likelihood=rl.Gaussian(variance=0.01))
    m.zero_grad()
    loss = m.loss()
    loss.backward()
    grads = {nm: p.grad.tolist() for nm, p in m.named_parameters() if p.grad is not None}
    case = dict(name="linear_plus_rbf_plus_constant_16384_8", n=n, d=d, dy=1, noise=0.01, loss=float(loss.item()), grads=grads,
                ref_seconds=time.time() - t0)
    with open(os.path.join(out, "composite_16k_case.json"), "w") as f:
        json.dump(case, f, indent=1)
    print("composite 16k:", case["loss"], "%.1f s" % case["ref_seconds"])


def gen_sparse_wellcond(out):
    """a WELL-CONDITIONED VFE case held to north_star's 1e-8 ABSOLUTE (round-4 review item 3): inducing points = k-means
    centres of the data (distinct, spread like the data: K(Z) factors without any jitter rung, cond(K(Z)) recorded), noise 1e-2,
    N = 8192, M = 256, D = 4.  The bound, its raw-parameter / inducing-point gradients and predictions from the reference
    (sparse_gpr.py:108-195); the centres are stored with the fixture (k-means is not bit-reproducible across library versions)."""
    from gptorch.models.sparse_gpr import VFE
    from scipy.cluster.vq import kmeans2
    n, d, m = 8192, 4, 256
    x, y = rng.make_regression(n, d, 1, seed=0)
    z, _ = kmeans2(x, m, minit="points", seed=1234, iter=20)
    case = dict(name="vfe_wellcond_rbf_8192_256_4", n=n, d=d, dy=1, m=m, kind="Rbf", variance=1.2, length_scales=1.0, noise=1e-2, seed_xs=71)

    def build():
        return VFE(x, y, rk.Rbf(d, variance=case["variance"], length_scales=case["length_scales"]), inducing_points=z.copy(),
                   likelihood=rl.Gaussian(variance=case["noise"]), mean_function=rm.Zero(1))
    mm = build()
    with torch.no_grad():
        Kuu = mm.kernel.K(mm.Z)
        ev = torch.linalg.eigvalsh(Kuu)
        case["cond_Kuu"] = float(ev[-1] / ev[0])
        torch.linalg.cholesky(Kuu)                      # factors as it is: no ladder rung
        case["elbo"] = float(mm.log_likelihood().item())
        xs = rng.normal(case["seed_xs"], (16, d))
        mu, var = mm._predict(torch.tensor(xs))
        _, cov = mm._predict(torch.tensor(xs), diag=False)
    oo = orc.VFEOracle(x, y, z, "Rbf", case["variance"], case["length_scales"], case["noise"])
    case["oracle_abs_diff"] = float(abs(oo.log_likelihood().item() - case["elbo"]))
    assert case["oracle_abs_diff"] < 1e-8, case["oracle_abs_diff"]
    mg = build()
    mg.zero_grad()
    mg.loss().backward()
    case.update(g_variance=mg.kernel.variance.grad.tolist(), g_length_scales=mg.kernel.length_scales.grad.tolist(),
                g_noise=mg.likelihood.variance.grad.tolist())
    np.savez(os.path.join(out, "vfe_wellcond_case.npz"), z=z, mean=mu.numpy(), var=var.numpy(), cov=cov.numpy(), g_Z=mg.Z.grad.numpy())
    with open(os.path.join(out, "vfe_wellcond_case.json"), "w") as f:
        json.dump(case, f, indent=1)
    print("sparse well-conditioned: elbo %.10f cond(Kuu) %.3e oracle diff %.2e" % (case["elbo"], case["cond_Kuu"], case["oracle_abs_diff"]))


def gen_sparse_composite(out):
    """VFE over kernels without a single native kind (sparse_gpr.py:126-129 takes any kernel object):
    bound, raw-parameter / inducing-point gradients and predictions from the reference."""
    from gptorch.models.sparse_gpr import VFE
    n, d, m = 700, 3, 40
    x, y = rng.make_regression(n, d, 1, seed=0)
    z = rng.normal(61, (m, d))
    xs = rng.normal(62, (12, d))
    specs = {"rbf_plus_linear": lambda: rk.Rbf(d, variance=1.1, length_scales=1.4) + rk.Linear(d, variance=np.array([0.2, 0.4, 0.6])),
             "m52_times_rbf": lambda: rk.Matern52(d, variance=0.9, length_scales=2.0) * rk.Rbf(d, variance=1.2, length_scales=np.array([1.0, 2.0, 3.0]), ARD=True)}
    cases = []
    for name, mk in specs.items():
        def build():
            return VFE(x, y, mk(), inducing_points=z.copy(), likelihood=rl.Gaussian(variance=0.05), mean_function=rm.Zero(1))
        mm = build()
        with torch.no_grad():
            elbo = float(mm.log_likelihood().item())
            mu, var = mm._predict(torch.tensor(xs))
            _, cov = mm._predict(torch.tensor(xs), diag=False)
        mg = build()                      # _predict froze Z (sparse_gpr.py:165)
        mg.zero_grad()
        mg.loss().backward()
        grads = {nm: p.grad.tolist() for nm, p in mg.named_parameters() if p.grad is not None}
        cases.append(dict(name=name, n=n, d=d, m=m, noise=0.05, seed_z=61, seed_xs=62, elbo=elbo, grads=grads,
                          mean=mu.tolist(), var=var.tolist(), cov=cov.tolist()))
    with open(os.path.join(out, "vfe_composite_cases.json"), "w") as f:
        json.dump(cases, f, indent=1)
    print("sparse composite:", [(c["name"], c["elbo"], sorted(c["grads"])) for c in cases])


def gen_c2_grad(out):
    """BASELINE config 2 (N = 8192, D = 8, Rbf): d loss / d raw parameters from autograd through the
    reference (gpr.py:47-67 + PyTorch's CholeskyBackward0 ...), ~15 s -- the full-size gradient pin."""
    case = C2_CASE
    x, y = rng.make_regression(case["n"], case["d"], case["dy"], seed=0)
    t0 = time.time()
    m = ref_model(case, x, y)
    loss = m.loss()
    loss.backward()
    entry = dict(case)
    entry["x_checksum"], entry["y_checksum"] = rng.checksum(x), rng.checksum(y)
    entry["lml"] = float(-loss.item())
    entry["grad_loss"] = {"kernel.variance": m.kernel.variance.grad.numpy().tolist(),
                          "kernel.length_scales": m.kernel.length_scales.grad.numpy().tolist(),
                          "likelihood.variance": m.likelihood.variance.grad.numpy().tolist()}
    entry["ref_seconds"] = time.time() - t0
    o = oracle_model(case, x, y)
    ol, og = o.loss_and_grads()
    for a, b in zip(og, [m.kernel.variance.grad, m.kernel.length_scales.grad, m.likelihood.variance.grad]):
        assert rel(a.numpy(), b.numpy()) < 1e-10, (a, b)
    with open(os.path.join(out, "lml_c2_grad.json"), "w") as f:
        json.dump(entry, f, indent=1)
    print("c2 grad:", entry["lml"], entry["grad_loss"], "%.1f s" % entry["ref_seconds"])


def gen_lbfgs(out):
    """What examples/regression_1d.py:34-53 actually runs: GPR over Linear + Rbf + Constant on n = 100
    points, model.optimize(method="L-BFGS-B") (base.py:298-320 -> scipy.optimize.minimize over
    model._loss_and_grad, model.py:84-133).  Golden = every loss value the reference printed
    (one per function evaluation), the final flat parameter vector and predictions."""
    rs = np.random.RandomState(42)
    n = 100
    x = np.linspace(0, 1, n).reshape((-1, 1))
    y = np.sin(2.0 * np.pi * x) + np.cos(3.5 * np.pi * x) - 3.0 * x + 5.0 + 0.1 * rs.randn(n, 1)
    kern = rk.Linear(1) + rk.Rbf(1) + rk.Constant(1)
    m = RefGPR(x, y, kern)
    buf = io.StringIO()
    with contextlib.redirect_stdout(buf):
        m.optimize(method="L-BFGS-B", max_iter=25)
    losses = [float(l.split("loss:")[1]) for l in buf.getvalue().splitlines() if l.startswith("loss:")]
    names = [nm for nm, p in m.named_parameters() if p.requires_grad]
    xt = np.linspace(-0.2, 1.2, 11).reshape((-1, 1))
    with torch.no_grad():
        mu, var = m.predict_y(xt)
    res = dict(n=n, seed=42, max_iter=25, x=x.ravel().tolist(), y=y.ravel().tolist(), losses=losses,
               param_names=names, final_params=m._get_param_array().tolist(), final_loss=float(m.loss().item()),
               x_test=xt.ravel().tolist(), mean_y=np.asarray(mu).ravel().tolist(), var_y=np.asarray(var).ravel().tolist())
    with open(os.path.join(out, "lbfgs_case.json"), "w") as f:
        json.dump(res, f, indent=1)
    print("lbfgs:", len(losses), "evaluations", losses[:3], "...", losses[-1], names)


def gen_api(out):
    """API behaviours of the shell."""
    x, y = rng.make_regression(20, 3, 2, seed=5)
    m = RefGPR(x, y, rk.Rbf(3, ARD=True))
    names = [(n, bool(p.requires_grad), list(p.shape)) for n, p in m.named_parameters()]
    m2 = RefGPR(torch.tensor(x), torch.tensor(y), rk.Rbf(3))
    res = dict(param_names=names,
               default_noise_numpy=float(m.likelihood.variance.transform().item()),
               default_noise_tensor=float(m2.likelihood.variance.transform().item()),
               loss_shape=list(m.loss().shape), seed=5, n=20, d=3, dy=2,
               loss_default_numpy=float(m.loss().item()))
    sd = ru.squared_distance(torch.tensor([[0.0], [1.0], [2.0]], dtype=torch.float64) + 1.0 / 65.0,
                             torch.tensor([[0.0], [2.0], [4.0]], dtype=torch.float64) + 1.0 / 65.0)
    res["squared_distance_test_util"] = sd.tolist()
    with open(os.path.join(out, "api_cases.json"), "w") as f:
        json.dump(res, f, indent=1)
    print("api:", names)


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--big", action="store_true")
    ap.add_argument("--only", default="")
    args = ap.parse_args()
    torch.manual_seed(0)
    steps = dict(refk=lambda: gen_ref_kernel_fixtures(HERE), kern=lambda: gen_kernel_cases(HERE),
                 lml=lambda: gen_lml(HERE, args.big), mid=lambda: gen_mid(HERE), adam=lambda: gen_adam(HERE), adammid=lambda: gen_adam_mid(HERE),
                 func=lambda: gen_functions(HERE), api=lambda: gen_api(HERE), sparse=lambda: gen_sparse(HERE),
                 comp=lambda: gen_composite(HERE), compbig=lambda: gen_composite_big(HERE), comp16k=lambda: gen_composite_16k(HERE), c2grad=lambda: gen_c2_grad(HERE), spcomp=lambda: gen_sparse_composite(HERE), spwell=lambda: gen_sparse_wellcond(HERE), lbfgs=lambda: gen_lbfgs(HERE))
    for k, fn in steps.items():
        if (not args.only and k != "adammid") or k in args.only.split(","):      # adammid: ~30 min, by name only
            fn()

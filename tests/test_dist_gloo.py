"""CPU suite, part 5: the 2-D block-cyclic orchestration (gptorch_amd/dist.py) under
gloo with world_size 2 and 4, tile arithmetic supplied by a torch-CPU TileOps (test
infrastructure; the product's NativeTileOps needs a GPU and is covered by -m gpu)."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from gptorch_amd import dist as gdist
from gptorch_amd import rng
from oracle import gp_oracle as orc


class CpuTileOps:
    """torch-CPU stand-in for NativeTileOps (same contract)."""

    def new_tile(self, rows, cols):
        return torch.zeros((rows + 127) // 128 * 128 + 16, (cols + 127) // 128 * 128, dtype=torch.float64)

    def kernel_tile(self, kind, Xi, Xj, variance, ls, noise, out):
        K = orc.kernel_K(kind, Xi, Xj, variance, ls)
        if Xj is None and noise is not None:
            K = K + noise * torch.eye(K.shape[0], dtype=torch.float64)
        out[:K.shape[0], :K.shape[1]] = K

    def potrf(self, tile, n):
        info = torch.zeros(1, dtype=torch.int32)
        L, inf = torch.linalg.cholesky_ex(tile[:n, :n])
        info[0] = int(inf)
        if int(inf) == 0:
            tile[:n, :n] = L
        return torch.zeros(1, dtype=torch.float64), info

    def winv_numel(self, n):
        return 1

    def trsm(self, L, winv, n, B, m):
        B[:m, :n] = torch.linalg.solve_triangular(L[:n, :n], B[:m, :n].t(), upper=False).t()

    def update(self, C, A, B, m, n, k, lower, alpha=-1.0):
        upd = A[:m, :k] @ B[:n, :k].t()
        if lower:
            upd = torch.tril(upd)
        C[:m, :n] += alpha * upd

    def set_identity(self, tile, n):
        tile.diagonal()[:n].fill_(1.0)

    def kernel_grad(self, kind, Xi, Xj, variance, ls, G):
        v = variance.clone().requires_grad_(True)
        l = ls.clone().requires_grad_(True)
        (orc.kernel_K(kind, Xi, Xj, v, l) * G).sum().backward()
        return torch.cat([v.grad, l.grad])

    def log_diag_sum(self, tile, n):
        return tile.diagonal()[:n].log().sum()

    def sumsq(self, tile, m, n):
        return tile[:m, :n].pow(2).sum()


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, n, d, dy, tile, kind, noise, out_path):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    torch.set_num_threads(1)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        x, y = rng.make_regression(n, d, dy, seed=0)
        X, Y = torch.tensor(x), torch.tensor(y)
        g = gdist.BlockCyclicGP(X, Y, kind, tile=tile, ops=CpuTileOps())
        var, ls = torch.tensor([1.3], dtype=torch.float64), torch.tensor([1.7], dtype=torch.float64)
        lml = g.log_likelihood(var, ls, torch.tensor([noise], dtype=torch.float64), Y)
        # every rank must hold only its block-cyclic share
        for (I, J) in g.tiles:
            assert g.owner(I, J) == rank
        if rank == 0:
            np.save(out_path, np.array([float(lml), float(g.info)]))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,n,tile,dy,kind", [(2, 300, 128, 1, "Rbf"), (4, 700, 128, 2, "Matern52"),
                                                    (2, 129, 128, 1, "Rbf"), (4, 128, 128, 1, "Rbf")])
def test_block_cyclic_lml_matches_oracle(tmp_path, world, n, tile, dy, kind):
    out = str(tmp_path / "lml.npy")
    mp.spawn(_worker, args=(world, _free_port(), n, 3, dy, tile, kind, 0.05, out), nprocs=world, join=True)
    lml, info = np.load(out)
    x, y = rng.make_regression(n, 3, dy, seed=0)
    o = orc.GPROracle(x, y, kind=kind, variance=1.3, length_scales=1.7, noise=0.05)
    with torch.no_grad():
        ref = o.log_likelihood().item()
    assert info == 0
    assert abs(lml - ref) < 1e-9 * max(1.0, abs(ref)), (lml, ref)


def test_grid_and_ownership():
    assert gdist.choose_grid(1) == (1, 1) and gdist.choose_grid(2) == (1, 2)
    assert gdist.choose_grid(4) == (2, 2) and gdist.choose_grid(8) == (2, 4)
    x, y = rng.make_regression(200, 2, 1, seed=0)
    g = gdist.BlockCyclicGP(torch.tensor(x), torch.tensor(y), "Rbf", tile=128, ops=CpuTileOps())
    assert g.nt == 2 and g.rows_of(1) == 72 and g.rows_of(2) == 1
    one = torch.ones(1, dtype=torch.float64)
    lml = g.log_likelihood(one, one, 0.1 * one, torch.tensor(y))      # world_size 1 path
    o = orc.GPROracle(x, y, kind="Rbf", noise=0.1)
    with torch.no_grad():
        assert abs(lml.item() - o.log_likelihood().item()) < 1e-9


def _grad_worker(rank, world, port, n, d, dy, tile, kind, noise, out_path):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    torch.set_num_threads(1)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        x, y = rng.make_regression(n, d, dy, seed=0)
        X, Y = torch.tensor(x), torch.tensor(y)
        g = gdist.BlockCyclicGP(X, Y, kind, tile=tile, ops=CpuTileOps())
        var = torch.tensor([1.3], dtype=torch.float64)
        ls = torch.tensor([1.1, 1.7, 2.3][:d], dtype=torch.float64)
        lml, grad = g.log_likelihood_and_grad(var, ls, torch.tensor([noise], dtype=torch.float64), Y)
        if rank == 0:
            np.save(out_path, np.concatenate([[float(lml)], grad.numpy()]))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,n,tile,dy,kind", [(2, 300, 128, 1, "Rbf"), (4, 700, 128, 2, "Matern52"), (1, 260, 128, 3, "Rbf")])
def test_block_cyclic_gradients_match_oracle(tmp_path, world, n, tile, dy, kind):
    """distributed closed-form backward (U = L^-T carried as identity rows, Kyy^-1 = U U^T on the
    grid, D + 2 scalars all-reduced) vs the oracle's closed form (gradients w.r.t. log-parameters
    = value * gradient w.r.t. the constrained value)."""
    out = str(tmp_path / "g.npy")
    mp.spawn(_grad_worker, args=(world, _free_port(), n, 3, dy, tile, kind, 0.05, out), nprocs=world, join=True)
    got = np.load(out)
    x, y = rng.make_regression(n, 3, dy, seed=0)
    ls = np.array([1.1, 1.7, 2.3])
    ref = orc.lml_closed_form_grads(kind, x, y, 1.3, ls, 0.05)
    ref_lml = float(ref[0])
    ref_g = np.concatenate([[float(ref[1]) / 1.3], np.asarray(ref[2], dtype=np.float64).ravel() / ls, [float(ref[3]) / 0.05]])
    assert abs(got[0] - ref_lml) < 1e-9 * max(1.0, abs(ref_lml))
    assert np.abs(got[1:] - ref_g).max() < 1e-8 * max(1.0, np.abs(ref_g).max()), (got[1:], ref_g)

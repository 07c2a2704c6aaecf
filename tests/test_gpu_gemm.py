"""fp64 MFMA contraction (csrc/gemm_f64.hip) against exact integer data and
torch.matmul on the same device (checker only)."""
import pytest
import torch

from gptorch_amd import _native, _ops

pytestmark = pytest.mark.gpu


def _pad_rows(t, mult=16):
    r = (-t.shape[0]) % mult
    if r:
        t = torch.cat([t, torch.zeros(r, t.shape[1], dtype=t.dtype, device=t.device)])
    return t.contiguous()


@pytest.mark.parametrize("variant", [0, 1])
@pytest.mark.parametrize("M,N,K", [(16, 16, 16), (64, 64, 64), (128, 128, 32), (200, 72, 48), (1, 130, 16),
                                   (333, 257, 80), (512, 384, 256), (1024, 1024, 512)])
def test_gemm_exact_integers(device, variant, M, N, K):
    """Small integers: every product and sum is exact in fp64, so the result must be
    bit-identical whatever the summation order; A != B and asymmetric, which catches
    row/column swaps in the MFMA fragment maps."""
    _native.debug_begin().gpn_debug_set_gemm_variant(variant)
    try:
        g = torch.Generator(device="cpu").manual_seed(M * 7 + N * 3 + K)
        A = torch.randint(-8, 9, (M, K), generator=g).double().to(device)
        B = torch.randint(-8, 9, (N, K), generator=g).double().to(device)
        C0 = torch.randint(-8, 9, (M, N), generator=g).double().to(device)
        Ap, Bp = _pad_rows(A), _pad_rows(B)
        C = C0.clone()
        _ops.gemm_nt(Ap, Bp, M, N, K, alpha=-1.0, beta=1.0, C=C)
        ref = C0 - A @ B.t()
        assert torch.equal(C, ref)
        C2 = _ops.gemm_nt(Ap, Bp, M, N, K)
        assert torch.equal(C2, A @ B.t())
    finally:
        _native.debug_end()


@pytest.mark.parametrize("n,K", [(64, 64), (200, 32), (640, 128), (1100, 64)])
def test_syrk_lower(device, n, K):
    g = torch.Generator(device="cpu").manual_seed(n + K)
    P = torch.randint(-5, 6, (n, K), generator=g).double().to(device)
    C0 = torch.randint(-5, 6, (n, n), generator=g).double().to(device)
    Pp = _pad_rows(P)
    C = C0.clone()
    _ops.gemm_nt(Pp, Pp, n, n, K, alpha=-1.0, beta=1.0, C=C, lower=True)
    full = C0 - P @ P.t()
    assert torch.equal(torch.tril(C), torch.tril(full))
    # strictly-upper entries must be untouched
    assert torch.equal(torch.triu(C, 1), torch.triu(C0, 1))


@pytest.mark.parametrize("variant", [0, 3, 4, 6, 8, 11])
@pytest.mark.parametrize("M,N,K", [(128, 128, 64), (2048 + 640, 2048, 256), (9000, 512, 128), (700, 200, 48), (5000, 72, 64), (33000, 2048, 32)])
def test_trapezoid_launch(device, variant, M, N, K):
    """lower = 2: the N x N top square lower-tile only (entries above its diagonal untouched), the rows below it
    whole -- one tile column of a block-cyclic trailing update incl. its diagonal tile, in one launch; every tile
    shape, exact integer data."""
    _native.debug_begin().gpn_debug_set_gemm_variant(variant)
    try:
        g = torch.Generator(device="cpu").manual_seed(M + 3 * N + K)
        A = torch.randint(-4, 5, (M, K), generator=g).double().to(device)
        B = torch.randint(-4, 5, (N, K), generator=g).double().to(device)
        C0 = torch.randint(-4, 5, (M, N), generator=g).double().to(device)
        C = C0.clone()
        _ops.gemm_nt(_pad_rows(A), _pad_rows(B), M, N, K, alpha=-1.0, beta=1.0, C=C, lower=2)
        full = C0 - A @ B.t()
        keep = torch.ones(M, N, dtype=torch.bool, device=device).tril()       # j <= i (all of the rows below the square)
        assert torch.equal(C[keep], full[keep])
        assert torch.equal(C[~keep], C0[~keep])
    finally:
        _native.debug_end()


@pytest.mark.parametrize("variant", [0, 3, 4, 8, 11])
@pytest.mark.parametrize("M,nb,blk,K,step,diag", [(1024, 1, 256, 64, 0, 1), (3000, 4, 256, 128, 512, 1), (3000, 4, 256, 128, 256, 0),
                                                  (2500, 6, 128, 48, 512, 1), (9000, 3, 2048, 256, 2048, 1), (20000, 5, 512, 64, 1024, 0),
                                                  (1500, 8, 128, 32, 256, 1)])
def test_staircase_launch(device, variant, M, nb, blk, K, step, diag):
    """gpn_gemm_nt_stair: column block b has the rows from b*step on (those above stay untouched), with diag a
    lower-only first square -- the local tile columns of a block-cyclic trailing update in ONE launch; incl. blocks
    that start below the last row (no work) and every tile shape; exact integer data."""
    _native.debug_begin().gpn_debug_set_gemm_variant(variant)
    try:
        g = torch.Generator(device="cpu").manual_seed(M + 5 * nb + blk + K + step)
        N = nb * blk
        A = torch.randint(-4, 5, (M, K), generator=g).double().to(device)
        B = torch.randint(-4, 5, (N, K), generator=g).double().to(device)
        C0 = torch.randint(-4, 5, (M, N), generator=g).double().to(device)
        C = C0.clone()
        _ops.gemm_nt_stair(_pad_rows(A), _pad_rows(B), C, M, nb, blk, K, step, diag)
        full = C0 - A @ B.t()
        rows = torch.arange(M, device=device)[:, None]
        cols = torch.arange(N, device=device)[None, :]
        start = (cols // blk) * step
        keep = rows >= start
        if diag:
            keep = keep & ((cols % blk) <= (rows - start))
        assert torch.equal(C[keep], full[keep])
        assert torch.equal(C[~keep], C0[~keep])
    finally:
        _native.debug_end()


def test_gemm_random_fp64(device):
    torch.manual_seed(0)
    M, N, K = 700, 900, 1024
    A = torch.randn(M + 4, K, dtype=torch.float64, device=device)
    B = torch.randn(N + 12, K, dtype=torch.float64, device=device)
    C = _ops.gemm_nt(A, B, M, N, K)
    ref = A[:M] @ B[:N].t()
    assert (C - ref).abs().max().item() < 1e-11


@pytest.mark.parametrize("variant", [3, 4, 5, 6, 7, 8, 9, 10, 11])   # 7..11: the pipelined loop / 8-wave forms (8 and 11 ship)
@pytest.mark.parametrize("M,N,K,lower", [(333, 257, 80, False), (1000, 700, 144, False), (129, 127, 16, False),
                                         (777, 777, 96, True), (1100, 1100, 64, True), (130, 130, 32, True)])
def test_every_tile_shape_on_ragged_sizes(device, variant, M, N, K, lower):
    """The launcher picks a tile shape by problem size (128x128 from 8 rounds of tiles up, 64x64,
    64x64 / 32x32 with the deep LDS-DMA ring); here each one is FORCED onto small ragged problems
    (sizes that are no multiple of any tile edge) with exact integer data: bit-identical results."""
    _native.debug_begin().gpn_debug_set_gemm_variant(variant)
    try:
        g = torch.Generator(device="cpu").manual_seed(variant * 1000 + M + N + K)
        A = torch.randint(-8, 9, (M, K), generator=g).double().to(device)
        B = A if lower else torch.randint(-8, 9, (N, K), generator=g).double().to(device)
        C0 = torch.randint(-8, 9, (M, N), generator=g).double().to(device)
        Ap, Bp = _pad_rows(A), _pad_rows(B)
        C = C0.clone()
        _ops.gemm_nt(Ap, Bp, M, N, K, alpha=-1.0, beta=1.0, C=C, lower=lower)
        ref = C0 - A @ B.t()
        if lower:
            assert torch.equal(torch.tril(C), torch.tril(ref))
            assert torch.equal(torch.triu(C, 1), torch.triu(C0, 1))
        else:
            assert torch.equal(C, ref)
    finally:
        _native.debug_end()


@pytest.mark.parametrize("variant", [0, 3, 4, 6, 8, 11])
@pytest.mark.parametrize("n", [200, 777, 1300])
def test_k_clipped_triangular_operands(device, variant, n):
    """GPN_TRI_* flags (the K range of a tile is clipped where an operand is structurally zero) on
    every tile shape: U U^T for an upper-triangular U (the backward's Kyy^-1 product) and B W^T for
    a lower-triangular W, exact integers."""
    _native.debug_begin().gpn_debug_set_gemm_variant(variant)
    try:
        g = torch.Generator(device="cpu").manual_seed(variant * 100 + n)
        kp = (n + 15) // 16 * 16
        U = torch.zeros(kp + 16, kp, dtype=torch.float64)
        U[:n, :n] = torch.triu(torch.randint(-4, 5, (n, n), generator=g).double())
        U = U.to(device)
        C = torch.zeros(n, n, dtype=torch.float64, device=device)
        _ops.gemm_nt(U, U, n, n, kp, C=C, lower=True, tri=_ops.TRI_A_UPPER | _ops.TRI_B_UPPER)
        ref = U[:n, :n] @ U[:n, :n].t()
        assert torch.equal(torch.tril(C), torch.tril(ref))
        W = torch.zeros(kp + 16, kp, dtype=torch.float64)
        W[:n, :n] = torch.tril(torch.randint(-4, 5, (n, n), generator=g).double())
        W = W.to(device)
        Bm = torch.zeros(112, kp, dtype=torch.float64)
        Bm[:100, :n] = torch.randint(-4, 5, (100, n), generator=g).double()
        Bm = Bm.to(device)
        X = _ops.gemm_nt(Bm, W, 100, n, kp, tri=_ops.TRI_B_LOWER)
        assert torch.equal(X, Bm[:100, :n] @ W[:n, :n].t())
    finally:
        _native.debug_end()

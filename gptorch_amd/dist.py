"""
2-D block-cyclic exact-GP log marginal likelihood over the GPUs of one node
(BASELINE.json config 4: N = 65536 across 8 x MI355X).

One process per GPU (`torch.distributed`, backend "nccl" = RCCL over xGMI).  The
N x N Gram matrix is cut into T x T tiles; tile (I, J), I >= J, lives on rank
(I mod Pr) * Pc + (J mod Pc) of a Pr x Pc process grid.  X [N, D] is tiny and
replicated; every rank ASSEMBLES ITS OWN TILES with the native K-assembly kernel
(no communication).  Right-looking factorisation, one exchange step per tile column k:

  1. owner of (k,k) factors it (gpn_potrf_lower) and broadcasts L_kk (+ the inverses of
     its 128x128 diagonal blocks) down its process COLUMN  -> column sub-communicator
  2. owners of (I,k), I > k, solve  A_Ik <- A_Ik L_kk^-T  (gpn_trsm_right_lt)
  3. panel tile (I,k) is broadcast along process ROW I mod Pr (it multiplies from the
     left in the updates of tile row I) and along process COLUMN I mod Pc (it multiplies
     from the right in tile column I)            -> row / column sub-communicators
  4. every rank updates the trailing tiles it owns:  A_IJ -= P_I P_J^T  (gpn_gemm_nt,
     lower-only on diagonal tiles)

Look-ahead: after the exchange of panel k, tile column k+1 is updated FIRST and
its diagonal tile is factored, broadcast and its panel solved; the row and column broadcasts of
panel k+1 are then issued asynchronously (`async_op=True`: RCCL runs them on its own stream) in
front of the two halves of the remaining trailing update by panel k (`factor`), so the next
panel's collectives are in flight while every rank is busy with the bulk of step 4.

The residual (y - m)^T is carried as one extra tile ROW (index nt) exactly like the
single-GPU "extra rows", so alpha^T = (L^-1 (y-m))^T falls out of steps 2-4.  xGMI is a
full mesh of point-to-point links, so the row/column broadcasts of step 3 run on
disjoint links concurrently; only the two scalars (sum log L_ii, ||alpha||^2) are
all-reduced.

The dense arithmetic is behind a small `TileOps` interface: `NativeTileOps` (the product)
calls libgpnative through `_ops`; the CPU test-suite injects a torch-CPU implementation
to exercise this orchestration under gloo with world_size 2 and 4 (tests/test_dist_gloo.py).
"""
import math

import torch
import torch.distributed as dist

from . import _ops


def choose_grid(world):
    """Pr x Pc with Pr <= Pc, as square as possible: 1x1, 1x2, 2x2, 2x4."""
    pr = int(math.sqrt(world))
    while world % pr:
        pr -= 1
    return pr, world // pr


class NativeTileOps:
    """Tile arithmetic on libgpnative (fp64 tensors on this rank's GPU)."""

    def __init__(self, device):
        self.device = device

    def new_tile(self, rows, cols):
        """zeroed factor-style buffer holding a rows x cols tile (ld multiple of 128, +apron)."""
        return torch.zeros(_ops.round_up(rows, _ops.LEAF) + 16, _ops.round_up(cols, _ops.LEAF), dtype=torch.float64,
                           device=self.device)

    def kernel_tile(self, kind, Xi, Xj, variance, ls, noise, out):
        """out[:ri, :rj] <- K(Xi, Xj) (+ noise*I when Xj is None)."""
        _ops.kernel_matrix(kind, Xi, Xj, variance, ls, noise=noise, out=out, ldk=out.stride(0))

    def potrf(self, tile, n):
        """in-place lower Cholesky of tile[:n,:n]; returns (winv, info_tensor)."""
        winv = torch.empty(int(_ops._native.lib().gpn_winv_bytes(n)) // 8, dtype=torch.float64, device=tile.device)
        info = torch.zeros(1, dtype=torch.int32, device=tile.device)
        st = _ops._native.lib().gpn_potrf_lower(_ops._stream(tile.device), _ops._ptr(tile), n, 0, tile.stride(0),
                                                _ops._ptr(winv), _ops._ptr(info))
        _ops._native.check(st, "gpn_potrf_lower")
        return winv, info

    def winv_numel(self, n):
        return int(_ops._native.lib().gpn_winv_bytes(n)) // 8

    def trsm(self, L, winv, n, B, m):
        """B[:m,:n] <- B L^-T."""
        st = _ops._native.lib().gpn_trsm_right_lt(_ops._stream(B.device), _ops._ptr(L), n, L.stride(0),
                                                  _ops._ptr(winv), _ops._ptr(B), m, B.stride(0))
        _ops._native.check(st, "gpn_trsm_right_lt")

    def update(self, C, A, B, m, n, k, lower, alpha=-1.0):
        """C[:m,:n] += alpha A[:m,:k] B[:n,:k]^T (lower: only j <= i)."""
        _ops.gemm_nt(A, B, m, n, _ops.round_up(k, 16), alpha=alpha, beta=1.0, C=C, lower=lower)

    def set_identity(self, tile, n):
        tile.diagonal()[:n].fill_(1.0)

    def kernel_grad(self, kind, Xi, Xj, variance, ls, G):
        """-> tensor [1 + nls]: sum G * dK(Xi, Xj)/d(variance, length_scales) (gpn_kernel_grad)."""
        from . import _backward
        gv, gl = _backward.kernel_backward(kind, Xi, Xj, variance, ls, G)
        return torch.cat([gv, gl])

    def log_diag_sum(self, tile, n):
        return tile.diagonal()[:n].log().sum()

    def sumsq(self, tile, m, n):
        return tile[:m, :n].pow(2).sum()


class BlockCyclicGP:
    """Distributed LML for a stationary kernel.  All ranks call every method collectively."""

    def __init__(self, X, Y, kind, tile=2048, grid=None, ops=None, group=None):
        self.rank = dist.get_rank(group) if dist.is_initialized() else 0
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.pr, self.pc = grid if grid is not None else choose_grid(self.world)
        assert self.pr * self.pc == self.world
        self.my_r, self.my_c = divmod(self.rank, self.pc)
        self.X, self.Y, self.kind = X, Y, kind
        self.n, self.dy = Y.shape
        self.T = int(tile)
        assert self.T % _ops.LEAF == 0
        self.nt = (self.n + self.T - 1) // self.T
        self.ops = ops if ops is not None else NativeTileOps(X.device)
        self.group = group
        # sub-communicators: one per process row and per process column (created collectively)
        self.row_groups, self.col_groups = {}, {}
        if self.world > 1:
            for r in range(self.pr):
                ranks = [r * self.pc + c for c in range(self.pc)]
                g = dist.new_group(ranks)
                self.row_groups[r] = (g, ranks)
            for c in range(self.pc):
                ranks = [r * self.pc + c for r in range(self.pr)]
                g = dist.new_group(ranks)
                self.col_groups[c] = (g, ranks)
        self.tiles = {}
        self.info = 0
        self.with_inverse = False      # carry I through the factorisation (-> U = L^-T) for the backward

    # -- geometry ---------------------------------------------------------------
    def owner(self, I, J):
        """tile rows 0..nt-1: the matrix; nt: the residual rows; nt+1+i: identity rows of block i
        (backward only; they live in process row i mod Pr, like matrix row i)."""
        if I > self.nt:
            I = I - self.nt - 1
        return (I % self.pr) * self.pc + (J % self.pc)

    def mine(self, I, J):
        return self.owner(I, J) == self.rank

    def rows_of(self, I):
        """row count of tile row I (tile row nt = the residual rows)."""
        if I == self.nt:
            return self.dy
        if I > self.nt:
            I = I - self.nt - 1
        return min(self.T, self.n - I * self.T)

    def _prow(self, I):
        """process row of tile row I."""
        return ((I - self.nt - 1) if I > self.nt else I) % self.pr

    def _panel_rows(self, k):
        """tile rows that have a tile in column k below the diagonal tile (k,k): matrix rows,
        the residual row and -- when the inverse is carried along -- the identity rows already
        met by the pivots (block i becomes non-zero at column i)."""
        rows = list(range(k + 1, self.nt)) + [self.nt]
        if self.with_inverse:
            rows += [self.nt + 1 + i for i in range(k + 1)]
        return rows

    def _bcast(self, t, src, groups, key):
        if self.world == 1:
            return
        g, ranks = groups[key]
        if len(ranks) > 1:
            dist.broadcast(t, src=src, group=g)

    # -- assembly ---------------------------------------------------------------
    def assemble(self, variance, length_scales, noise, resid):
        """each rank builds the tiles it owns; resid = y - m(x) [n, dy] (replicated)."""
        ops, T = self.ops, self.T
        self.tiles = {}
        for I in range(self.nt):
            xi = self.X[I * T:I * T + self.rows_of(I)]
            for J in range(I + 1):
                if not self.mine(I, J):
                    continue
                t = ops.new_tile(self.rows_of(I), self.rows_of(J))
                if I == J:
                    ops.kernel_tile(self.kind, xi, None, variance, length_scales, noise, t)
                else:
                    ops.kernel_tile(self.kind, xi, self.X[J * T:J * T + self.rows_of(J)], variance, length_scales,
                                    None, t)
                self.tiles[(I, J)] = t
        for J in range(self.nt):   # residual tile row
            if self.mine(self.nt, J):
                t = ops.new_tile(self.dy, self.rows_of(J))
                t[:self.dy, :self.rows_of(J)] = resid[J * T:J * T + self.rows_of(J)].t()
                self.tiles[(self.nt, J)] = t
        if self.with_inverse:      # identity rows: block i = rows of I that start at column i
            for i in range(self.nt):
                R = self.nt + 1 + i
                for J in range(i, self.nt):
                    if self.mine(R, J):
                        t = ops.new_tile(self.rows_of(i), self.rows_of(J))
                        if J == i:
                            ops.set_identity(t, self.rows_of(i))
                        self.tiles[(R, J)] = t

    # -- factorisation ------------------------------------------------------------
    def _panel_phase(self, k, info_local):
        """steps 1-2 for tile column k: diagonal factor + broadcast down its process column,
        panel solves on my tiles of that column.  Returns the updated local info."""
        ops, nt, dev = self.ops, self.nt, self.X.device
        nk, ck = self.rows_of(k), k % self.pc
        if self.my_c != ck:
            return info_local
        diag_owner = self.owner(k, k)
        if self.rank == diag_owner:
            Lkk = self.tiles[(k, k)]
            winv, info = ops.potrf(Lkk, nk)
            bad = info.to(torch.int64)
            info_local = torch.where((info_local == 0) & (bad != 0), bad + k * self.T, info_local)
        else:
            Lkk = ops.new_tile(nk, nk)
            winv = torch.empty(self.ops.winv_numel(nk), dtype=torch.float64, device=dev)
        self._bcast(Lkk, diag_owner, self.col_groups, ck)
        self._bcast(winv, diag_owner, self.col_groups, ck)
        for I in self._panel_rows(k):
            if self.mine(I, k):
                ops.trsm(Lkk, winv, nk, self.tiles[(I, k)], self.rows_of(I))
        return info_local

    def _bcast_async(self, t, src, groups, key):
        """-> Work handle or None; the collective runs on the backend's own stream."""
        if self.world == 1:
            return None
        g, ranks = groups[key]
        if len(ranks) > 1:
            return dist.broadcast(t, src=src, group=g, async_op=True)
        return None

    def _start_rows(self, k):
        """step 3a, asynchronous: panel tile (I,k) along process row I mod Pr (left operand of
        tile row I).  Returns (left, works)."""
        ops, nt, nk = self.ops, self.nt, self.rows_of(k)
        left, works = {}, []
        for I in self._panel_rows(k):
            src = self.owner(I, k)
            rI = self._prow(I)
            if self.my_r == rI:
                t = self.tiles[(I, k)] if self.rank == src else ops.new_tile(self.rows_of(I), nk)
                w = self._bcast_async(t, src, self.row_groups, rI)
                if w is not None:
                    works.append(w)
                left[I] = t
        return left, works

    def _start_cols(self, k, left):
        """step 3b, asynchronous (after 3a has completed on this rank): P_I down process column
        I mod Pc (right operand of tile column I).  Returns (right, works)."""
        ops, nt, nk = self.ops, self.nt, self.rows_of(k)
        right, works = {}, []
        for I in range(k + 1, nt):
            cI = I % self.pc
            if self.my_c == cI:
                # the source is the member of process column cI that already holds P_I: (I mod Pr, cI)
                src = (I % self.pr) * self.pc + cI
                t = left[I] if self.rank == src else left.get(I)
                if t is None:
                    t = ops.new_tile(self.rows_of(I), nk)
                w = self._bcast_async(t, src, self.col_groups, cI)
                if w is not None:
                    works.append(w)
                right[I] = t
        return right, works

    @staticmethod
    def _wait(works):
        for w in works:
            w.wait()

    def _update(self, k, left, right, columns):
        """step 4 restricted to my tiles in the given tile columns."""
        nk = self.rows_of(k)
        for (I, J), t in self.tiles.items():
            # matrix / residual rows: tiles on or below the diagonal; identity rows: only those
            # the pivots have met (they are exactly the ones with a panel tile in `left`)
            if J > k and J in columns and I in left and (I >= J):
                self.ops.update(t, left[I], right[J], self.rows_of(I), self.rows_of(J), nk, lower=(I == J))

    def factor(self):
        """right-looking block-cyclic Cholesky carrying the residual row, with look-ahead: after
        the exchange of panel k only tile column k+1 is updated before ITS diagonal tile is
        factored, broadcast and its panel solved (the critical path of step k+1); the rest of
        the trailing update by panel k follows, so the other process columns never wait for the
        next panel.  Every tile still receives its updates in the order k = 0, 1, ... .
        Returns the global LAPACK-style info (0 = ok)."""
        nt = self.nt
        info_local = torch.zeros(1, dtype=torch.int64, device=self.X.device)
        info_local = self._panel_phase(0, info_local)
        left, works = self._start_rows(0)
        self._wait(works)
        right, works = self._start_cols(0, left)
        for k in range(nt):
            self._wait(works)                                   # panel k is everywhere it is needed
            self._update(k, left, right, {k + 1})
            rest = list(range(k + 2, nt))
            half = (len(rest) + 1) // 2
            if k + 1 < nt:
                info_local = self._panel_phase(k + 1, info_local)
                nleft, works = self._start_rows(k + 1)          # in flight under the first half ...
                self._update(k, left, right, set(rest[:half]))
                self._wait(works)
                nright, works = self._start_cols(k + 1, nleft)  # ... and under the second half
                self._update(k, left, right, set(rest[half:]))
                left, right = nleft, nright
            else:
                works = []
        if self.world > 1:
            dist.all_reduce(info_local, op=dist.ReduceOp.MAX, group=self.group)
        self.info = int(info_local.item())
        return self.info

    def lml(self):
        """LML of gpr.py:63-67 from the distributed factor (all-reduce of two scalars)."""
        ops = self.ops
        acc = torch.zeros(2, dtype=torch.float64, device=self.X.device)
        for k in range(self.nt):
            if self.mine(k, k):
                acc[0] += ops.log_diag_sum(self.tiles[(k, k)], self.rows_of(k))
            if self.mine(self.nt, k):
                acc[1] += ops.sumsq(self.tiles[(self.nt, k)], self.dy, self.rows_of(k))
        if self.world > 1:
            dist.all_reduce(acc, op=dist.ReduceOp.SUM, group=self.group)
        return -0.5 * acc[1] - self.dy * acc[0] - 0.5 * self.dy * self.n * math.log(2.0 * math.pi)

    # -- backward (closed form on the same grid) ---------------------------------
    def _kinv_tiles(self):
        """Kyy^-1 = U U^T (lower tiles, same owners as the matrix tiles) from U = L^-T, which the
        factorisation left in the identity rows: (Kyy^-1)_IJ = sum_{K >= I} U_IK U_JK^T.  Per tile
        column K of U the tiles U_IK (I <= K) travel exactly like a panel of the factorisation:
        along process row I mod Pr (left operands), then down process column I mod Pc (right)."""
        ops, nt = self.ops, self.nt
        kinv = {(I, J): ops.new_tile(self.rows_of(I), self.rows_of(J))
                for I in range(nt) for J in range(I + 1) if self.mine(I, J)}
        for K in range(nt):
            nk = self.rows_of(K)
            left, right = {}, {}
            for I in range(K + 1):
                R = nt + 1 + I
                src = self.owner(R, K)
                if self.my_r == I % self.pr:
                    t = self.tiles[(R, K)] if self.rank == src else ops.new_tile(self.rows_of(I), nk)
                    self._bcast(t, src, self.row_groups, I % self.pr)
                    left[I] = t
            for I in range(K + 1):
                cI = I % self.pc
                if self.my_c == cI:
                    src = (I % self.pr) * self.pc + cI
                    t = left[I] if self.rank == src else left.get(I)
                    if t is None:
                        t = ops.new_tile(self.rows_of(I), nk)
                    self._bcast(t, src, self.col_groups, cI)
                    right[I] = t
            for (I, J), t in kinv.items():
                if I <= K:
                    ops.update(t, left[I], right[J], self.rows_of(I), self.rows_of(J), nk, lower=(I == J), alpha=1.0)
        return kinv

    def backward(self, variance, length_scales):
        """-> tensor [2 + nls]: dLML/d(variance, length_scales..., noise) w.r.t. the CONSTRAINED
        values, after a factorisation that carried the identity rows (with_inverse).
        a = Kyy^-1 (y - m) = U alpha;  G = 1/2 (a a^T - dy Kyy^-1);  every rank contracts the G
        tiles it owns with dK/dtheta (re-computed from the points) and D + 2 scalars are
        all-reduced."""
        assert self.with_inverse, "factor with with_inverse=True first"
        ops, nt, T, dy, dev = self.ops, self.nt, self.T, self.dy, self.X.device
        nls = length_scales.numel()
        # alpha^T [dy, n], replicated
        alphaT = torch.zeros(dy, self.n, dtype=torch.float64, device=dev)
        for K in range(nt):
            if self.mine(nt, K):
                alphaT[:, K * T:K * T + self.rows_of(K)] = self.tiles[(nt, K)][:dy, :self.rows_of(K)]
        if self.world > 1:
            dist.all_reduce(alphaT, op=dist.ReduceOp.SUM, group=self.group)
        # a^T = alpha^T U^T: block I gets sum_{K >= I} alpha_K^T U_IK^T from the owners of U_IK
        aT = torch.zeros(dy, self.n, dtype=torch.float64, device=dev)
        for (R, K), t in self.tiles.items():
            if R <= nt:
                continue
            I = R - nt - 1
            ri, rk = self.rows_of(I), self.rows_of(K)
            At = ops.new_tile(dy, rk)
            At[:dy, :rk] = alphaT[:, K * T:K * T + rk]
            Ct = ops.new_tile(dy, ri)
            ops.update(Ct, At, t, dy, ri, rk, lower=False, alpha=1.0)
            aT[:, I * T:I * T + ri] += Ct[:dy, :ri]
        if self.world > 1:
            dist.all_reduce(aT, op=dist.ReduceOp.SUM, group=self.group)
        kinv = self._kinv_tiles()
        acc = torch.zeros(2 + nls, dtype=torch.float64, device=dev)
        kpad = _ops.round_up(dy, 16)
        for (I, J), Kt in kinv.items():
            ri, rj = self.rows_of(I), self.rows_of(J)
            aI = ops.new_tile(ri, kpad)
            aI[:ri, :dy] = aT[:, I * T:I * T + ri].t()
            aJ = ops.new_tile(rj, kpad)
            aJ[:rj, :dy] = aT[:, J * T:J * T + rj].t()
            Gt = ops.new_tile(ri, rj)
            ops.update(Gt, aI, aJ, ri, rj, kpad, lower=False, alpha=0.5)          # 1/2 a_I a_J^T
            Kd = Kt[:ri, :rj]
            if I == J:
                Kd = torch.tril(Kd) + torch.tril(Kd, -1).t()
            G = Gt[:ri, :rj] - 0.5 * dy * Kd
            if I == J:
                acc[1 + nls] += G.diagonal().sum()                                 # d/d noise = tr G
            else:
                G = 2.0 * G                                                        # symmetric partner (J, I)
            xi = self.X[I * T:I * T + ri]
            xj = self.X[J * T:J * T + rj]
            acc[:1 + nls] += ops.kernel_grad(self.kind, xi, xj, variance, length_scales, G.contiguous())
        if self.world > 1:
            dist.all_reduce(acc, op=dist.ReduceOp.SUM, group=self.group)
        return acc

    def log_likelihood_and_grad(self, variance, length_scales, noise, resid, max_tries=10):
        """(LML, [dLML/dvariance, dLML/dlength_scales..., dLML/dnoise]) with the factorisation
        carrying U = L^-T along (identity rows); same jitter ladder as log_likelihood."""
        self.with_inverse = True
        try:
            lml = self.log_likelihood(variance, length_scales, noise, resid, max_tries)
            return lml, self.backward(variance, length_scales)
        finally:
            self.with_inverse = False

    def log_likelihood(self, variance, length_scales, noise, resid, max_tries=10):
        """assemble + factor with the jitter ladder of functions.py:20-43 (decided on the
        all-reduced info, so every rank takes the same branch)."""
        self.assemble(variance, length_scales, noise, resid)
        if self.factor() == 0:
            return self.lml()
        for i in range(max_tries):
            self.assemble(variance, length_scales, noise + 10.0 ** (-max_tries + i), resid)
            if self.factor() == 0:
                return self.lml()
        raise RuntimeError("Max tries exceeded.")

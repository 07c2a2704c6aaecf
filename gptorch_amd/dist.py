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

    def update(self, C, A, B, m, n, k, lower):
        """C[:m,:n] -= A[:m,:k] B[:n,:k]^T (lower: only j <= i)."""
        _ops.gemm_nt(A, B, m, n, _ops.round_up(k, 16), alpha=-1.0, beta=1.0, C=C, lower=lower)

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

    # -- geometry ---------------------------------------------------------------
    def owner(self, I, J):
        return (I % self.pr) * self.pc + (J % self.pc)

    def mine(self, I, J):
        return self.owner(I, J) == self.rank

    def rows_of(self, I):
        """row count of tile row I (tile row nt = the residual rows)."""
        if I == self.nt:
            return self.dy
        return min(self.T, self.n - I * self.T)

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
        for I in list(range(k + 1, nt)) + [nt]:
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
        for I in list(range(k + 1, nt)) + [nt]:
            src = self.owner(I, k)
            rI = I % self.pr
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
            if J > k and I >= J and J in columns:
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

"""
2-D block-cyclic exact-GP log marginal likelihood (and its closed-form backward) over the GPUs
of one node (BASELINE.json config 4: N = 65536 across 8 x MI355X).  The
reference has no multi-GPU path at all (`gptorch/models/base.py:33` "Assume single GPU"); this
is the distributed form of `GPR.log_likelihood` (gpr.py:47-67) and of its autograd backward.

One process per GPU (`torch.distributed`, backend "nccl" = RCCL over xGMI).  The N x N Gram
matrix is cut into T x T tiles; tile (I, J), I >= J, lives on rank (I mod Pr) * Pc + (J mod Pc)
of a Pr x Pc process grid (Pr divides Pc: 1x1, 1x2, 2x2, 2x4).

Layout in HBM (one allocation per rank, made once and reused by every evaluation):

    Aloc [rows, ld]   row-major fp64, ld = (#my tile columns) * T
      rows  0 .. nrow_t*T          my tile rows I = r, r+Pr, ... stacked in ascending order
                                   (a ragged last tile still takes T rows, zero padded)
      rows  res_off .. +dy         the residual (y - m)^T  -- only in process row nt mod Pr;
                                   becomes alpha^T = (L^-1 (y-m))^T            ("extra rows")
      rows  id_off .. +nrow_t*T    (backward only) identity blocks i = r, r+Pr, ...: block i is
                                   I at tile column i and becomes U_i,: = (L^-T)_i,: from there on
      columns                      my tile columns J = c, c+Pc, ... side by side

With this order the rows that take part in panel k -- matrix tile rows I > k, the residual and
the identity blocks i <= k -- are ONE contiguous row range of Aloc on every rank, so each step
of the factorisation is a handful of large launches and packed collectives instead of one
launch and one message per tile:

  1. the owner of (k,k) factors it (gpn_potrf_lower) and sends  [L_kk | leaf inverses]  as ONE
     packed buffer down its process COLUMN (nothing to send when Pr = 1);
  2. every rank of that process column solves ALL its panel rows in one call
     (gpn_trsm_right_lt on the stacked rows);
  3. the solved rows are packed and sent along each process ROW (one broadcast per process row:
     every rank then holds the panel rows of its own tile rows = the left operands);
  4. the tiles P_J that multiply from the right in my tile columns J all live in process row
     c mod Pr (because Pr | Pc), so ONE packed broadcast down each process column delivers them;
  5. trailing update: per local tile column one contraction over the stacked rows below it
     (gpn_gemm_nt, K = T) plus a lower-only one for a diagonal tile.

Look-ahead: after the exchange of panel k only tile column k+1 is updated before
ITS diagonal tile is factored and its panel solved; the row broadcast of panel k+1 is then in
flight (async_op: RCCL runs it on its own stream) under the first half of the remaining
trailing update by panel k and the column broadcast under the second half.  Receive buffers
are allocated once (two sets, alternating by panel parity); nothing is allocated and the host
never waits inside `factor()` -- the only read-back is one small all-reduced vector at the end
(log-det partials, |alpha|^2 and the per-tile `info` words).

xGMI is a full mesh of point-to-point links, so the row and column broadcasts run on disjoint
links concurrently.  The backward runs on the same grid: U = L^-T falls out of the same panel
solves / updates (the stacked [A ; I] trick of the leaf kernel at tile scale), Kyy^-1 = U U^T is
accumulated with the tile columns of U travelling exactly like factorisation panels, and D + 2
gradient scalars are all-reduced.

The dense arithmetic is behind a small `TileOps` interface: `NativeTileOps` (the product) calls
libgpnative through `_ops`; the CPU test-suite injects a torch-CPU implementation to exercise
this orchestration under gloo with world_size 2 and 4 (tests/test_dist_gloo.py).

`phantom=(rank, world)`: timing aid for a 1-GPU box -- do the work of ONE rank of a larger grid
with the collectives skipped (results are meaningless, the launches and their sizes are real).
"""
import math

import torch
import torch.distributed as dist

from . import _ops

LEAF = _ops.LEAF


MESH_STAGES = 4            # pipeline depth of the scatter + all-gather form of the mesh broadcast
MESH_DIRECT_BYTES = 4 << 20  # below this a panel goes root -> every peer directly (one hop, latency-bound)


def mesh_plan(members, root, me, count, stages=MESH_STAGES, direct_below=None):
    """Point-to-point schedule of ONE broadcast of `count` elements from `root` inside the
    sub-communicator `members` (global ranks), as seen by rank `me`: a list of stages, each a list of
    ("send" | "recv", peer, offset, length) that are issued as one grouped call (ncclGroupStart/End,
    dist.batch_isend_irecv).  xGMI is a full mesh of point-to-point links, so every (sender, receiver)
    pair of a stage uses its own link ("do not route these through a ring"):

      * small panels (count < direct_below) or a single peer: the root sends the whole buffer to every
        peer directly -- one hop, q = len(members) - 1 links in parallel;
      * otherwise scatter + all-gather, pipelined: the buffer is cut into `stages` slices of q pieces;
        in stage t the root sends piece (t, i) to peer i while peer i forwards piece (t-1, i), which it
        received one stage earlier, to the q - 1 other peers.  Every link carries count / q elements
        (a ring carries `count` over each of its links), in stages + 1 grouped calls.

    The plan is a pure function of its arguments and identical on every member, so sends and receives
    match by construction (tests/test_dist_gloo.py::test_mesh_plan_is_consistent checks it exhaustively).
    The same function is restated in C in csrc/rccl_adapter.cpp (`mesh_bcast`)."""
    peers = [m for m in members if m != root]
    q = len(peers)
    if q == 0 or count <= 0:
        return []
    if direct_below is None:
        import os
        direct_below = int(os.environ.get("GPN_DIST_MESH_DIRECT_BYTES", MESH_DIRECT_BYTES)) // 8
    if q == 1 or count < direct_below:
        if me == root:
            return [[("send", p, 0, count) for p in peers]]
        return [[("recv", root, 0, count)]]
    S = max(1, min(int(stages), count // q))
    npieces = S * q
    base, extra = divmod(count, npieces)

    def piece(t, i):               # slice t, piece i -> (offset, length); the first `extra` pieces are one longer
        idx = t * q + i
        return idx * base + min(idx, extra), base + (1 if idx < extra else 0)

    plan = []
    for t in range(S + 1):
        ops = []
        if me == root:
            if t < S:
                ops += [("send", p, *piece(t, i)) for i, p in enumerate(peers)]
        else:
            i = peers.index(me)
            if t < S:
                ops.append(("recv", root, *piece(t, i)))
            if t >= 1:
                ops += [("send", p, *piece(t - 1, i)) for p in peers if p != me]
                ops += [("recv", p, *piece(t - 1, j)) for j, p in enumerate(peers) if p != me]
        plan.append([o for o in ops if o[3] > 0])     # (a stage may be empty for one member: count < number of pieces)
    return plan


def mesh_broadcast(t, src, me, members, group, stages=None, stats=None):
    """execute `mesh_plan` for the contiguous tensor `t` with torch.distributed point-to-point ops (one
    dist.batch_isend_irecv = one ncclGroupStart/End per stage).  -> list of Work still in flight (stream-ordered
    backends); [] when the transport completed on the host.  stats: object with sent_bytes / recv_bytes counters."""
    flat = t.view(-1)
    es = t.element_size()
    # a stage forwards what the stage before received: on a stream-ordered backend (RCCL) the grouped calls
    # are queued back to back on the communicator's stream; a host-side transport (gloo) has to complete a
    # stage before the next one may read its pieces
    stream_ordered = dist.get_backend(group) == "nccl"
    staged = t.is_cuda and not stream_ordered      # gloo moves host memory only (shared-GPU test mode): stage the pieces
    works = []
    for stage in mesh_plan(members, src, me, flat.numel(), MESH_STAGES if stages is None else stages):
        if not stage:
            continue
        ops, landed = [], []
        for kind, peer, off, ln in stage:
            view = flat[off:off + ln]
            if kind == "send":
                if stats is not None:
                    stats.sent_bytes[peer] = stats.sent_bytes.get(peer, 0) + ln * es
                ops.append(dist.P2POp(dist.isend, view.cpu() if staged else view, peer, group=group))
            else:
                if stats is not None:
                    stats.recv_bytes += ln * es
                host = torch.empty(ln, dtype=t.dtype) if staged else view
                if staged:
                    landed.append((view, host))
                ops.append(dist.P2POp(dist.irecv, host, peer, group=group))
        ws = dist.batch_isend_irecv(ops)
        if not stream_ordered:
            for w in ws:
                w.wait()
            for view, host in landed:
                view.copy_(host)
        else:
            works += ws
    return works


def choose_grid(world):
    """Pr x Pc with Pr <= Pc, as square as possible: 1x1, 1x2, 2x2, 2x4."""
    pr = int(math.sqrt(world))
    while world % pr:
        pr -= 1
    return pr, world // pr


def _flatten(works):
    for w in works:
        if isinstance(w, (list, tuple)):
            yield from _flatten(w)
        else:
            yield w


class NativeTileOps:
    """Tile arithmetic on libgpnative (fp64 tensors / strided 2-D views on this rank's GPU)."""

    def __init__(self, device):
        self.device = device

    def zeros(self, rows, cols):
        return torch.zeros(rows, cols, dtype=torch.float64, device=self.device)

    # `kind`: the name of a native stationary kernel (hyper-parameters: variance [1], ls [1 | d]) or an _expr.Program -- a Sum /
    # Product tree over native leaves (round 6: the reference's example model Linear + Rbf + Constant,
    # examples/regression_1d.py:34-53, on the grid), whose packed constrained parameters travel in `variance` (ls: empty)
    def kernel_block(self, kind, Xi, Xj, variance, ls, out):
        """out[:ri, :rj] <- K(Xi, Xj) (rectangular; out is a view with a leading dimension)."""
        if not isinstance(kind, str):
            from . import _expr
            _expr.kernel_matrix(kind, _ops._c(variance.detach()), Xi, Xj, out=out, ldk=out.stride(0))
            return
        _ops.kernel_matrix(kind, Xi, Xj, variance, ls, out=out, ldk=out.stride(0))

    def kernel_param_count(self, kind, variance, ls):
        return int(variance.numel()) if not isinstance(kind, str) else 1 + int(ls.numel())

    def winv_numel(self, n):
        return int(_ops._native.lib().gpn_winv_bytes(n)) // 8

    def potrf(self, A, n, e, winv, info):
        """in-place lower Cholesky of the tile A[:n,:n]; the e rows below it (panel rows of the enclosing
        matrix on this GPU) come out as R L^-T; nothing right of column n is touched; info: int32 view."""
        st = _ops._native.lib().gpn_potrf_lower_panel(_ops._stream(A.device), _ops._ptr(A), n, e, A.stride(0),
                                                      _ops._ptr(winv), _ops._ptr(info))
        _ops._native.check(st, "gpn_potrf_lower_panel")

    def trsm(self, L, winv, n, B, m):
        """B[:m,:n] <- B L^-T."""
        st = _ops._native.lib().gpn_trsm_right_lt(_ops._stream(B.device), _ops._ptr(L), n, L.stride(0),
                                                  _ops._ptr(winv), _ops._ptr(B), m, B.stride(0))
        _ops._native.check(st, "gpn_trsm_right_lt")

    def update(self, C, A, B, m, n, k, lower, alpha=-1.0, beta=1.0):
        """C[:m,:n] = alpha A[:m,:k] B[:n,:k]^T + beta C (lower = True: only j <= i; lower = 2, m >= n: only j <= i
        inside the top n x n square, the rows below it whole)."""
        _ops.gemm_nt(A, B, m, n, _ops.round_up(k, 16), alpha=alpha, beta=beta, C=C, lower=lower)

    def update_stair(self, C, A, B, m, nb, blk, k, step, diag, alpha=-1.0):
        """C[:m, :nb*blk] += alpha A[:m,:k] B[:nb*blk,:k]^T restricted to the staircase: column block b has the rows
        from b*step on; diag: its first blk x blk square is lower-only (gpn_gemm_nt_stair)."""
        _ops.gemm_nt_stair(A, B, C, m, nb, blk, _ops.round_up(k, 16), step, diag, alpha=alpha)

    def copy(self, dst, src, rows, cols):
        """dst[:rows,:cols] <- src[:rows,:cols] (both strided row-major views)."""
        st = _ops._native.lib().gpn_copy_matrix(_ops._stream(dst.device), _ops._ptr(src), rows, cols, src.stride(0),
                                                _ops._ptr(dst), dst.stride(0), 0)
        _ops._native.check(st, "gpn_copy_matrix")

    def kernel_grad(self, kind, Xi, Xj, variance, ls, G):
        """-> tensor [1 + nls]: sum G * dK(Xi, Xj)/d(variance, length_scales) (gpn_kernel_grad)."""
        if not isinstance(kind, str):
            from . import _expr
            outs, _ = _expr._sweeps(kind, _ops._c(variance.detach()), Xi, Xj, G, G.stride(0))
            return _expr.flat_grad(kind, outs)
        from . import _backward
        gv, gl = _backward.kernel_backward(kind, Xi, Xj, variance, ls, G)
        return torch.cat([gv, gl])

    def log_diag_sum(self, A, n):
        out = torch.empty(3, dtype=torch.float64, device=A.device)
        st = _ops._native.lib().gpn_lml_reduce(_ops._stream(A.device), _ops._ptr(A), n, 0, A.stride(0), _ops._ptr(out))
        _ops._native.check(st, "gpn_lml_reduce")
        return out[0]

    def sumsq(self, A, m, n):
        return _ops.row_sumsq(A, m, n).sum()

    def row_sumsq(self, A, m, n):
        """-> [m]: sum_c A[r, c]^2 over the first n columns."""
        return _ops.row_sumsq(A, m, n)

    # -- the refinement step of the quadratic form, in pieces (BlockCyclicGP._refine; csrc/refine.hip) ------------
    def tile_inverse(self, L, n):
        """-> W = L[:n,:n]^-1 (lower triangular, row-major) in a zero-padded [round_up(n,128)]^2 buffer: leaf inverses + the
        level-parallel triangular inversion of _backward._upper_inverse + one transpose, for ONE diagonal tile.  (Row-major W
        makes a = L^-T s a column-sum product: gemv_t_acc(W, n, n, s, a).)"""
        lib = _ops._native.lib()
        npad = _ops.round_up(max(n, 1), LEAF)
        U = _ops.zeros(npad, npad, L.device)
        S = _ops.zeros(npad, npad, L.device)
        winv = torch.empty(max(1, self.winv_numel(n)), dtype=torch.float64, device=L.device)
        info = torch.zeros(1, dtype=torch.int32, device=L.device)
        st = lib.gpn_trtri_diag(_ops._stream(L.device), _ops._ptr(L), n, L.stride(0), _ops._ptr(winv), _ops._ptr(info))
        _ops._native.check(st, "gpn_trtri_diag")
        if n > 2 * LEAF:
            st = lib.gpn_trtri_upper_ws(_ops._stream(L.device), _ops._ptr(L), n, L.stride(0), _ops._ptr(winv), _ops._ptr(U), npad,
                                        _ops._ptr(S), npad)
        else:
            st = lib.gpn_trtri_upper(_ops._stream(L.device), _ops._ptr(L), n, L.stride(0), _ops._ptr(winv), _ops._ptr(U), npad)
        _ops._native.check(st, "gpn_trtri_upper")
        st = lib.gpn_transpose(_ops._stream(L.device), _ops._ptr(U), npad, npad, npad, _ops._ptr(S), npad)      # S <- U^T = L^-1
        _ops._native.check(st, "gpn_transpose")
        U.record_stream(torch.cuda.current_stream(L.device))
        return S

    def gemv_t_acc(self, L, rows, cols, a, c):
        """c[:, :cols] += a[:, :rows] @ L[:rows, :cols]  (a, c: [dy, ld] row-major, fixed summation order)."""
        lib = _ops._native.lib()
        dy = a.shape[0]
        need = max(1, int(lib.gpn_gemv_t_work_bytes(rows, cols, dy)) // 8)
        work = getattr(self, "_gemv_work", None)
        if work is None or work.numel() < need or work.device != a.device:
            work = self._gemv_work = torch.empty(need, dtype=torch.float64, device=a.device)
        st = lib.gpn_gemv_t_acc(_ops._stream(a.device), _ops._ptr(L), L.stride(0), rows, cols, _ops._ptr(a), a.stride(0), dy,
                                _ops._ptr(c), c.stride(0), _ops._ptr(work))
        _ops._native.check(st, "gpn_gemv_t_acc")

    def refine_tile_count(self, n):
        return int(_ops._native.lib().gpn_refine_tile_count(n))

    def resid_part(self, kind, X, variance, ls, noise, a, q0, q1):
        """-> ka [dy, round_up(n, 128), 2]: (hi, lo) of the rows of Kyy a restricted to the lower 64 x 64 tiles q0 <= q < q1
        (and their mirror entries), Kyy re-computed from X; a: [dy, round_up(n, 128)] contiguous."""
        lib = _ops._native.lib()
        n, d = X.shape
        dy, lds = a.shape
        ka = torch.empty(dy, lds, 2, dtype=torch.float64, device=X.device)
        work = torch.empty(max(1, int(lib.gpn_refine_resid_part_work_bytes(dy, q1 - q0)) // 8), dtype=torch.float64, device=X.device)
        if not isinstance(kind, str):
            theta, nz = _ops._c(variance.detach()), _ops._c(noise.detach()).reshape(-1)[:1].contiguous()
            st = lib.gpn_refine_resid_part_expr(_ops._stream(X.device), kind.terms, len(kind.instances), kind.gstart, kind.ngroups, _ops._ptr(theta),
                                                _ops._ptr(_ops._c(X)), n, d, _ops._ptr(nz), _ops._ptr(a), dy, q0, q1, _ops._ptr(work), _ops._ptr(ka))
            _ops._native.check(st, "gpn_refine_resid_part_expr")
            ka[:, n:].zero_()
            return ka
        var, l, nz = _ops._c(variance.detach()), _ops._c(ls.detach()), _ops._c(noise.detach())
        st = lib.gpn_refine_resid_part(_ops._stream(X.device), _ops.KINDS[kind], _ops._ptr(_ops._c(X)), n, d, _ops._ptr(var), _ops._ptr(l),
                                       l.numel(), _ops._ptr(nz), _ops._ptr(a), dy, q0, q1, _ops._ptr(work), _ops._ptr(ka))
        _ops._native.check(st, "gpn_refine_resid_part")
        ka[:, n:].zero_()
        return ka

    def refine_finish(self, R, a, ka):
        """-> tensor []: y^T a + a^T (y - Kyy a) with y = R [n, dy], a [dy, lds], ka from resid_part (all shares summed)."""
        n, dy = R.shape
        out = torch.zeros(3, dtype=torch.float64, device=R.device)
        st = _ops._native.lib().gpn_refine_finish(_ops._stream(R.device), _ops._ptr(_ops._c(R)), None, _ops._ptr(a), _ops._ptr(ka), n, dy,
                                                  _ops._ptr(out))
        _ops._native.check(st, "gpn_refine_finish")
        return out[1]


class BlockCyclicGP:
    """Distributed LML for a stationary kernel.  All ranks call every method collectively."""

    def __init__(self, X, Y, kind, tile=2048, grid=None, ops=None, group=None, phantom=None, force_comm=False, share=None,
                 schedule=None):
        """force_comm: issue the row / column collectives even where a sub-communicator has a single
        member (world 1, or Pr = 1) -- a test hook that drives the RCCL calls on a 1-GPU box.
        schedule: how a panel travels inside a row / column sub-communicator -- "bcast" = the
        backend's broadcast (RCCL picks the route: a ring / tree), "mesh" = grouped point-to-point
        sends over the direct xGMI links (`mesh_plan`); default from GPN_DIST_SCHEDULE, else "bcast".
        Both move the same bytes into the same buffers: results are bit-identical."""
        import os
        self.schedule = schedule or os.environ.get("GPN_DIST_SCHEDULE", "bcast")
        if self.schedule not in ("bcast", "mesh"):
            raise ValueError("schedule must be 'bcast' or 'mesh'")
        self.mesh_stages = int(os.environ.get("GPN_DIST_MESH_STAGES", MESH_STAGES))
        self.comm_timing = False       # record an event pair around every wait on a collective (exposed_comm_ms)
        self.refine = None             # refinement step of the quadratic form after the factorisation: None = from refine_min_n() rows on
        self.refined = False
        self._wait_events = []
        self.sent_bytes = {}           # global peer rank -> payload bytes this rank sent it ("mesh"; reset_comm_stats)
        self.bcast_root_bytes = 0      # payload bytes this rank was the root of ("bcast": the route is the backend's)
        self.recv_bytes = 0            # payload bytes this rank received in row / column exchanges
        live = dist.is_available() and dist.is_initialized() and phantom is None
        self.rank = dist.get_rank(group) if live else 0
        self.world = dist.get_world_size(group) if live else 1
        self.comm = live and (self.world > 1 or force_comm)
        if phantom is not None:
            self.rank, self.world = phantom
        self.pr, self.pc = grid if grid is not None else choose_grid(self.world)
        if self.pr * self.pc != self.world or self.pc % self.pr:
            raise ValueError("process grid %dx%d does not fit %d ranks (Pr must divide Pc)" % (self.pr, self.pc, self.world))
        self.my_r, self.my_c = divmod(self.rank, self.pc)
        self.X, self.Y, self.kind = X, Y, kind
        self.n, self.dy = Y.shape
        self.T = int(tile)
        if self.T % LEAF:
            raise ValueError("tile must be a multiple of %d" % LEAF)
        self.nt = (self.n + self.T - 1) // self.T
        self.ops = ops if ops is not None else NativeTileOps(X.device)
        self.group = group
        # sub-communicators: one per process row and per process column (created collectively)
        self.row_group = self.col_group = None
        self.xrow = self.comm and (self.pc > 1 or force_comm)      # panels travel along process rows
        self.xcol = self.comm and (self.pr > 1 or force_comm)      # ... and down process columns
        self.row_ranks = [self.my_r * self.pc + c for c in range(self.pc)]       # global ranks of my process row ...
        self.col_ranks = [r * self.pc + self.my_c for r in range(self.pr)]       # ... and of my process column
        if share is not None:      # a second engine on the same grid (predict): reuse the sub-communicators
            self.row_group, self.col_group = share.row_group, share.col_group
        elif self.comm:
            for r in range(self.pr):
                ranks = [r * self.pc + c for c in range(self.pc)]
                g = dist.new_group(ranks) if self.xrow else None
                if r == self.my_r:
                    self.row_group = g
            for c in range(self.pc):
                ranks = [r * self.pc + c for r in range(self.pr)]
                g = dist.new_group(ranks) if self.xcol else None
                if c == self.my_c:
                    self.col_group = g
            if dist.get_backend(group) == "nccl":
                # RCCL builds a sub-communicator at its first collective, with EVERY member in the call.  The mesh
                # schedule's grouped point-to-point stages do not involve every member every time, so make the
                # communicators exist first (row groups are disjoint, so are column groups: no ordering hazard)
                tok = torch.zeros(1, device=X.device)
                for g in (self.row_group, self.col_group):
                    if g is not None:
                        dist.all_reduce(tok, group=g)
        # -- local geometry -------------------------------------------------------
        T, nt, r, c = self.T, self.nt, self.my_r, self.my_c
        self.nrow_t = len(range(r, nt, self.pr))
        self.ncol_t = len(range(c, nt, self.pc))
        self.has_res = (nt % self.pr) == r
        self.res_off = self.nrow_t * T
        self.id_off = self.res_off + (_ops.round_up(self.dy, LEAF) if self.has_res else 0)
        self.ld = max(self.ncol_t, 1) * T
        dev = X.device
        # X rows in my local row / column order (the ragged tile is the last one in both orders,
        # so the real rows are a prefix)
        ridx = torch.cat([torch.arange(I * T, I * T + self.rows_of(I)) for I in range(r, nt, self.pr)] or
                         [torch.zeros(0, dtype=torch.long)]).to(dev)
        cidx = torch.cat([torch.arange(J * T, J * T + self.rows_of(J)) for J in range(c, nt, self.pc)] or
                         [torch.zeros(0, dtype=torch.long)]).to(dev)
        self.ridx, self.cidx = ridx, cidx
        self.Xrow = X.index_select(0, ridx).contiguous()
        self.Xcol = X.index_select(0, cidx).contiguous()
        self.A = None                  # allocated on first use (size depends on with_inverse)
        self.kinv = None
        self._alloc_inverse = False
        self.info = 0
        self.lml_rows = self.dy        # leading rows of the residual segment that enter |alpha|^2 (the rest: predict)
        self.last_a = None             # a^T = (Kyy^-1 (y - m))^T [dy, n], replicated, after backward()
        self.with_inverse = False      # carry I through the factorisation (-> U = L^-T) for the backward
        self._dirty = False            # a failed factorisation may have left non-finite padding

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

    def _rows_le(self, k, r=None):
        """how many of process row r's tile rows have index <= k."""
        r = self.my_r if r is None else r
        return 0 if k < r else min((k - r) // self.pr + 1, len(range(r, self.nt, self.pr)))

    def _cols_le(self, k):
        c = self.my_c
        return 0 if k < c else min((k - c) // self.pc + 1, self.ncol_t)

    def _active(self, k):
        """[lo, hi): my local rows that take part in panel k (matrix tile rows > k, the residual,
        identity blocks <= k) -- contiguous by construction of the layout."""
        lo = self._rows_le(k) * self.T
        if self.with_inverse:
            hi = self.id_off + self._rows_le(k) * self.T
        else:
            hi = self.res_off + (self.dy if self.has_res else 0)
        return lo, max(hi, lo)

    def local_shape(self):
        return tuple(self.A.shape) if self.A is not None else None

    # -- buffers ------------------------------------------------------------------
    def _ensure_buffers(self):
        T, ops = self.T, self.ops
        if self.A is not None and (self._alloc_inverse or not self.with_inverse):
            return
        self._alloc_inverse = self.with_inverse
        rows = self.id_off + (self.nrow_t * T if self.with_inverse else 0) + LEAF
        self.A = ops.zeros(rows, self.ld)
        # panel operands: left = the panel rows of my tile rows, right = the panel tiles of my tile
        # columns; two sets, alternating by panel parity (one is read by the trailing update while
        # the next panel is being received into the other)
        lrows = (rows + T - 1) // T * T
        self.left = [ops.zeros(lrows + 16, T) for _ in range(2)]
        self.right = [ops.zeros(max(self.ncol_t, 1) * T + 16, T) for _ in range(2)]
        self.wn = ops.winv_numel(T)
        self.diag = ops.zeros(T * T + self.wn, 1).view(-1)        # [L_kk | leaf inverses], packed
        self.winv = ops.zeros(self.wn, 1).view(-1)
        self.stats = ops.zeros(self.nt + 2, 1).view(-1)           # log-det partial, |alpha|^2, info per tile
        self.info_t = torch.zeros(self.nt, dtype=torch.int32, device=self.X.device)
        self._dirty = False

    # -- collectives ----------------------------------------------------------------
    def _bcast(self, t, src, group, async_op=False):
        """broadcast a contiguous tensor inside a row / column sub-communicator; -> Work, list of Work or None.
        schedule "mesh": the grouped point-to-point plan of `mesh_plan` (one dist.batch_isend_irecv per stage)."""
        if not self.comm or group is None:
            return None
        nbytes = t.numel() * t.element_size()
        members = self.row_ranks if group is self.row_group else self.col_ranks
        if self.schedule == "bcast" or len(members) == 1:
            if src == self.rank:
                self.bcast_root_bytes += nbytes * (len(members) - 1)
            else:
                self.recv_bytes += nbytes
            w = dist.broadcast(t, src=src, group=group, async_op=async_op)
            return w if async_op else None
        works = mesh_broadcast(t, src, self.rank, members, group, self.mesh_stages, self)
        if async_op:
            return works
        for w in works:
            w.wait()
        return None

    def _wait(self, works):
        """make the compute stream wait for the collectives in `works`; with comm_timing an event pair around the
        wait measures how long the stream stood still for them (the exposed part of the exchange)."""
        works = [w for w in _flatten(works) if w is not None]
        if not works:
            return
        if self.comm_timing and self.X.is_cuda:
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for w in works:
                w.wait()
            e1.record()
            self._wait_events.append((e0, e1, getattr(self, "_panel_tag", -1)))
        else:
            for w in works:
                w.wait()

    def reset_comm_stats(self):
        self._wait_events, self.sent_bytes, self.bcast_root_bytes, self.recv_bytes = [], {}, 0, 0

    def comm_stats(self):
        """-> dict: exposed_comm_ms (sum over the recorded waits; synchronises), bytes sent per peer, bytes received."""
        ms = 0.0
        per_panel = {}
        if self._wait_events:
            torch.cuda.synchronize()
            for a, b, tag in self._wait_events:
                dt = a.elapsed_time(b)
                ms += dt
                per_panel[tag] = per_panel.get(tag, 0.0) + dt
        # exposed_ms_per_panel[k]: how long this rank's compute stream stood still for the exchanges of panel k of the
        # factorisation (its diagonal tile's broadcast, its rows, its columns); -1 = waits outside the factorisation
        return {"exposed_comm_ms": ms, "waits": len(self._wait_events), "sent_bytes_per_peer": dict(self.sent_bytes),
                "bcast_root_bytes": self.bcast_root_bytes, "recv_bytes": self.recv_bytes,
                "exposed_ms_per_panel": [per_panel.get(k, 0.0) for k in range(self.nt)], "exposed_ms_outside_panels": per_panel.get(-1, 0.0)}

    # -- assembly ---------------------------------------------------------------
    def assemble(self, variance, length_scales, noise, resid):
        """each rank builds the tiles it owns -- one rectangular K(X_rows, X_J) per local tile
        column over all my tile rows at or below the diagonal; resid = y - m(x) [n, dy]."""
        ops, T, nt = self.ops, self.T, self.nt
        self._ensure_buffers()
        if self._dirty:
            self.A.zero_()
            for b in self.left + self.right:
                b.zero_()
            self._dirty = False
        A = self.A
        nreal = self.Xrow.shape[0]
        for lj, J in enumerate(range(self.my_c, nt, self.pc)):
            nJ = self.rows_of(J)
            li0 = self._rows_le(J - 1)                 # first of my tile rows with I >= J
            r0 = li0 * T
            if r0 >= nreal:
                continue
            xj = self.Xcol[lj * T:lj * T + nJ]
            ops.kernel_block(self.kind, self.Xrow[r0:nreal], xj, variance, length_scales, A[r0:nreal, lj * T:lj * T + nJ])
            if li0 * self.pr + self.my_r == J:         # the diagonal tile is mine: + noise I (gpr.py:83-86)
                A[r0:r0 + nJ, lj * T:lj * T + nJ].diagonal().add_(noise.reshape(()))
        if self.has_res and self.ncol_t:
            A[self.res_off:self.res_off + self.dy, :self.cidx.numel()] = resid.index_select(0, self.cidx).t()
        if self.with_inverse:
            A[self.id_off:self.id_off + self.nrow_t * T].zero_()
            for li, i in enumerate(range(self.my_r, nt, self.pr)):
                if i % self.pc == self.my_c:
                    lj = (i - self.my_c) // self.pc
                    ni = self.rows_of(i)
                    A[self.id_off + li * T:self.id_off + li * T + ni, lj * T:lj * T + ni].diagonal().fill_(1.0)

    # -- factorisation ------------------------------------------------------------
    def _panel_phase(self, k):
        """steps 1-2 for tile column k: diagonal factor (+ packed broadcast down its process
        column), panel solve of all my rows of that column in one call."""
        ops, T = self.ops, self.T
        nk, ck = self.rows_of(k), k % self.pc
        if self.my_c != ck:
            return
        lk = (k - self.my_c) // self.pc
        lo, hi = self._active(k)
        m = hi - lo
        colk = self.A[:, lk * T:lk * T + nk]
        diag_mine = (k % self.pr) == self.my_r
        exchange = self.xcol
        Lp, Wp = self.diag[:T * T].view(T, T), self.diag[T * T:]
        L = Lp
        if diag_mine:
            d0 = (self._rows_le(k) - 1) * T
            L = colk[d0:]
            if not exchange:
                # nobody else needs L_kk (Pr = 1): my panel rows ride along as extra rows of the SAME call, so the
                # tile's leaf chain runs underneath their solves / updates instead of alone on the chip
                self.ops.potrf(L, nk, hi - (d0 + nk), Wp, self.info_t[k:k + 1])
                return
            self.ops.potrf(L, nk, 0, Wp, self.info_t[k:k + 1])
            ops.copy(Lp, L, nk, nk)
            L = Lp
        if exchange:
            self._wait([self._bcast(self.diag, (k % self.pr) * self.pc + ck, self.col_group, async_op=True)])
        if m:
            ops.trsm(L, Wp, nk, colk[lo:], m)

    def _start_rows(self, k, matrix=None):
        """step 3, asynchronous: the solved panel rows of process row r, packed, from their owner
        (r, k mod Pc) to the whole process row.  -> (left buffer, [work])."""
        ops, T = self.ops, self.T
        nk, ck = self.rows_of(k), k % self.pc
        lo, hi = self._active(k) if matrix is None else matrix
        m = hi - lo
        buf = self.left[k & 1]
        if m == 0:
            return buf, []
        if nk < T:
            buf[:, nk:].zero_()                                   # K padding of the ragged last panel
        if self.my_c == ck:
            lk = (k - self.my_c) // self.pc
            ops.copy(buf, self.A[lo:hi, lk * T:lk * T + nk], m, nk)
        w = self._bcast(buf.view(-1)[:m * T], self.my_r * self.pc + ck, self.row_group, async_op=True)
        return buf, [w]

    def _start_cols(self, k, left, first_tile=None, count=None):
        """step 4, asynchronous (after step 3 has completed on this rank): the panel tiles P_J of
        my tile columns J > k.  They all sit in process row c mod Pr (Pr | Pc), in that rank's
        left buffer, every Pc/Pr-th tile: gathered into one packed buffer and broadcast down the
        process column.  -> (right operand [count*T, T], [work])."""
        T = self.T
        rs = self.my_c % self.pr                                  # process row that holds my P_J
        if first_tile is None:
            lj0 = self._cols_le(k)                                # my first tile column > k
            count = self.ncol_t - lj0
            J0 = lj0 * self.pc + self.my_c
            first_tile = (J0 - rs) // self.pr - self._rows_le(k, rs)   # its slot in rs's left buffer
        if count <= 0:
            return self.right[k & 1], []
        step = self.pc // self.pr
        i_am_src = self.my_r == rs
        if i_am_src and step == 1:
            buf = left[first_tile * T:]                           # already contiguous: no gather
        else:
            buf = self.right[k & 1]
            if i_am_src:
                nslot = (left.shape[0] - 16) // T
                src = left[:nslot * T].view(nslot, T, T)[first_tile::step][:count]
                buf[:count * T].view(count, T, T).copy_(src)
        w = self._bcast(buf.view(-1)[:count * T * T], rs * self.pc + self.my_c, self.col_group, async_op=True) \
            if self.xcol else None
        return buf, [w]

    def _update(self, k, left, right, lj_from, lj_to):
        """step 5 restricted to my local tile columns [lj_from, lj_to): ONE staircase launch -- every tile column
        starts Pc/Pr tile rows below its left neighbour (my first tile row at or below the diagonal), and where the
        diagonal tiles are mine (c mod Pr = r) each column's first tile is lower-only."""
        T = self.T
        nk = self.rows_of(k)
        lo, hi = self._active(k)
        base = self._cols_le(k)
        a, b = max(lj_from, base), min(lj_to, self.ncol_t)
        if b <= a:
            return
        J = a * self.pc + self.my_c
        r0 = self._rows_le(J - 1) * T                    # >= lo: J > k
        if r0 >= hi:
            return
        diag = (self.my_c % self.pr) == self.my_r
        self.ops.update_stair(self.A[r0:, a * T:], left[r0 - lo:], right[(a - base) * T:], hi - r0, b - a, T, nk,
                              (self.pc // self.pr) * T, diag)

    def factor(self):
        """right-looking block-cyclic Cholesky carrying the residual row, with look-ahead (module
        docstring).  Every tile still receives its updates in the order k = 0, 1, ... .
        Returns the global LAPACK-style info (0 = ok)."""
        nt = self.nt
        self.info_t.zero_()
        self._panel_tag = 0                                       # (comm_timing: which panel's exchange a wait belongs to)
        self._panel_phase(0)
        left = right = None
        works = []
        if nt > 1:
            left, works = self._start_rows(0)
            self._wait(works)
            right, works = self._start_cols(0, left)
        for k in range(nt - 1):
            self._panel_tag = k
            self._wait(works)                                     # panel k is everywhere it is needed
            works = []
            self._panel_tag = k + 1
            nxt = self._cols_le(k + 1) - 1 if (k + 1) % self.pc == self.my_c else -1
            if nxt >= 0:
                self._update(k, left, right, nxt, nxt + 1)        # tile column k+1 first ...
            self._panel_phase(k + 1)                              # ... so only ITS panel is on the critical path
            if k + 2 < nt:
                nleft, works = self._start_rows(k + 1)            # in flight under the first half ...
                a = self._cols_le(k + 1)
                half = a + (self.ncol_t - a + 1) // 2
                self._update(k, left, right, a, half)
                self._wait(works)
                nright, works = self._start_cols(k + 1, nleft)    # ... and under the second half
                self._update(k, left, right, half, self.ncol_t)
                left, right = nleft, nright
        self._panel_tag = -1
        return self._finish()

    def _finish(self):
        """one small all-reduce: log-det partials, |alpha|^2 and every tile's info word."""
        ops, T, nt = self.ops, self.T, self.nt
        st = self.stats
        st.zero_()
        for lj, J in enumerate(range(self.my_c, nt, self.pc)):
            if J % self.pr == self.my_r:
                li = (J - self.my_r) // self.pr
                st[0] += ops.log_diag_sum(self.A[li * T:, lj * T:], self.rows_of(J))
        if self.has_res and self.ncol_t:
            st[1] += ops.sumsq(self.A[self.res_off:], self.lml_rows, self.cidx.numel())
        st[2:] += self.info_t.to(torch.float64)
        if self.comm:
            dist.all_reduce(st, op=dist.ReduceOp.SUM, group=self.group)
        host = st.cpu()
        self._logdet, self._sumsq = float(host[0]), float(host[1])
        if bool((host[2:] < 0).any()):
            raise _ops.NativeError("a tile factorisation reported an internal status (not a property of the matrix)")
        bad = torch.nonzero(host[2:] != 0)
        self.info = 0 if bad.numel() == 0 else int(bad[0, 0]) * T + int(host[2 + int(bad[0, 0])])
        if self.info:
            self._dirty = True
        return self.info

    def lml(self):
        """LML of gpr.py:63-67 from the distributed factor (the two sums were all-reduced by factor())."""
        p = self.lml_rows
        v = -0.5 * self._sumsq - p * self._logdet - 0.5 * p * self.n * math.log(2.0 * math.pi)
        return torch.tensor(v, dtype=torch.float64, device=self.X.device)

    # -- backward (closed form on the same grid) ---------------------------------
    def _kinv_local(self):
        """Kyy^-1 = U U^T into self.kinv (same layout as the matrix tiles; diagonal tiles full)
        from U = L^-T, which the factorisation left in the identity rows:
        (Kyy^-1)_IJ = sum_{K >= I} U_IK U_JK^T.  Per tile column K of U the blocks U_IK (I <= K)
        travel exactly like a panel of the factorisation: packed along each process row (left
        operands), then the tiles of my tile columns down the process column (right operands)."""
        ops, T, nt = self.ops, self.T, self.nt
        if self.kinv is None:
            self.kinv = ops.zeros(self.nrow_t * T + LEAF, self.ld)
        else:
            self.kinv.zero_()
        C = self.kinv
        rs = self.my_c % self.pr

        def start(K):
            nid = self._rows_le(K)                                 # my identity blocks i <= K
            left, works = self._start_rows(K, matrix=(self.id_off, self.id_off + nid * T))
            return left, works

        left, works = start(0)
        for K in range(nt):
            nK = self.rows_of(K)
            self._wait(works)
            ncol = self._cols_le(K)                                # my tile columns J <= K, first at slot ...
            first = (self.my_c - rs) // self.pr                    # ... of tile J = c in process row rs's buffer
            right, works = self._start_cols(K, left, first_tile=first, count=ncol)
            self._wait(works)
            nleft, works = start(K + 1) if K + 1 < nt else (None, [])
            hi = self._rows_le(K) * T                              # rows I <= K of the result
            r0 = self._rows_le(self.my_c - 1) * T                  # my first tile column starts here, the next Pc/Pr tiles lower
            if ncol and r0 < hi:
                ops.update_stair(C[r0:], left[r0:], right, hi - r0, ncol, T, nK, (self.pc // self.pr) * T, False, alpha=1.0)
            left = nleft
        return C

    def backward(self, variance, length_scales):
        """-> tensor [2 + nls]: dLML/d(variance, length_scales..., noise) w.r.t. the CONSTRAINED
        values, after a factorisation that carried the identity rows (with_inverse).
        a = Kyy^-1 (y - m) = U alpha;  G = 1/2 (a a^T - dy Kyy^-1);  every rank contracts the G
        tiles it owns with dK/dtheta (re-computed from the points) and D + 2 scalars are
        all-reduced."""
        assert self.with_inverse, "factor with with_inverse=True first"
        ops, nt, T, dy, dev = self.ops, self.nt, self.T, self.dy, self.X.device
        count = getattr(ops, "kernel_param_count", None)
        nls = (count(self.kind, variance, length_scales) if count else 1 + length_scales.numel()) - 1      # kernel parameters - 1
        A = self.A
        ncr = self.cidx.numel()
        # alpha^T [dy, n], replicated
        alphaT = torch.zeros(dy, self.n, dtype=torch.float64, device=dev)
        if self.has_res and ncr:
            alphaT[:, self.cidx] = A[self.res_off:self.res_off + dy, :ncr]
        if self.comm:
            dist.all_reduce(alphaT, op=dist.ReduceOp.SUM, group=self.group)
        # a^T = alpha^T U^T: my identity rows x my tile columns give a partial sum
        aT = torch.zeros(dy, self.n, dtype=torch.float64, device=dev)
        nrr = self.ridx.numel()
        if nrr and ncr:
            al = ops.zeros(_ops.round_up(dy, 16), self.ld)
            al[:dy, :ncr] = alphaT.index_select(1, self.cidx)
            part = ops.zeros(_ops.round_up(dy, 16), _ops.round_up(nrr, 16))
            ops.update(part, al, A[self.id_off:], dy, nrr, self.ld, lower=False, alpha=1.0, beta=0.0)
            aT[:, self.ridx] = part[:dy, :nrr]
        if self.comm:
            dist.all_reduce(aT, op=dist.ReduceOp.SUM, group=self.group)
        self.last_a = aT
        C = self._kinv_local()
        acc = torch.zeros(2 + nls, dtype=torch.float64, device=dev)
        kpad = _ops.round_up(dy, 16)
        arow = ops.zeros(_ops.round_up(max(nrr, 1), 16) + 16, kpad)      # a for my tile rows / columns, K-padded
        acol = ops.zeros(_ops.round_up(max(ncr, 1), 16) + 16, kpad)
        if nrr:
            arow[:nrr, :dy] = aT.index_select(1, self.ridx).t()
        if ncr:
            acol[:ncr, :dy] = aT.index_select(1, self.cidx).t()
        for lj, J in enumerate(range(self.my_c, nt, self.pc)):
            nJ = self.rows_of(J)
            li0 = self._rows_le(J - 1)
            r0 = li0 * T
            if r0 >= nrr:
                continue
            xj = self.Xcol[lj * T:lj * T + nJ]
            # G = 1/2 a_I a_J^T - dy/2 Kinv_IJ, in place over the stacked rows of this tile column
            ops.update(C[r0:, lj * T:], arow[r0:], acol[lj * T:], nrr - r0, nJ, kpad, lower=False, alpha=0.5, beta=-0.5 * dy)
            if li0 * self.pr + self.my_r == J:
                G = C[r0:r0 + nJ, lj * T:lj * T + nJ]
                acc[1 + nls] += G.diagonal().sum()                                 # d/d noise = tr G
                acc[:1 + nls] += ops.kernel_grad(self.kind, self.Xrow[r0:r0 + nJ], xj, variance, length_scales, G)
                r0 += T
            if r0 < nrr:                                                           # symmetric partners (J, I): x 2
                G = C[r0:nrr, lj * T:lj * T + nJ]
                acc[:1 + nls] += 2.0 * ops.kernel_grad(self.kind, self.Xrow[r0:nrr], xj, variance, length_scales, G)
        if self.comm:
            dist.all_reduce(acc, op=dist.ReduceOp.SUM, group=self.group)
        return acc

    def predict(self, variance, length_scales, noise, resid, x_new, diag=True, max_tries=10):
        """GPR._predict (gpr.py:88-117) on the grid, for a zero mean function: (A^T V [n*, dy], var) with
        A = L^-1 K(X, x*), V = L^-1 (y - m); var = Kdiag - colsumsq(A) [n*] or K(x*) - A^T A [n*, n*].
        The test points ride through ONE factorisation as further residual rows: K(x*, X) appended to
        (y - m)^T comes out as A^T exactly like alpha^T does, distributed over the tile columns of the
        process row that holds the residual; mean and variance are column-partial sums, one all-reduce.
        (The reference re-factorises on every predict call as well, gpr.py:104.)"""
        ops, dy, dev = self.ops, self.dy, self.X.device
        ns = x_new.shape[0]
        Ks = ops.zeros(self.n, ns)
        ops.kernel_block(self.kind, self.X, x_new, variance, length_scales, Ks)
        R = torch.cat([resid, Ks], 1).contiguous()
        eng = BlockCyclicGP(self.X, R, self.kind, tile=self.T, grid=(self.pr, self.pc), ops=ops, group=self.group,
                            phantom=None if (self.comm or self.world == 1) else (self.rank, self.world), share=self,
                            schedule=self.schedule)
        eng.comm, eng.xrow, eng.xcol = self.comm, self.xrow, self.xcol
        eng.lml_rows = dy
        eng.refine = False        # the LML of this engine is discarded: no back-substitution sweep / residual pass for it
        eng.log_likelihood(variance, length_scales, noise, R, max_tries)
        self.info, self.jitter_rung = eng.info, eng.jitter_rung
        ncr = eng.cidx.numel()
        mean = torch.zeros(ns, dy, dtype=torch.float64, device=dev)
        out = torch.zeros((ns,) if diag else (ns, ns), dtype=torch.float64, device=dev)
        if eng.has_res and ncr:
            rows = eng.A[eng.res_off:]
            alphaT, AT = rows, rows[dy:]                       # [dy, ld], [ns, ld] (views; K padding is zero)
            mp = ops.zeros(_ops.round_up(ns, 16), _ops.round_up(dy, 16))
            ops.update(mp, AT, alphaT, ns, dy, ncr, lower=False, alpha=1.0, beta=0.0)
            mean += mp[:ns, :dy]
            if diag:
                out += ops.row_sumsq(AT, ns, ncr)
            else:
                g = ops.zeros(_ops.round_up(ns, 16), _ops.round_up(ns, 16))
                ops.update(g, AT, AT, ns, ns, ncr, lower=False, alpha=1.0, beta=0.0)
                out += g[:ns, :ns]
        if self.comm:
            dist.all_reduce(mean, op=dist.ReduceOp.SUM, group=self.group)
            dist.all_reduce(out, op=dist.ReduceOp.SUM, group=self.group)
        if diag and isinstance(self.kind, str):
            var = variance.reshape(()) - out                   # Kdiag = variance (kernels.py:174-179)
        elif diag:                                             # an expression: its own diagonal at the test points
            kss = ops.zeros(ns, ns)
            ops.kernel_block(self.kind, x_new, x_new, variance, length_scales, kss)
            var = kss.diagonal() - out
        else:
            kss = ops.zeros(ns, ns)
            ops.kernel_block(self.kind, x_new, x_new, variance, length_scales, kss)
            var = kss - out
        return mean, var

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
        refine = self.refine if self.refine is not None else \
            self.n >= min(_ops.refine_min_n(grid=True), _ops.refine_min_n(expression=not isinstance(self.kind, str)))
        refine = refine and (self.comm or self.world == 1)        # (a phantom rank of tools/dist_phantom_profile.py has no peers to ask)
        self.refined = False
        self.assemble(variance, length_scales, noise, resid)
        self.jitter_rung = -1
        if self.factor() == 0:
            if refine:
                self._refine(variance, length_scales, noise, resid)
            return self.lml()
        for i in range(max_tries):
            nz = noise + 10.0 ** (-max_tries + i)
            self.assemble(variance, length_scales, nz, resid)
            self.jitter_rung = i
            if self.factor() == 0:
                if refine:
                    self._refine(variance, length_scales, nz, resid)
                return self.lml()
        raise RuntimeError("Max tries exceeded.")

    def _refine(self, variance, length_scales, noise, resid):
        """One refinement step of y^T Kyy^-1 y on the grid (the single-GPU step of csrc/refine.hip / DESIGN 3.5, so that the
        value does not depend on how many GPUs computed it: at N = 65536 the plain value is 6e-8 from the CPU reference on
        1 x 2 GPUs and 9e-9 on 2 x 4, the refined one 4e-9 everywhere).
          1. alpha^T from the residual segment, replicated (one all-reduce of N x dy);
          2. a = L^-T alpha, tile row by tile row from the bottom: the owner of (J, J) applies its (pre-inverted) diagonal tile, a_J
             is broadcast, and process row J mod Pr adds L[J, j]^T a_J to what it owes the tile columns j < J (summed over
             the process column when their turn comes: one small all-reduce + one small broadcast per tile row);
          3. Kyy a from the points in double-double: the 64 x 64 tiles of the lower triangle are dealt evenly over the
             ranks regardless of where the factor's tiles live; one all-reduce of N x dy x 2;
          4. quad = y^T a + a^T (y - Kyy a)  ->  self._sumsq."""
        ops, T, nt, n, p = self.ops, self.T, self.nt, self.n, self.lml_rows
        lv = nt * T
        ev0 = ev1 = None
        if self.comm_timing and self.X.is_cuda:       # how long this rank's stream spends in the step (bench.py refine_ms_per_rank)
            ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            ev0.record()
        alpha = ops.zeros(p, lv)
        if self.has_res:
            for lj, J in enumerate(range(self.my_c, nt, self.pc)):
                nJ = self.rows_of(J)
                alpha[:, J * T:J * T + nJ] = self.A[self.res_off:self.res_off + p, lj * T:lj * T + nJ]
        if self.comm:
            dist.all_reduce(alpha, group=self.group)
        a = ops.zeros(p, lv)
        owed = ops.zeros(p, max(self.ncol_t, 1) * T)       # sum over MY tile rows I of L[I, j]^T a_I, per local column
        # the inverses of MY diagonal tiles, all ranks at once and before the serial sweep: a_J is then ONE skinny product
        # per step (the step tolerates any approximate a: the corrected value is exact to second order in y - Kyy a)
        buf, recv, sJ, mine_a = ops.zeros(p, T), ops.zeros(p, T), ops.zeros(p, T), ops.zeros(p, T)
        Uinv = {}
        for J in range(nt):
            if self.mine(J, J):
                Uinv[J] = ops.tile_inverse(self.A[((J - self.my_r) // self.pr) * T:, ((J - self.my_c) // self.pc) * T:], self.rows_of(J))
        for J in range(nt - 1, -1, -1):
            nJ, cj, rj = self.rows_of(J), J % self.pc, J % self.pr
            owner = rj * self.pc + cj
            if self.my_c == cj:
                lj = (J - self.my_c) // self.pc
                buf.zero_()
                buf[:, :nJ] = owed[:, lj * T:lj * T + nJ]
                if self.xcol:
                    dist.all_reduce(buf, group=self.col_group)
            aJ = recv                                      # (overwritten by the broadcast on every rank but the owner)
            if self.rank == owner:
                torch.sub(alpha[:, J * T:(J + 1) * T], buf, out=sJ)
                aJ = mine_a
                aJ.zero_()
                ops.gemv_t_acc(Uinv.pop(J), nJ, nJ, sJ, aJ)               # a_J = W^T s_J = L_JJ^-T s_J
            if self.comm:
                dist.broadcast(aJ, src=owner if self.group is None else dist.get_global_rank(self.group, owner), group=self.group)
            a[:, J * T:J * T + nJ] = aJ[:, :nJ]
            if self.my_r == rj and J > 0:
                ncl = self._cols_le(J - 1)
                if ncl:
                    li = (J - self.my_r) // self.pr
                    ops.gemv_t_acc(self.A[li * T:, :], nJ, ncl * T, aJ, owed)
        lds = _ops.round_up(n, LEAF)
        ar = a[:, :lds].contiguous()
        ntri = ops.refine_tile_count(n)
        q0, q1 = ntri * self.rank // self.world, ntri * (self.rank + 1) // self.world
        ka = ops.resid_part(self.kind, self.X, variance, length_scales, noise, ar, q0, q1)
        if self.comm:
            dist.all_reduce(ka, group=self.group)
        self._sumsq_plain = self._sumsq
        self._sumsq = float(ops.refine_finish(resid[:, :p].contiguous(), ar, ka))
        self.refined = True
        if ev0 is not None:
            ev1.record()
            ev1.synchronize()
            self.last_refine_ms = ev0.elapsed_time(ev1)


class NativeDistLML:
    """The same distributed evaluation through the C ABI (`gpn_dist_lml_forward`, csrc/dist.hip): the whole
    panel loop runs in the library, Python only supplies the communicator.  `comm` is either
    "rccl" (three RCCL communicators created here from the torch.distributed store -> the adapter
    library libgpnative_rccl.so; what a non-Python consumer does with its own ncclComm_t) or
    "torch" (callbacks over torch.distributed collectives -- any backend; how the test-suite runs
    several ranks on one GPU over gloo)."""

    def __init__(self, X, Y, kind, tile=2048, grid=None, comm="torch", force_comm=False, schedule=None):
        """schedule "mesh": GPN_DIST_MESH_EXCHANGE in the table's flags -- the RCCL adapter then moves panels by grouped
        ncclSend / ncclRecv over the direct links (csrc/rccl_adapter.cpp), the torch transport by `mesh_broadcast`."""
        import ctypes
        import os
        self.schedule = schedule or os.environ.get("GPN_DIST_SCHEDULE", "bcast")
        self.sent_bytes, self.recv_bytes = {}, 0
        from . import _native
        self._ct, self._native = ctypes, _native
        live = dist.is_available() and dist.is_initialized()
        self.rank = dist.get_rank() if live else 0
        self.world = dist.get_world_size() if live else 1
        self.pr, self.pc = grid if grid is not None else choose_grid(self.world)
        if self.pr * self.pc != self.world or self.pc % self.pr:
            raise ValueError("process grid %dx%d does not fit %d ranks (Pr must divide Pc)" % (self.pr, self.pc, self.world))
        self.X, self.Y, self.kind, self.T = X.contiguous(), Y.contiguous(), kind, int(tile)
        self.n, self.dy = Y.shape
        self.d = X.shape[1]
        lib = _native.lib()
        nbytes = int(lib.gpn_dist_work_bytes(self.rank, self.pr, self.pc, self.n, self.d, self.dy, self.T))
        if nbytes < 0:
            raise ValueError("bad grid / tile for gpn_dist_lml_forward")
        self.work = torch.empty(nbytes // 8, dtype=torch.float64, device=X.device)
        self.work_is_grad_sized = False
        self.rwork = None                        # workspace of gpn_dist_lml_refine (first use)
        self.refine = None                       # None = from refine_min_n() rows on (as the single-GPU path)
        self.refined = False
        self.out = torch.zeros(4, dtype=torch.float64, device=X.device)
        self.info = 0
        self.table = None
        self._keep = []
        need = live and (self.world > 1 or force_comm)
        if need and comm == "torch":
            self.table = self._torch_table(force_comm)
        elif need and comm == "rccl":
            self.table = self._rccl_table(force_comm)

    def _rccl_table(self, force):
        """RCCL communicators of our own (torch.distributed does not expose its ncclComm_t): rank 0 draws
        an ncclUniqueId, torch.distributed carries its 128 bytes to everybody, ncclCommInitRank builds the
        world communicator and ncclCommSplit the process-row / process-column ones -- exactly what a
        non-Python consumer does with its own bootstrap -- then libgpnative_rccl.so wraps the three in
        the callback table."""
        ct, nat = self._ct, self._native
        rccl = ct.CDLL("librccl.so")

        class UniqueId(ct.Structure):
            _fields_ = [("internal", ct.c_char * 128)]
        uid = UniqueId()
        if self.rank == 0 and rccl.ncclGetUniqueId(ct.byref(uid)) != 0:
            raise _ops.NativeError("ncclGetUniqueId failed")
        blob = [bytes(bytearray(uid)) if self.rank == 0 else None]
        dist.broadcast_object_list(blob, src=0)
        ct.memmove(ct.byref(uid), blob[0], 128)
        world = ct.c_void_p()
        rccl.ncclCommInitRank.argtypes = [ct.POINTER(ct.c_void_p), ct.c_int, UniqueId, ct.c_int]
        if rccl.ncclCommInitRank(ct.byref(world), self.world, uid, self.rank) != 0:
            raise _ops.NativeError("ncclCommInitRank failed")
        my_r, my_c = divmod(self.rank, self.pc)
        row, col = ct.c_void_p(), ct.c_void_p()
        rccl.ncclCommSplit.argtypes = [ct.c_void_p, ct.c_int, ct.c_int, ct.POINTER(ct.c_void_p), ct.c_void_p]
        if rccl.ncclCommSplit(world, my_r, my_c, ct.byref(row), None) != 0:      # colour = process row, key = column index
            raise _ops.NativeError("ncclCommSplit (process row) failed")
        if rccl.ncclCommSplit(world, my_c, my_r, ct.byref(col), None) != 0:      # colour = process column, key = row index
            raise _ops.NativeError("ncclCommSplit (process column) failed")
        table = nat.rccl_lib().gpn_rccl_comm_create(row, col, world)
        if not table:
            raise _ops.NativeError("gpn_rccl_comm_create failed")
        table.contents.flags = (nat.DIST_FORCE_COLLECTIVES if force else 0) | (nat.DIST_MESH_EXCHANGE if self.schedule == "mesh" else 0)
        self._keep = [rccl, world, row, col, table]
        return table.contents

    def _torch_table(self, force):
        ct, nat = self._ct, self._native
        my_r, my_c = divmod(self.rank, self.pc)
        row_group = col_group = None
        for r in range(self.pr):
            g = dist.new_group([r * self.pc + c for c in range(self.pc)]) if (self.pc > 1 or force) else None
            if r == my_r:
                row_group = g
        for c in range(self.pc):
            g = dist.new_group([r * self.pc + c for r in range(self.pr)]) if (self.pr > 1 or force) else None
            if c == my_c:
                col_group = g
        def view(ptr, count):
            for work in (self.work, self.rwork):   # looked up per call: the workspaces may have been re-allocated
                if work is None:
                    continue
                off = (ptr - work.data_ptr()) // 8
                if 0 <= off and off + count <= work.numel():
                    return work[off:off + count]
            raise AssertionError("collective on a buffer outside the workspaces")

        def bcast(ctx, which, buf, count, root, stream):
            try:
                torch.cuda.synchronize()          # the table's contract is stream order; this transport is host-side
                src = my_r * self.pc + root if which == 0 else root * self.pc + my_c
                group = row_group if which == 0 else col_group
                members = [my_r * self.pc + c for c in range(self.pc)] if which == 0 else [r * self.pc + my_c for r in range(self.pr)]
                if self.schedule == "mesh" and len(members) > 1:
                    for w in mesh_broadcast(view(buf, count), src, self.rank, members, group, None, self):
                        w.wait()
                else:
                    dist.broadcast(view(buf, count), src=src, group=group)
                torch.cuda.synchronize()
                return 0
            except Exception:                     # never let an exception cross the C frame
                import traceback
                traceback.print_exc()
                return -1

        def allreduce(ctx, buf, count, stream):
            try:
                torch.cuda.synchronize()
                dist.all_reduce(view(buf, count))
                torch.cuda.synchronize()
                return 0
            except Exception:
                import traceback
                traceback.print_exc()
                return -1

        b, a = nat.BCAST_FN(bcast), nat.ALLREDUCE_FN(allreduce)
        self._keep = [b, a]                       # the C side holds raw pointers to these
        return nat.DistComm(None, b, a, (nat.DIST_FORCE_COLLECTIVES if force else 0) | (nat.DIST_MESH_EXCHANGE if self.schedule == "mesh" else 0))

    def _evaluate(self, variance, length_scales, noise):
        ct = self._ct
        lib = self._native.lib()
        var, ls, nz = (_ops._c(t.detach()) for t in (variance, length_scales, noise))
        st = lib.gpn_dist_lml_forward(_ops._stream(self.X.device), ct.byref(self.table) if self.table is not None else None,
                                      self.rank, self.pr, self.pc, _ops.KINDS[self.kind], _ops._ptr(self.X), self.n, self.d,
                                      _ops._ptr(self.Y), self.dy, _ops._ptr(var), _ops._ptr(ls), ls.numel(), _ops._ptr(nz),
                                      self.T, _ops._ptr(self.work), self.work.numel() * 8, _ops._ptr(self.out))
        self._native.check(st, "gpn_dist_lml_forward")
        host = self.out.cpu()
        self.info = int(host[3])
        if self.info < 0:
            raise _ops.NativeError("a tile factorisation reported an internal status (not a property of the matrix)")
        return host

    def log_likelihood_and_grad(self, variance, length_scales, noise, max_tries=10):
        """(LML, [dLML/dvariance, dLML/dlength_scales..., dLML/dnoise], dLML/d(y - m) [n, dy]) through
        gpn_dist_lml_grad: forward + closed-form backward on the grid in ONE library call per attempt."""
        lib = self._native.lib()
        if not self.work_is_grad_sized:        # the callbacks map pointers into self.work: grow it in place of the old one
            nbytes = int(lib.gpn_dist_grad_work_bytes(self.rank, self.pr, self.pc, self.n, self.d, self.dy, self.T))
            self.work.resize_(nbytes // 8)
            self.work_is_grad_sized = True
        nls = length_scales.numel()
        grads = torch.zeros(2 + nls, dtype=torch.float64, device=self.X.device)
        g_resid = torch.zeros(self.n, self.dy, dtype=torch.float64, device=self.X.device)

        def attempt(nz):
            var, ls, nzc = (_ops._c(t.detach()) for t in (variance, length_scales, nz))
            st = lib.gpn_dist_lml_grad(_ops._stream(self.X.device), self._ct.byref(self.table) if self.table is not None else None,
                                       self.rank, self.pr, self.pc, _ops.KINDS[self.kind], _ops._ptr(self.X), self.n, self.d,
                                       _ops._ptr(self.Y), self.dy, _ops._ptr(var), _ops._ptr(ls), nls, _ops._ptr(nzc),
                                       self.T, _ops._ptr(self.work), self.work.numel() * 8, _ops._ptr(self.out), _ops._ptr(grads),
                                       _ops._ptr(g_resid))
            self._native.check(st, "gpn_dist_lml_grad")
            host = self.out.cpu()
            self.info = int(host[3])
            if self.info < 0:
                raise _ops.NativeError("a tile factorisation reported an internal status (not a property of the matrix)")
            return host
        host = attempt(noise)
        self.jitter_rung = -1
        for i in range(max_tries):
            if self.info == 0:
                break
            self.jitter_rung = i
            host = attempt(noise + 10.0 ** (-max_tries + i))
        if self.info != 0:
            raise RuntimeError("Max tries exceeded.")
        self.refined = False
        if self.refine if self.refine is not None else self.n >= _ops.refine_min_n(grid=True):
            # the value by the same rule as log_likelihood (the factor and alpha^T are where the forward part left them: the
            # backward writes its inverse elsewhere in the workspace)
            nz = noise if self.jitter_rung < 0 else noise + 10.0 ** (-max_tries + self.jitter_rung)
            host = self._refine(variance, length_scales, nz)
        return host[2].to(self.X.device), grads, g_resid

    def predict(self, variance, length_scales, noise, x_new, mean_new=None, diag=True, max_tries=10):
        """GPR._predict (gpr.py:88-117) through gpn_dist_predict: (mean [n*, dy], var [n*] or cov [n*, n*]), identical on
        every rank; self.Y is the residual y - m(X), mean_new the mean function at the test points (None = zero).  Same
        jitter ladder as log_likelihood (the reference re-factorises in every predict call, gpr.py:104)."""
        ct, lib = self._ct, self._native.lib()
        xs = _ops._c(x_new.detach())
        ns = xs.shape[0]
        nbytes = int(lib.gpn_dist_predict_work_bytes(self.rank, self.pr, self.pc, self.n, self.d, self.dy, ns, self.T, 0 if diag else 1))
        if self.work.numel() * 8 < nbytes:
            self.work.resize_(nbytes // 8)          # (in place: the torch transport's callbacks map pointers into self.work)
        mean = torch.empty(ns, self.dy, dtype=torch.float64, device=self.X.device)
        var = torch.empty((ns,) if diag else (ns, ns), dtype=torch.float64, device=self.X.device)
        ms = None if mean_new is None else _ops._c(mean_new.detach().expand(ns, self.dy))
        v, ls = _ops._c(variance.detach()), _ops._c(length_scales.detach())

        def attempt(nz):
            nzc = _ops._c(nz.detach())
            st = lib.gpn_dist_predict(_ops._stream(self.X.device), ct.byref(self.table) if self.table is not None else None,
                                      self.rank, self.pr, self.pc, _ops.KINDS[self.kind], _ops._ptr(self.X), self.n, self.d,
                                      _ops._ptr(self.Y), self.dy, _ops._ptr(xs), ns, _ops._ptr(ms), _ops._ptr(v), _ops._ptr(ls), ls.numel(),
                                      _ops._ptr(nzc), self.T, 0 if diag else 1, _ops._ptr(self.work), self.work.numel() * 8,
                                      _ops._ptr(self.out), _ops._ptr(mean), _ops._ptr(var))
            self._native.check(st, "gpn_dist_predict")
            self.info = int(self.out.cpu()[3])
            if self.info < 0:
                raise _ops.NativeError("a tile factorisation reported an internal status (not a property of the matrix)")
        attempt(noise)
        self.jitter_rung = -1
        for i in range(max_tries):
            if self.info == 0:
                break
            self.jitter_rung = i
            attempt(noise + 10.0 ** (-max_tries + i))
        if self.info != 0:
            raise RuntimeError("Max tries exceeded.")
        return mean, var

    def log_likelihood(self, variance, length_scales, noise, max_tries=10):
        """LML with the jitter ladder of functions.py:20-43 on the all-reduced info word."""
        host = self._evaluate(variance, length_scales, noise)
        self.jitter_rung = -1
        for i in range(max_tries):
            if self.info == 0:
                break
            self.jitter_rung = i
            host = self._evaluate(variance, length_scales, noise + 10.0 ** (-max_tries + i))
        if self.info != 0:
            raise RuntimeError("Max tries exceeded.")
        self.refined = False
        if self.refine if self.refine is not None else self.n >= _ops.refine_min_n(grid=True):
            nz = noise if self.jitter_rung < 0 else noise + 10.0 ** (-max_tries + self.jitter_rung)
            host = self._refine(variance, length_scales, nz)
        return host[2].to(self.X.device)

    def _refine(self, variance, length_scales, noise):
        """gpn_dist_lml_refine on the factor the forward call left in self.work."""
        ct, lib = self._ct, self._native.lib()
        if self.rwork is None:
            nbytes = int(lib.gpn_dist_lml_refine_work_bytes(self.rank, self.pr, self.pc, self.n, self.d, self.dy, self.T))
            self.rwork = torch.empty(nbytes // 8, dtype=torch.float64, device=self.X.device)
        var, ls, nz = (_ops._c(t.detach()) for t in (variance, length_scales, noise))
        st = lib.gpn_dist_lml_refine(_ops._stream(self.X.device), ct.byref(self.table) if self.table is not None else None,
                                     self.rank, self.pr, self.pc, _ops.KINDS[self.kind], _ops._ptr(self.X), self.n, self.d,
                                     _ops._ptr(self.Y), self.dy, _ops._ptr(var), _ops._ptr(ls), ls.numel(), _ops._ptr(nz),
                                     self.T, _ops._ptr(self.work), _ops._ptr(self.rwork), self.rwork.numel() * 8, _ops._ptr(self.out))
        self._native.check(st, "gpn_dist_lml_refine")
        self.refined = True
        return self.out.cpu()

#!/usr/bin/env python3
"""GPU-box tool: two-level panels (potrf.hip outer_width) -- ms per LML evaluation, single and lock-step batch of B,
for inner / outer panel widths given as "PW:OW" pairs.  usage: outer_ab.py <n> <B> PW:OW [PW:OW ...]   (tools' build)"""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), "..")))
from gptorch_amd import _native, _ops, rng
dev = torch.device("cuda:0")
n, B = int(sys.argv[1]), int(sys.argv[2])
d = 8
x, y = rng.make_regression(n, d, 1, seed=0)
X, Y = torch.as_tensor(x).to(dev), torch.as_tensor(y).to(dev)
var = torch.linspace(1.0, 1.1, B, dtype=torch.float64, device=dev)
ls = (float(np.sqrt(d)) * torch.linspace(1.0, 1.2, B, dtype=torch.float64, device=dev))[:, None]
nz = torch.full((B,), 1e-2, dtype=torch.float64, device=dev)
lib = _native.debug_begin()
fb1 = fbB = None
for rep in range(2):
    for pair in sys.argv[3:]:
        pw, ow, fl, ow2 = ([int(v, 0) for v in pair.split(":")] + [0, 0])[:4]      # PW:W1[:variant bits, e.g. 0x20 = right-looking in-panel[:W2]]
        lib.gpn_debug_set_potrf_variant(((pw // 128) << 8) | (fl & 0xffff))
        lib.gpn_debug_set_outer_width(ow, ow2)
        lib.gpn_debug_set_extra_rows(0 if (fl & 0x10000) else 1)     # flag 0x10000: the extra rows as a tile row of the big update
        lib.gpn_debug_set_inner_lookahead(1 if (fl & 0x40000) else 0)     # 0x40000: look-ahead inside the outer panel
        lib.gpn_debug_set_fused_steps(1 if (fl & 0x100000) else 0)     # 0x100000: fused chain steps
        lib.gpn_debug_set_fused_colstep(1 if (fl & 0x200000) else 0)     # 0x200000: the chain's two column passes in one launch
        lib.gpn_debug_set_thin_tiles(0 if (fl & 0x400000) else 1)     # 0x400000: no thin-tile path in the contraction kernels
        lib.gpn_debug_set_inner_left(1 if (fl & 0x800000) else 0)     # 0x800000: left-looking formation of the inner panels (round 5)
        fl &= 0xffff
        lib.gpn_debug_set_big_tile_min_trapezoid(int(os.environ.get('TRAP_MIN', '4096')))
        res = []
        for nb in ((1, B) if B > 1 else (1,)):
            fb = fb1 if nb == 1 else fbB
            for _ in range(2):
                fb, t = _ops.lml_forward_batched("Rbf", X, Y, var[:nb], ls[:nb], nz[:nb], fb=fb)
            torch.cuda.synchronize()
            reps = max(3, int(20 * (8192 / n) ** 3 / nb) )
            t0 = time.perf_counter()
            for _ in range(reps):
                fb, t = _ops.lml_forward_batched("Rbf", X, Y, var[:nb], ls[:nb], nz[:nb], fb=fb)
            torch.cuda.synchronize()
            dt = (time.perf_counter() - t0) / reps
            res.append((dt, float(t[0].sum().item())))
            if nb == 1: fb1 = fb
            else: fbB = fb
        print("n %d PW %4d W1 %4d W2 %4d fl 0x%x: single %.3f ms | B=%d %.2f ms = %.1f evals/s | lml[0] %.10f %.10f" % (
            n, pw, ow, ow2, fl, res[0][0] * 1e3, B, res[-1][0] * 1e3, B / res[-1][0], res[0][1], res[-1][1]), flush=True)
_native.debug_end()

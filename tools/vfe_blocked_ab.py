#!/usr/bin/env python3
"""GPU-box tool: config 5 (VFE, N = 1e6, M = 4096) with the chunk right-solves through the inverted 1024 x 1024 diagonal blocks
(round 4) and through the recursion down to the 128-wide leaf inverses, alternating processes."""
import os
import subprocess
import sys
ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
for mn in (2048, 10 ** 9, 2048, 10 ** 9):
    code = ("import sys, runpy; sys.path.insert(0, %r); from gptorch_amd.models import sparse_gpr; sparse_gpr.BLOCKED_SOLVE_MIN_M = %d; "
            "sys.argv = ['vfe_bench.py', '--steps', '3']; runpy.run_path(%r, run_name='__main__')" % (ROOT, mn, os.path.join(ROOT, "tools", "vfe_bench.py")))
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, cwd=ROOT)
    print("BLOCKED_SOLVE_MIN_M = %d:" % mn, (out.stdout + out.stderr).strip()[:420], flush=True)

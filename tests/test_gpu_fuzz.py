"""Randomised parity where the driver sees it: fixed-seed, bounded slices of the sweeps under tests/sweeps/ (run by hand at
380 cases per round, profiles/r3_sweeps_final.txt) as `-m gpu` tests.  Each slice is the sweep script itself in a child
process (they edit module globals such as the VFE chunk size), with a fixed case count and seed; the script exits 1 on the
first violated tolerance and prints the offending case.  The CPU oracle (oracle/gp_oracle.py) is the checker on one side of
fuzz_parity / fuzz_vfe; fuzz_expr checks the fused expression kernels against the reference's way of composing the same
tree (children's dense matrices combined by + and *, kernels.py:286-306)."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))


def _run(script, cases, seed, timeout):
    out = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "sweeps", script), str(cases), str(seed)],
                         cwd=ROOT, capture_output=True, text=True, timeout=timeout)
    tail = (out.stdout + out.stderr)[-3000:]
    assert out.returncode == 0, tail
    summary = [ln for ln in out.stdout.splitlines() if ln.startswith("cases ")]
    assert summary and "violations 0" in summary[-1], tail
    return summary[-1]


@pytest.mark.gpu
@pytest.mark.parametrize("seed", [11, 12])
def test_fuzz_gpr_against_the_oracle(device, seed):
    """GPR loss, the three raw-parameter gradients and predict_f at random sizes on and around every blocking edge (16-pivot
    blocks, 128 leaf, 1536 panel), all stationary kinds, ARD / isotropic, dy 1..4, noise 1e-3 .. 0.1
    (gpr.py:47-117 through gptorch_amd against oracle.GPROracle)."""
    print(_run("fuzz_parity.py", 14, seed, 600))


@pytest.mark.gpu
def test_fuzz_vfe_against_the_oracle(device):
    """VFE bound, every gradient incl. the inducing points, predictions, with random chunk sizes of the streamed evaluation
    (sparse_gpr.py:108-195 against oracle.VFEOracle and its autograd)."""
    print(_run("fuzz_vfe.py", 8, 5, 600))


@pytest.mark.gpu
def test_fuzz_fused_expressions_against_composed_kernels(device):
    """random Sum / Product trees: K(X), K(X, X2), parameter gradients, GPR loss + backward on the fused path against the same
    tree composed from its children's dense matrices."""
    print(_run("fuzz_expr.py", 24, 77, 600))


@pytest.mark.gpu
def test_fuzz_lockstep_fit_against_sequential(device):
    """random lock-step groups (sizes around every blocking edge, all kinds, ARD, dy 1..3, shared / own data, two groups
    interleaved per call) through batched_loss_and_grad against each model's own loss(); backward() (base.py:260-269):
    losses and gradients bit for bit."""
    print(_run("fuzz_lockstep.py", 16, 3, 600))

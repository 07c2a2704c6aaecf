"""CPU suite, part 4: the plain-C restatement (oracle/gp_oracle.c) agrees with the
torch-CPU oracle and with the goldens generated from the reference."""
import ctypes
import os
import subprocess

import numpy as np
import pytest

from gptorch_amd import rng
from tests._util import load_json, load_npz

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
KIND = {"Rbf": 0, "Matern52": 1, "Matern32": 2, "Exp": 3, "Periodic": 5}


@pytest.fixture(scope="module")
def clib():
    subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "oracle")])
    lib = ctypes.CDLL(os.path.join(ROOT, "oracle", "libgporacle.so"))
    dp = ctypes.POINTER(ctypes.c_double)
    lib.gpo_kernel_matrix.argtypes = [ctypes.c_int, dp, ctypes.c_int, dp, ctypes.c_int, ctypes.c_int,
                                      ctypes.c_double, dp, ctypes.c_int, dp]
    lib.gpo_gpr_lml.argtypes = [ctypes.c_int, dp, ctypes.c_int, ctypes.c_int, dp, ctypes.c_int, ctypes.c_double,
                                dp, ctypes.c_int, ctypes.c_double, dp]
    lib.gpo_cholesky.argtypes = [dp, ctypes.c_int]
    return lib


def _p(a):
    return a.ctypes.data_as(ctypes.POINTER(ctypes.c_double))


def test_c_kernels_match_reference_fixtures(clib):
    z = load_npz("ref_kernel_fixtures.npz")
    x1, x2 = np.ascontiguousarray(z["x1"]), np.ascontiguousarray(z["x2"])
    one, ard = np.ones(1), np.ascontiguousarray(z["ard_length_scales"])
    for name, kind in KIND.items():
        out = np.empty((4, 5))
        clib.gpo_kernel_matrix(kind, _p(x1), 4, _p(x2), 5, 3, 1.0, _p(one), 1, _p(out))
        assert np.allclose(out, z[f"{name}_kx2"])
        clib.gpo_kernel_matrix(kind, _p(x1), 4, _p(x2), 5, 3, 1.0, _p(ard), 3, _p(out))
        assert np.allclose(out, z[f"{name}_kx2_ard"])
        sq = np.empty((4, 4))
        clib.gpo_kernel_matrix(kind, _p(x1), 4, _p(x1), 4, 3, 1.0, _p(one), 1, _p(sq))
        assert np.allclose(sq, z[f"{name}_kx"])


def test_c_lml_matches_goldens(clib):
    for case in load_json("lml_cases.json"):
        if case["n"] > 512:
            continue
        x, y = rng.make_regression(case["n"], case["d"], case["dy"], seed=0)
        resid = np.ascontiguousarray(y - (np.asarray(case["mean"]) if case.get("mean") else 0.0))
        ls = np.atleast_1d(np.asarray(case["length_scales"], dtype=np.float64)).copy()
        out = ctypes.c_double()
        rung = clib.gpo_gpr_lml(KIND[case["kind"]], _p(np.ascontiguousarray(x)), case["n"], case["d"], _p(resid),
                                case["dy"], case["variance"], _p(ls), ls.size, case["noise"], ctypes.byref(out))
        assert rung == -1
        assert abs(out.value - case["lml"]) < 1e-8 * max(1.0, abs(case["lml"])), case["name"]


def test_c_jitter_ladder(clib):
    a = np.array([[1.0, 2.0], [2.0, 1.0]])
    assert clib.gpo_cholesky(_p(a), 2) == 2
    b = np.array([[4.0, 2.0], [2.0, 5.0]])
    assert clib.gpo_cholesky(_p(b), 2) == 0 and np.allclose(b, [[2.0, 0.0], [1.0, 2.0]])

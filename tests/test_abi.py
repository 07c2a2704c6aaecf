"""CPU suite, part 2: the C-ABI shared library loads and exports every symbol that
include/gpnative.h declares; pure-host entry points and argument validation behave.
No compute is launched (there is no GPU here)."""
import ctypes
import os
import re

import pytest

from gptorch_amd import _native

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))


def _declared():
    text = open(os.path.join(ROOT, "include", "gpnative.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(gpn_\w+)\s*\(", text)))


def test_library_present_and_loads():
    assert os.path.exists(_native.LIB_PATH), "run __graft_entry__.build() first"
    lib = _native.lib()
    assert lib.gpn_version() == 1
    assert lib.gpn_arch() == b"gfx950"
    assert lib.gpn_last_hip_error() == b""


def test_every_declared_symbol_is_exported_and_bound():
    lib = ctypes.CDLL(_native.LIB_PATH)
    names = _declared()
    assert len(names) >= 18
    rccl = ctypes.CDLL(_native.RCCL_LIB_PATH)       # the RCCL adapter of the gpn_dist_comm table: its own library
    for n in names:
        if n in _native.RCCL_SIGNATURES:
            assert hasattr(rccl, n), "libgpnative_rccl.so does not export %s" % n
            continue
        assert hasattr(lib, n), "libgpnative.so does not export %s" % n
        assert n in _native.SIGNATURES, "gptorch_amd._native has no ctypes signature for %s" % n
    for n in list(_native.SIGNATURES) + list(_native.RCCL_SIGNATURES):
        assert n in names, "%s is bound but not declared in include/gpnative.h" % n


def test_factor_geometry():
    lib = _native.lib()
    for n, e in [(1, 0), (63, 1), (64, 0), (64, 1), (8192, 1), (1000, 3)]:
        ld, rows = lib.gpn_factor_ld(n, e), lib.gpn_factor_rows(n, e)
        assert ld % 128 == 0 and ld >= n + e and ld - (n + e) < 128
        assert rows == ld + 16
        assert lib.gpn_winv_bytes(n) == ((n + 127) // 128) * 128 * 128 * 8
    assert lib.gpn_grad_work_bytes(128, 128, 3, 1) == 3 * 5 * 8
    assert lib.gpn_grad_work_bytes(128, 64, 1, 0) == 2 * 3 * 8


def test_argument_validation_without_launch():
    """bad arguments are rejected with -(argument index) / GPN_E_* before any HIP call."""
    lib = _native.lib()
    null = None
    # staircase with lower-only first squares: a block with fewer than blk rows (M = 3000, blk = step = 2048, 2 blocks) is refused
    assert lib.gpn_gemm_nt_stair(null, 3000, 2, 2048, 16, -1.0, null, 16, null, 16, 1.0, null, 4096, 2048, 1) == -15
    assert lib.gpn_gemm_nt_stair(null, 1000, 1, 2048, 16, -1.0, null, 16, null, 16, 1.0, null, 4096, 2048, 1) == -15
    assert lib.gpn_gemm_nt(null, 16, 16, 10, 1.0, null, 16, null, 16, 0.0, null, 16, 0, 0) == -4     # K % 16
    assert lib.gpn_gemm_nt(null, 16, 8, 16, 1.0, null, 16, null, 16, 0.0, null, 16, 1, 0) == -13    # lower, M != N
    assert lib.gpn_gemm_nt(null, 16, 16, 16, 1.0, null, 15, null, 16, 0.0, null, 16, 0, 0) == -101  # odd lda
    assert lib.gpn_kernel_matrix(null, 9, null, 4, null, 4, 2, null, null, 1, null, 0, null, 4) == -2
    assert lib.gpn_kernel_matrix(null, 0, null, 4, null, 4, 2, null, null, 1, null, 0, null, 4) == -3
    assert lib.gpn_potrf_lower(null, null, 4, 0, 128, null, null) == -2
    assert lib.gpn_trsm_right_lt(null, null, 4, 128, null, null, 1, 128) == -2
    assert lib.gpn_lml_grad(null, 0, null, 4, 2, null, null, 1, null, 4, null, 4, 1, null, null) == -3
    # whole-path entry points
    assert lib.gpn_lml_forward(null, 0, null, -1, 2, null, null, 1, null, null, 1, null, null, 128, null, null, null) == -4
    assert lib.gpn_lml_forward(null, 0, null, 4, 2, null, null, 1, null, null, 1, null, null, 128, null, null, null) == -6
    assert lib.gpn_lml_backward(null, 0, null, 4, 2, null, null, 1, null, 128, null, 1, null, null, null) == -9
    assert lib.gpn_predict(null, 0, null, 4, 2, null, 3, null, null, null, 1, null, 128, null, 1, 0, null, null, null) == -6
    assert lib.gpn_lml_backward_work_bytes(1000, 2, 3) >= 2 * lib.gpn_factor_rows(1000, 0) * lib.gpn_factor_ld(1000, 0) * 8
    assert lib.gpn_predict_work_bytes(1000, 5, 2) == 128 * lib.gpn_factor_ld(1000, 2) * 8
    # distributed driver: grid / tile validation and workspace sizing are pure host code
    assert lib.gpn_dist_work_bytes(0, 2, 3, 1000, 2, 1, 128) == -1             # Pr must divide Pc
    assert lib.gpn_dist_work_bytes(0, 2, 4, 1000, 2, 1, 100) == -1             # tile % 128
    assert lib.gpn_dist_work_bytes(8, 2, 4, 1000, 2, 1, 128) == -1             # rank outside the grid
    w1, w8 = lib.gpn_dist_work_bytes(0, 1, 1, 65536, 32, 1, 2048), lib.gpn_dist_work_bytes(3, 2, 4, 65536, 32, 1, 2048)
    assert w1 > 65536 * 65536 * 8 and w8 < w1 / 5 and w8 % 256 == 0
    assert lib.gpn_dist_lml_forward(null, null, 0, 2, 4, 0, null, 100, 2, null, 1, null, null, 1, null, 128, null, 0, null) == -2   # no comm table
    # the refinement step in pieces and on the grid: sizing and validation are host code as well
    assert lib.gpn_refine_tile_count(65536) == 1024 * 1025 // 2 and lib.gpn_refine_tile_count(65) == 3
    assert lib.gpn_refine_resid_part_work_bytes(1, 10) == 2 * 10 * 64 * 2 * 8 and lib.gpn_refine_resid_part_work_bytes(0, 10) == 0
    assert lib.gpn_gemv_t_work_bytes(2048, 16384, 1) > 0 and lib.gpn_gemv_t_work_bytes(0, 16384, 1) == 0
    assert lib.gpn_gemv_t_acc(null, null, 16, 4, 4, null, 4, 1, null, 4, null) == -2
    assert lib.gpn_refine_resid_part(null, 0, null, 4, 2, null, null, 1, null, null, 1, 0, 1, null, null) == -3
    assert lib.gpn_refine_finish(null, null, null, null, null, 4, 1, null) == -2
    assert lib.gpn_dist_lml_refine_work_bytes(0, 2, 3, 1000, 2, 1, 128) == -1                       # Pr must divide Pc
    r1, r8 = lib.gpn_dist_lml_refine_work_bytes(0, 1, 1, 65536, 32, 1, 2048), lib.gpn_dist_lml_refine_work_bytes(3, 2, 4, 65536, 32, 1, 2048)
    assert r8 % 256 == 0 and 0 < r8 < r1 / 3          # (the tile inverses and the residual share divide by the ranks)
    assert lib.gpn_dist_lml_refine(null, null, 0, 2, 4, 0, null, 100, 2, null, 1, null, null, 1, null, 128, null, null, 0, null) == -2   # no comm table
    one_ = ctypes.c_double(0.0)
    p_ = ctypes.cast(ctypes.pointer(one_), ctypes.c_void_p)
    # round 6: the lock-step forms of the single-purpose entries, the ragged batch, the persistent factorisation
    assert lib.gpn_kernel_matrix_batched(null, 9, 2, null, 0, 4, null, 0, 4, 2, null, null, 1, null, 0, null, 4, 16) == -2
    assert lib.gpn_kernel_matrix_batched(null, 0, 0, null, 0, 4, null, 0, 4, 2, null, null, 1, null, 0, null, 4, 16) == -3
    assert lib.gpn_kernel_matrix_batched(null, 0, 2, null, 0, 4, null, 0, 4, 2, null, null, 1, null, 0, null, 4, 16) == -4
    assert lib.gpn_trsm_right_lt_batched(null, null, 4, 128, 0, null, 0, null, 1, 128, 0, 2) == -2
    assert lib.gpn_trtri_upper_batched(null, null, 4, 128, 0, null, 0, null, 128, 0, null, 128, 0, 2) == -2
    assert lib.gpn_gemm_nt_batched_scaled(null, 16, 16, 10, null, null, 16, 0, null, 16, 0, 0.0, null, 16, 0, 0, 0, 2) == -4      # K % 16
    assert lib.gpn_gemm_nt_batched_scaled(null, 16, 16, 16, null, null, 16, 0, null, 16, 0, 0.0, null, 16, 0, 0, 0, 2) == -5      # no scales
    assert lib.gpn_kernel_grad_batched(null, 0, 0, null, 0, 4, null, 0, 4, 2, null, null, 1, null, 4, 0, null, null) == -3
    assert lib.gpn_kernel_grad_x2_batched(null, 0, 2, null, 0, 4, null, 0, 4, 2, null, null, 1, null, 4, 0, 1.0, 0, null, null) == -4
    assert lib.gpn_dot2d_batched(null, null, 1, 0, null, 0, 0, 4, 1, null, 1) == -2
    assert lib.gpn_lml_forward_ragged(null, 0, 2, null, 0, 1000, null, 2, null, 0, 1, null, null, 1, null, null, 1152, 0, null, 0, null, null) == -4
    assert lib.gpn_lml_forward_ragged(null, 0, 2, p_, 0, 200, null, 2, null, 0, 1, null, null, 1, null, null, 256, 0, null, 0, null, null) == -6   # n <= 256
    assert lib.gpn_lml_forward_ragged(null, 0, 2, p_, 0, 1000, null, 2, null, 0, 1, null, null, 1, null, null, 1152, 0, null, 0, null, null) == -7  # no sizes
    assert lib.gpn_lml_backward_ragged(null, 0, 2, p_, 0, 1000, null, 2, null, null, 1, null, 1152, 0, null, 0, 1, null, null) == -7
    assert lib.gpn_potrf_persistent_supported(8192, 1) == 1 and lib.gpn_potrf_persistent_supported(1000, 1) == 0
    assert lib.gpn_kernel_matrix_expr_batched(null, None, 0, None, 0, 2, null, 0, null, 0, 4, 2, null, null, 4, 16) == -2     # no program
    assert lib.gpn_kernel_expr_grad_batched(null, None, 0, None, 0, 2, null, 0, 0, null, 0, 4, 2, null, 4, 0, null, 4, 0, 1, 0, null, null) == -2
    assert lib.gpn_potrf_persistent_supported(32768, 1) == 0 and lib.gpn_potrf_persistent_supported(8200, 1) == 0
    # zero-size problems are no-ops that succeed
    one = ctypes.c_double(0.0)
    p = ctypes.cast(ctypes.pointer(one), ctypes.c_void_p)
    assert lib.gpn_potrf_lower(null, p, 0, 0, 128, p, p) == 0
    assert lib.gpn_transpose(null, p, 0, 0, 0, p, 0) == 0


def test_missing_library_fails_loudly(monkeypatch, tmp_path):
    monkeypatch.setattr(_native, "_lib", None)
    monkeypatch.setattr(_native, "LIB_PATH", str(tmp_path / "nope.so"))
    with pytest.raises(_native.NativeError, match="no CPU fallback"):
        _native.lib()


def _build_c_consumer(out_path):
    import subprocess
    lib_dir = os.path.dirname(_native.LIB_PATH)
    cmd = ["gcc", "-std=c99", "-Wall", "-Werror", os.path.join(ROOT, "examples", "lml_consumer.c"), "-I" + os.path.join(ROOT, "include"),
           "-I/opt/rocm/include", "-L" + lib_dir, "-lgpnative", "-L/opt/rocm/lib", "-lamdhip64",
           "-Wl,-rpath," + lib_dir, "-Wl,-rpath,/opt/rocm/lib", "-lm", "-o", str(out_path)]
    r = subprocess.run(cmd, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-3000:]
    return str(out_path)


def _build_c_dist_consumer(out_path):
    import subprocess
    lib_dir = os.path.dirname(_native.LIB_PATH)
    cmd = ["gcc", "-std=c99", "-Wall", "-Werror", os.path.join(ROOT, "examples", "dist_consumer.c"), "-I" + os.path.join(ROOT, "include"),
           "-I/opt/rocm/include", "-L" + lib_dir, "-lgpnative", "-lgpnative_rccl", "-L/opt/rocm/lib", "-lrccl", "-lamdhip64",
           "-Wl,-rpath," + lib_dir, "-Wl,-rpath,/opt/rocm/lib", "-lm", "-o", str(out_path)]
    r = subprocess.run(cmd, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-3000:]
    return str(out_path)


def test_c_dist_consumer_links(tmp_path):
    """examples/dist_consumer.c (C99, gcc): gpn_dist_lml_forward + the RCCL adapter's callback table link
    from plain C against libgpnative.so + libgpnative_rccl.so + librccl.  (RUN in the gpu suite.)"""
    _build_c_dist_consumer(tmp_path / "dist_consumer")


def test_header_is_plain_c_and_a_c_consumer_links(tmp_path):
    """include/gpnative.h compiles as C99 and examples/lml_consumer.c (gcc, no hipcc, no Python)
    links against libgpnative.so: the boundary really is a C ABI.  (It is RUN in the gpu suite.)"""
    _build_c_consumer(tmp_path / "lml_consumer")


@pytest.mark.parametrize("p", [1, 2, 3, 4, 8])
def test_c_mesh_plan_equals_python_plan(p):
    """gpn_mesh_plan (libgpnative_rccl.so: what the adapter's grouped ncclSend / ncclRecv loop executes) is the same
    schedule as gptorch_amd.dist.mesh_plan (checked for consistency by tests/test_dist_gloo.py) for every member, root,
    ragged counts, both forms and a too-small output buffer."""
    from gptorch_amd import dist as gdist
    rccl = _native.rccl_lib()
    members = list(range(p))
    for root in members:
        for count, stages, direct in [(0, 4, 0), (1, 4, 0), (5, 4, 0), (97, 4, 0), (1000, 3, 0), (1000, 4, 4096), (64, 64, 0),
                                      (268435456, 4, 524288)]:
            for me in members:
                want = []
                for t, stage in enumerate(gdist.mesh_plan(members, root, me, count, stages, direct)):
                    want += [(t, 0 if kind == "send" else 1, peer, off, ln) for kind, peer, off, ln in stage]
                buf = (ctypes.c_int64 * (5 * 2))()
                n = rccl.gpn_mesh_plan(p, root, me, count, stages, direct, buf, 2)
                assert n == len(want)
                buf = (ctypes.c_int64 * (5 * max(n, 1)))()
                assert rccl.gpn_mesh_plan(p, root, me, count, stages, direct, buf, n) == n
                got = [tuple(buf[5 * i:5 * i + 5]) for i in range(n)]
                assert got == want, (p, root, me, count)
    assert rccl.gpn_mesh_plan(2, 2, 0, 10, 4, 0, None, 0) == -1


def test_product_library_has_no_debug_switches():
    """the shipped libgpnative.so carries no mutable A/B state: the gpn_debug_* switches exist only in the tools' build
    libgpnative_dbg.so (same sources, -DGPN_DEBUG_SWITCHES, per calling thread), which still exports the whole C ABI."""
    prod = ctypes.CDLL(_native.LIB_PATH)
    dbg = ctypes.CDLL(_native.DEBUG_LIB_PATH)
    for n in _native.DEBUG_SIGNATURES:
        assert not hasattr(prod, n), "libgpnative.so exports %s" % n
        assert hasattr(dbg, n), "libgpnative_dbg.so lacks %s" % n
    for n in _native.SIGNATURES:
        assert hasattr(dbg, n)


def test_gemv_scratch_size_is_monotone_in_the_width():
    """gpn_gemv_t_work_bytes must not shrink when the width grows: gpn_dist_lml_refine sizes its scratch once for the widest
    call and reuses it for every narrower one (round-3 advice: chunks(c) * c alone is not monotone -- T = 2048: 31 tile
    columns needed 317440 doubles, 32 only 262144)."""
    from gptorch_amd import _native
    lib = _native.lib()
    for rows in (128, 2048, 4096):
        for dy in (1, 3):
            prev = 0
            for cols in list(range(1, 600, 7)) + [k * 2048 for k in range(1, 40)]:
                b = lib.gpn_gemv_t_work_bytes(rows, cols, dy)
                assert b > 0
            widths = sorted(set(list(range(1, 600, 7)) + [k * 2048 for k in range(1, 40)]))
            for cols in widths:
                b = lib.gpn_gemv_t_work_bytes(rows, cols, dy)
                assert b >= prev, (rows, cols, dy, b, prev)
                prev = b
            # and it still covers what one call uses: chunks * min(dy, 2) * cols doubles with chunks = ceil(1024 / ceil(cols / 256))
            for cols in widths:
                wgs = (cols + 255) // 256
                chunks = max(1, min((1024 + wgs - 1) // wgs, (rows + 31) // 32))
                assert lib.gpn_gemv_t_work_bytes(rows, cols, dy) >= chunks * min(dy, 2) * cols * 8


def test_panel_levels_nest_and_end_at_the_priced_width():
    """gpn_potrf_panel_levels: widths ascend, each divides the next, all are multiples of the 128-wide leaf, and the last one is
    gpn_potrf_panel_width (the lower-tile update bench.py prices); sizes the recursive driver takes report no levels."""
    import ctypes
    from gptorch_amd import _native
    lib = _native.lib()
    w = (ctypes.c_int64 * 3)()
    assert lib.gpn_potrf_panel_levels(256, None) == -2
    assert lib.gpn_potrf_panel_levels(256, ctypes.cast(w, ctypes.c_void_p)) == 0 and lib.gpn_potrf_panel_width(256) == 0
    for n, expect in ((1000, [256]), (2048, [256]), (4096, [256, 1024]), (8192, [256, 1024]), (16384, [256, 1024]),
                      (32768, [512, 2048]), (65536, [512, 2048, 4096])):
        cnt = lib.gpn_potrf_panel_levels(n, ctypes.cast(w, ctypes.c_void_p))
        got = [int(w[i]) for i in range(cnt)]
        assert got == expect, (n, got)
        assert all(v % 128 == 0 for v in got) and all(b % a == 0 and b > a for a, b in zip(got, got[1:]))
        assert lib.gpn_potrf_panel_width(n) == got[-1]

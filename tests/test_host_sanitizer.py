"""CPU suite, part 3 (round 5): the workspace-layout sweep of the block-cyclic driver and the HOST-side sanitizer leg.

* test_dist_workspace_layout_sweep -- for every rank of the grids 1x1 ... 2x4 (and 1x8), n in {300 ... 65536}, d in {1, 8, 32},
  dy in {1, 2, 5}, tile in {128 ... 4096} and the three workspaces (forward, backward, refinement): every sub-buffer
  (gpn_dist_layout) is 256-byte aligned, disjoint from the others, ends inside gpn_dist_*_work_bytes, and is at least as large
  as what the driver's calls into it need -- the needs are restated HERE from the geometry (tile (I, J) on rank
  (I mod Pr) Pc + (J mod Pc)) and the public size functions, not read from the layout code.  (Round 4's
  workspace overrun -- the gradient sweep's partial sums sized for the wrong call shape -- was found by eye.)
* test_sanitizer_leg -- the same sweep plus tests/test_abi.py once more in a CHILD process against
  lib/libgpnative_asan.so: the host half of every translation unit built with AddressSanitizer + UBSan
  (gptorch_amd/csrc/build_asan.sh; no device code, no visible device, LD_PRELOAD = clang's ASan runtime,
  PYTHONMALLOC=malloc so that ctypes buffers get redzones).  Signed overflow in the size arithmetic, out-of-bounds writes
  into caller buffers (gpn_mesh_plan's ops, gpn_dist_layout's pairs, gpn_potrf_panel_levels) and misuse of the host
  containers abort the child.  GPU ASan / XNACK runs are not available on this pool: this is the sanitizer coverage there is.
"""
import ctypes
import os
import subprocess
import sys

import pytest

from gptorch_amd import _native

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
LEAF = 128
GRIDS = [(1, 1), (1, 2), (2, 2), (1, 4), (2, 4), (1, 8)]
FWD = ["A", "left0", "left1", "right0", "right1", "diag", "winv", "xrow", "xcol", "stats", "info", "sums"]
BWD = FWD + ["kinv", "alphaT", "aT", "al", "part", "arow", "acol", "gwork", "gout", "acc"]
REF = ["alpha", "a", "owed", "buf", "sj", "aj", "ar", "ka", "U", "S", "W", "winv", "gwork", "rwork"]


def rup(x, m):
    return (x + m - 1) // m * m


def _layout(lib, which, rank, pr, pc, n, d, dy, T):
    buf = (ctypes.c_int64 * 64)()
    cnt = lib.gpn_dist_layout(which, rank, pr, pc, n, d, dy, T, buf, 32)
    names = (FWD, BWD, REF)[which]
    assert cnt == len(names), (which, cnt)
    # a buffer that is too small must be respected (the sanitizer leg would see a write past it)
    small = (ctypes.c_int64 * 4)()
    assert lib.gpn_dist_layout(which, rank, pr, pc, n, d, dy, T, small, 2) == cnt
    assert [small[i] for i in range(4)] == [buf[i] for i in range(4)]
    return {nm: (int(buf[2 * i]), int(buf[2 * i + 1])) for i, nm in enumerate(names)}


def _needs(lib, which, rank, pr, pc, n, d, dy, T):
    """doubles each sub-buffer must hold, from the geometry and the public size functions."""
    my_r, my_c = divmod(rank, pc)
    nt = (n + T - 1) // T
    rows_t = [i for i in range(nt) if i % pr == my_r]
    cols_t = [j for j in range(nt) if j % pc == my_c]
    nrow_t, ncol_t = len(rows_t), len(cols_t)
    has_res = (nt % pr) == my_r
    tile_rows = lambda i: min(T, n - i * T)
    ld = max(ncol_t, 1) * T
    wn = lib.gpn_winv_bytes(T) // 8
    inv = which == 1
    local_rows = nrow_t * T + (rup(dy, LEAF) if has_res else 0) + (nrow_t * T if inv else 0)
    if which in (0, 1):
        need = {"A": local_rows * ld, "left0": local_rows * T, "left1": local_rows * T, "right0": max(ncol_t, 1) * T * T,
                "right1": max(ncol_t, 1) * T * T, "diag": T * T + wn, "winv": wn, "xrow": nrow_t * T * d, "xcol": ncol_t * T * d,
                "stats": 3 * (ncol_t + 1) + dy, "info": nt, "sums": nt + 2}
        if inv:
            kp = rup(dy, 16)
            nrr = (nrow_t - 1) * T + tile_rows(rows_t[-1]) if nrow_t else 0
            gneed = 0
            for lj, J in enumerate(cols_t):          # the gradient sweep's calls (dist.hip, step 4 of the backward)
                nJ = tile_rows(J)
                li0 = len([i for i in rows_t if i <= J - 1])
                r0 = li0 * T
                if r0 >= nrr:
                    continue
                if li0 < nrow_t and rows_t[li0] == J:
                    gneed = max(gneed, lib.gpn_grad_work_bytes(nJ, nJ, d, 0) // 8)
                    r0 += T
                if r0 < nrr:
                    gneed = max(gneed, lib.gpn_grad_work_bytes(nrr - r0, nJ, d, 0) // 8)
            need.update({"kinv": nrow_t * T * ld, "alphaT": dy * n, "aT": dy * n, "al": kp * ld, "part": kp * nrow_t * T,
                         "arow": nrow_t * T * kp, "acol": ncol_t * T * kp, "gwork": gneed, "gout": 2 + d, "acc": 2 + d})
        return need
    ndiag = len([j for j in cols_t if j % pr == my_r])
    ntri = lib.gpn_refine_tile_count(n)
    world = pr * pc
    q0, q1 = ntri * rank // world, ntri * (rank + 1) // world
    lds = rup(n, LEAF)
    gneed = 0
    for J in range(nt):                              # back-substitution: one tile row of L^T against the owed vector per tile column
        nJ = tile_rows(J)
        gneed = max(gneed, lib.gpn_gemv_t_work_bytes(nJ, nJ, dy) // 8)
        if ncol_t:
            gneed = max(gneed, lib.gpn_gemv_t_work_bytes(nJ, ncol_t * T, dy) // 8)
    return {"alpha": dy * nt * T, "a": dy * nt * T, "owed": dy * max(ncol_t, 1) * T, "buf": dy * T, "sj": dy * T, "aj": dy * T,
            "ar": dy * lds, "ka": 2 * dy * lds, "U": T * T, "S": T * T, "W": max(ndiag, 1) * T * T, "winv": wn, "gwork": gneed,
            "rwork": lib.gpn_refine_resid_part_work_bytes(dy, q1 - q0) // 8}


def _sweep(lib, sizes, dims, dys, tiles):
    checked = 0
    totals = (lib.gpn_dist_work_bytes, lib.gpn_dist_grad_work_bytes, lib.gpn_dist_lml_refine_work_bytes)
    for pr, pc in GRIDS:
        for rank in range(pr * pc):
            for n in sizes:
                for T in tiles:
                    for d in dims:
                        for dy in dys:
                            for which in (0, 1, 2):
                                total = totals[which](rank, pr, pc, n, d, dy, T)
                                assert total > 0 and total % 8 == 0
                                lay = _layout(lib, which, rank, pr, pc, n, d, dy, T)
                                need = _needs(lib, which, rank, pr, pc, n, d, dy, T)
                                end = 0
                                for name, (off, size) in lay.items():
                                    ctx = (name, which, rank, pr, pc, n, d, dy, T)
                                    assert off % 32 == 0 and size % 32 == 0, ctx                 # 256-byte granules
                                    assert off >= end, ctx                                       # disjoint, in declaration order
                                    assert size >= need[name], ctx + (size, need[name])          # what its consumers use
                                    end = off + size
                                    assert end * 8 <= total, ctx
                                assert end * 8 == total, (which, rank, pr, pc, n, d, dy, T)
                                checked += 1
    return checked


def _mesh_plan_with_exact_buffers(rccl):
    """gpn_mesh_plan never writes past `cap` quintuples: every plan is produced into a buffer of exactly its size (and once
    into a buffer one quintuple short)."""
    for p in (1, 2, 3, 4, 8):
        for root in range(p):
            for me in range(p):
                for count in (0, 1, 7, 4096, (4 << 20) // 8 + 3, 2048 * 2048):
                    n = rccl.gpn_mesh_plan(p, root, me, count, 4, (4 << 20) // 8, None, 0)
                    assert n >= 0
                    if n:
                        buf = (ctypes.c_int64 * (5 * n))()
                        assert rccl.gpn_mesh_plan(p, root, me, count, 4, (4 << 20) // 8, buf, n) == n
                        for i in range(n):
                            off, length = buf[5 * i + 3], buf[5 * i + 4]
                            assert 0 <= off and length > 0 and off + length <= count
                        if n > 1:
                            short = (ctypes.c_int64 * (5 * (n - 1)))()
                            assert rccl.gpn_mesh_plan(p, root, me, count, 4, (4 << 20) // 8, short, n - 1) == n


def test_dist_workspace_layout_sweep():
    lib = _native.lib()
    checked = _sweep(lib, sizes=(300, 1000, 4097, 8192, 20000, 65536), dims=(1, 8, 32), dys=(1, 2, 5), tiles=(128, 256, 1024, 2048, 4096))
    assert checked == 27 * 6 * 5 * 3 * 3 * 3
    # bad arguments
    buf = (ctypes.c_int64 * 64)()
    assert lib.gpn_dist_layout(3, 0, 1, 1, 1000, 2, 1, 128, buf, 32) == -1
    assert lib.gpn_dist_layout(0, 0, 2, 3, 1000, 2, 1, 128, buf, 32) == -4           # Pr must divide Pc
    assert lib.gpn_dist_layout(0, 9, 2, 4, 1000, 2, 1, 128, buf, 32) == -3           # rank outside the grid
    assert lib.gpn_dist_layout(0, 0, 1, 1, 1000, 2, 1, 100, buf, 32) == -15          # tile % 128
    assert lib.gpn_dist_layout(0, 0, 1, 1, 1000, 2, 1, 128, None, 4) == -9


def test_mesh_plan_respects_the_callers_buffer():
    _mesh_plan_with_exact_buffers(_native.rccl_lib())


def test_size_functions_do_not_overflow_at_the_largest_sizes():
    """every *_work_bytes / geometry function at the sizes of BASELINE configs 3-5 and beyond (int64 arithmetic: under UBSan in
    the sanitizer leg a signed overflow aborts)."""
    lib = _native.lib()
    for n in (32768, 65536, 131072, 1000000):
        assert lib.gpn_factor_rows(n, 5) * lib.gpn_factor_ld(n, 5) * 8 > 0
        assert lib.gpn_lml_backward_work_bytes(n, 2, 32) > 2 * 8 * n * n
        assert lib.gpn_lml_backward_batched_work_bytes(n, 2, 32, 8) == 8 * ((lib.gpn_lml_backward_work_bytes(n, 2, 32) // 8 + 1) // 2 * 2) * 8
        assert lib.gpn_predict_work_bytes(n, 4096, 2) == 4096 * lib.gpn_factor_ld(n, 2) * 8
        assert lib.gpn_block_inverse_bytes(n) > 0 and lib.gpn_lml_refine_work_bytes(n, 3) > 0
        assert lib.gpn_grad_work_bytes(n, n, 64, 1) > 0 and lib.gpn_grad_x2_work_bytes(n, 4096, 64) > 0
        assert lib.gpn_refine_tile_count(n) > 0 and lib.gpn_gemv_t_work_bytes(4096, n, 5) > 0


def _asan_runtime():
    import glob
    cands = sorted(glob.glob("/opt/rocm/lib/llvm/lib/clang/*/lib/linux/libclang_rt.asan-x86_64.so"))
    return cands[-1] if cands else None


def test_sanitizer_leg():
    asan_lib = os.path.join(ROOT, "gptorch_amd", "lib", "libgpnative_asan.so")
    rccl_asan = os.path.join(ROOT, "gptorch_amd", "lib", "libgpnative_rccl_asan.so")
    rt = _asan_runtime()
    # the instrumented build follows the sources: rebuilt here (a few seconds, host code only) when a source is newer than it
    csrc = os.path.join(ROOT, "gptorch_amd", "csrc")
    newest = max(os.path.getmtime(os.path.join(csrc, f)) for f in os.listdir(csrc) if f.endswith((".hip", ".h", ".cpp")))
    newest = max(newest, os.path.getmtime(os.path.join(ROOT, "include", "gpnative.h")))
    if not (os.path.exists(asan_lib) and os.path.exists(rccl_asan)) or os.path.getmtime(asan_lib) < newest:
        subprocess.check_call(["bash", os.path.join(csrc, "build_asan.sh")], stdout=subprocess.DEVNULL)
    assert os.path.exists(asan_lib) and os.path.exists(rccl_asan), "run __graft_entry__.build() (gptorch_amd/csrc/build_asan.sh) first"
    assert rt is not None, "clang's ASan runtime not found under /opt/rocm/lib/llvm"
    if os.environ.get("GPN_SANITIZER_CHILD") == "1":
        pytest.skip("already inside the sanitizer leg")
    env = dict(os.environ, LD_PRELOAD=rt, PYTHONMALLOC="malloc", GPN_LIB=asan_lib, GPN_RCCL_LIB=rccl_asan, GPN_SANITIZER_CHILD="1",
               HIP_VISIBLE_DEVICES="", ROCR_VISIBLE_DEVICES="",
               ASAN_OPTIONS="detect_leaks=0:alloc_dealloc_mismatch=0:detect_odr_violation=0:abort_on_error=0:exitcode=97",
               UBSAN_OPTIONS="print_stacktrace=1:halt_on_error=1:exitcode=98")
    out = subprocess.run([sys.executable, "-m", "pytest", "-x", "-q", "-p", "no:cacheprovider", os.path.join(ROOT, "tests", "test_abi.py"),
                          os.path.join(ROOT, "tests", "test_host_sanitizer.py")], capture_output=True, text=True, env=env, cwd=ROOT, timeout=1500)
    tail = (out.stdout[-3000:] + "\n" + out.stderr[-3000:])
    assert out.returncode == 0, "sanitizer leg failed (exit code %d):\n%s" % (out.returncode, tail)
    assert "passed" in out.stdout and "AddressSanitizer" not in tail and "runtime error" not in tail, tail
    # the child really ran against the instrumented library
    probe = subprocess.run([sys.executable, "-c", "from gptorch_amd import _native; import ctypes; _native.lib(); "
                            "print(open('/proc/self/maps').read().count('libgpnative_asan.so') > 0)"],
                           capture_output=True, text=True, env=env, cwd=ROOT, timeout=300)
    assert probe.stdout.strip().endswith("True"), probe.stdout + probe.stderr
    # ... and the instrumentation is LIVE: a deliberate overrun of a caller buffer (32 pairs asked for, room for 2) is caught
    bad = subprocess.run([sys.executable, "-c", "import ctypes; from gptorch_amd import _native; lib = _native.lib(); "
                          "b = (ctypes.c_int64 * 4)(); lib.gpn_dist_layout(1, 0, 1, 1, 1000, 2, 1, 128, b, 32); print('survived')"],
                         capture_output=True, text=True, env=env, cwd=ROOT, timeout=300)
    assert bad.returncode != 0 and "survived" not in bad.stdout and "AddressSanitizer" in bad.stderr, (bad.returncode, bad.stderr[-800:])

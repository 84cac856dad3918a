# -*- coding: utf-8 -*-
"""CPU-side checks of the drop-in boundary: the C-ABI library loads and exports every symbol that
include/celerite2_amd.h declares, the pybind11 modules expose the reference's surface
(python/celerite2/driver.cpp:482-499, backprop.cpp:906-926) and its argument validation, and the
product path fails loudly (no CPU fallback) when no GPU is present."""
import ctypes
import os
import re
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def built():
    from celerite2_amd import build

    build.build_all()
    return build


def test_header_symbols_exported(built):
    from celerite2_amd import _lib

    header = open(os.path.join(ROOT, "include", "celerite2_amd.h")).read()
    declared = set(re.findall(r"\b(c2h?_[a-z0-9_]+)\s*\(", header))
    assert declared == set(_lib.SYMBOLS), declared ^ set(_lib.SYMBOLS)
    lib = ctypes.CDLL(_lib.LIB_PATH)
    for name in sorted(declared):
        assert hasattr(lib, name), name
    lib.c2_version.restype = ctypes.c_char_p
    assert b"gfx950" in lib.c2_version()


def test_gfx950_code_object(built):
    """The shared library must carry a gfx950 code object (hipcc --offload-arch=gfx950)."""
    data = open(built.LIB).read() if False else open(built.LIB, "rb").read()
    assert b"gfx950" in data


def test_argument_errors_without_gpu(built):
    from celerite2_amd import _lib

    lib = _lib.load()
    i64 = ctypes.c_int64
    null = ctypes.c_void_p(0)
    # invalid sizes / null pointers are rejected before anything touches the device
    assert lib.c2_loglik(i64(0), i64(4), i64(2), *([null] * 1), i64(0), null, i64(0), null, null, null, null, null, null,
                         null) == _lib.C2_ERR_INVALID
    assert lib.c2_loglik(i64(1), i64(4), i64(129), null, i64(0), null, i64(0), null, null, null, null, null, null,
                         null) == _lib.C2_ERR_UNSUPPORTED   # beyond C2_MAX_WIDTH = 128 (33 .. 128: csrc/c2_wide.hip)
    # workspace = wave-blocked packed checkpoints (one per segment + the state after the last row) + the W rows (B,N,J)
    # + (d,z) pairs (B,N,2) + one stability word per wavefront, each part padded to 16 bytes  
    B, N, J = 2, 8, 3                      # G = 4, C = 8 -> 1 segment, 1 wavefront
    ck = 1 * (1 + 1) * (64 + 3 * 32 + 64 + 64)   # S slot 0 (64) + slots 1..3 (32 owners each) + F (64) + W (64)
    assert lib.c2_loglik_grad_workspace_bytes(B, N, J) == 8 * (ck + 1 * N * 64 + B * N * 2 + 2)   # W: lane-major per wavefront
    # chip-filling J = 8 batches take the one-lane-per-series path: records W (B,N,8) + (d,z) (B,N,2) + t (B,N) + a
    # checkpoint of 44 doubles every 32 rows and twice as many extra slots (re-anchoring in front of gaps in time) with
    # their row list, overlaid with the replay kernels' workspace, + the guard words (two head words and one per wavefront:
    # a wavefront that runs out of slots sends ITS 64 series to the replay kernels)
    waves, nck = 65536 // 64, 3 * ((4096 - 2) // 32 + 1)
    rec = waves * 64 * (4096 * 8 + 4096 * 2 + 4096 + nck * 44) + waves * (4096 // 2)   # + the slot of every row (int32)
    assert lib.c2_loglik_grad_workspace_bytes(65536, 4096, 8) == 8 * (2 + waves + rec) < 34 * 2**30
    assert lib.c2_loglik_grad_workspace_bytes(65536, 4096, 6) == 8 * (2 + waves + rec)   # width 6 runs as 8: same records
    assert lib.c2_loglik_grad_workspace_bytes(1, 4096, 129) == 0   # unsupported width
    # a wide model (33 .. 128) runs the literal op chain: d, W, S, z, F, bd, bz, bW in the workspace
    assert lib.c2_loglik_grad_workspace_bytes(2, 100, 40) == 8 * 2 * 100 * (1 + 40 + 1600 + 1 + 40 + 1 + 1 + 40)


def test_driver_surface_and_validation(built):
    from celerite2_amd import backprop, driver

    for name in ("factor", "solve_lower", "solve_upper", "matmul_lower", "matmul_upper", "general_matmul_lower",
                 "general_matmul_upper", "get_celerite_matrices", "LinAlgError", "__version__"):
        assert hasattr(driver, name), name
    for name in ("factor_fwd", "factor_rev", "solve_lower_fwd", "solve_lower_rev", "solve_upper_fwd",
                 "solve_upper_rev", "matmul_lower_fwd", "matmul_lower_rev", "matmul_upper_fwd", "matmul_upper_rev",
                 "general_matmul_lower_fwd", "general_matmul_upper_fwd", "LinAlgError"):
        assert hasattr(backprop, name), name
    assert driver.LinAlgError is not backprop.LinAlgError  # separate classes, as upstream
    N, J = 5, 2
    t, c, a = np.zeros(N), np.zeros(J), np.ones(N)
    U, V = np.zeros((N, J)), np.zeros((N, J))
    with pytest.raises(ValueError, match="Invalid shape: a"):
        driver.factor(t, c, np.ones(N + 1), U, V, a, V)
    with pytest.raises(ValueError, match="Invalid shape: W"):
        driver.factor(t, c, a, U, V, a, np.zeros((N, J + 1)))
    with pytest.raises(ValueError, match="Invalid number of dimensions: Y"):
        driver.solve_lower(t, c, U, V, np.zeros(N), np.zeros(N))  # 1-D Y rejected (driver.cpp:89-91)
    with pytest.raises(ValueError, match="Invalid shape: S"):
        backprop.factor_fwd(t, c, a, U, V, a, V, np.zeros((N, J)))
    with pytest.raises(ValueError, match="dimension mismatch: bc"):
        driver.get_celerite_matrices(np.zeros(1), np.zeros(1), np.zeros(2), np.zeros(1), t, a, a, np.zeros((N, 3)),
                                     np.zeros((N, 3)))


def test_fails_loudly_without_gpu(built):
    """No silent CPU fallback: without a HIP device the product raises."""
    from celerite2_amd import _lib, driver

    if _lib.load().c2_device_count() > 0:
        pytest.skip("a GPU is visible here")
    N, J = 5, 2
    t = np.arange(N, dtype=float)
    with pytest.raises(RuntimeError, match="HIP error"):
        driver.factor(t, np.ones(J), np.ones(N), np.zeros((N, J)), np.zeros((N, J)), np.ones(N), np.zeros((N, J)))


def test_dispatch_options_table(built):
    """One option table (csrc/c2_dispatch.hpp): the environment is read once at load, c2_set_option changes an option at
    run time, no kernel source calls getenv, and INTEGRATION.md section 5 is the table the library was compiled with."""
    import glob
    import subprocess

    from celerite2_amd import _lib

    lib = _lib.load()
    opts = {o["name"]: o for o in _lib.options()}
    assert {"lanes", "timepar", "timepar_grad", "factor_iter", "lanes1_min_batch_grad", "timepar_cond_limit"} <= set(opts)
    assert opts["lanes1_min_batch_grad"]["default"] == 24576 and not opts["lanes"]["is_set"]
    _lib.set_option("lanes", 1)
    v, st = ctypes.c_double(), ctypes.c_int()
    assert lib.c2_get_option(b"C2_LANES", ctypes.byref(v), ctypes.byref(st)) == _lib.C2_OK and v.value == 1.0 and st.value == 1
    # the workspace query follows the option (one lane per series: records + the 16-byte guard)
    big = lib.c2_loglik_grad_workspace_bytes(128, 64, 8)
    _lib.set_option("lanes", None)
    assert lib.c2_get_option(b"lanes", ctypes.byref(v), ctypes.byref(st)) == _lib.C2_OK and st.value == 0
    assert lib.c2_loglik_grad_workspace_bytes(128, 64, 8) != big
    assert lib.c2_set_option(b"no_such_option", b"1") == _lib.C2_ERR_INVALID
    assert lib.c2_set_option(b"lanes", b"x") == _lib.C2_ERR_INVALID
    for src in glob.glob(os.path.join(ROOT, "celerite2_amd", "csrc", "*")):
        if os.path.basename(src) != "c2_dispatch.hip":
            assert "getenv(" not in open(src).read(), src
    assert subprocess.call([sys.executable, os.path.join(ROOT, "tools", "gen_dispatch_doc.py"), "--check"]) == 0, \
        "INTEGRATION.md section 5 is stale: run python tools/gen_dispatch_doc.py"

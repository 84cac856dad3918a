# -*- coding: utf-8 -*-
"""ctypes binding of the C-ABI (include/celerite2_amd.h).

There is NO CPU fallback: if libcelerite2_amd.so is missing or no HIP device is
visible, the product path raises -- it never routes through oracle/.
"""
import ctypes
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("C2_LIB_PATH", os.path.join(_HERE, "libcelerite2_amd.so"))  # override: A/B builds

C2_OK, C2_ERR_INVALID, C2_ERR_UNSUPPORTED, C2_ERR_HIP = 0, -1, -2, -3
C2_MAX_WIDTH = 32

# Every symbol include/celerite2_amd.h declares (checked by tests/test_abi.py).
SYMBOLS = [
    "c2_version", "c2_device_count", "c2_last_error",
    "c2_factor", "c2_solve_lower", "c2_solve_upper", "c2_matmul_lower", "c2_matmul_upper",
    "c2_general_matmul_lower", "c2_general_matmul_upper", "c2_factor_rev",
    "c2_solve_lower_rev", "c2_solve_upper_rev", "c2_matmul_lower_rev", "c2_matmul_upper_rev",
    "c2_get_celerite_matrices", "c2_loglik", "c2_loglik_grad_workspace_bytes", "c2_loglik_grad", "c2_dot_tril",
    "c2_kron_loglik_workspace_bytes", "c2_kron_loglik", "c2_kron_loglik_grad",
    "c2_loglik_terms_workspace_bytes", "c2_loglik_terms", "c2_loglik_terms_grad",
    "c2h_factor", "c2h_solve_lower", "c2h_solve_upper", "c2h_matmul_lower", "c2h_matmul_upper",
    "c2h_general_matmul_lower", "c2h_general_matmul_upper", "c2h_factor_rev",
    "c2h_solve_lower_rev", "c2h_solve_upper_rev", "c2h_matmul_lower_rev", "c2h_matmul_upper_rev",
    "c2h_get_celerite_matrices",
]

_lib = None


class BackendError(RuntimeError):
    pass


def load():
    """Load libcelerite2_amd.so (after torch, so both share one HIP runtime)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise BackendError(
            "celerite2_amd: %s not found -- build it with `python -m celerite2_amd.build` "
            "(hipcc --offload-arch=gfx950). There is no CPU fallback." % LIB_PATH
        )
    try:  # torch bundles libamdhip64.so.7; importing it first makes the soname resolve to that copy
        import torch  # noqa: F401
    except Exception:  # pragma: no cover - torch is plumbing, not required for the C-ABI itself
        pass
    lib = ctypes.CDLL(LIB_PATH)
    lib.c2_version.restype = ctypes.c_char_p
    lib.c2_last_error.restype = ctypes.c_char_p
    lib.c2_device_count.restype = ctypes.c_int
    lib.c2_loglik_grad_workspace_bytes.restype = ctypes.c_size_t
    lib.c2_loglik_grad_workspace_bytes.argtypes = [ctypes.c_int64] * 3
    lib.c2_loglik_terms_workspace_bytes.restype = ctypes.c_size_t
    lib.c2_loglik_terms_workspace_bytes.argtypes = [ctypes.c_int64] * 4 + [ctypes.c_int]
    lib.c2_kron_loglik_workspace_bytes.restype = ctypes.c_size_t
    lib.c2_kron_loglik_workspace_bytes.argtypes = [ctypes.c_int64] * 4 + [ctypes.c_int] * 2
    _lib = lib
    return lib


def check(rc, what):
    if rc == C2_OK:
        return
    lib = load()
    if rc == C2_ERR_INVALID:
        raise ValueError("celerite2_amd.%s: invalid shape / null argument" % what)
    if rc == C2_ERR_UNSUPPORTED:
        raise ValueError("celerite2_amd.%s: J exceeds C2_MAX_WIDTH=%d" % (what, C2_MAX_WIDTH))
    raise BackendError("celerite2_amd.%s: HIP error: %s" % (what, lib.c2_last_error().decode()))

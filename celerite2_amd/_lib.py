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
C2_MAX_WIDTH = 128

# Every symbol include/celerite2_amd.h declares (checked by tests/test_abi.py).
SYMBOLS = [
    "c2_version", "c2_device_count", "c2_last_error",
    "c2_factor", "c2_solve_lower", "c2_solve_upper", "c2_matmul_lower", "c2_matmul_upper",
    "c2_general_matmul_lower", "c2_general_matmul_upper", "c2_factor_rev",
    "c2_solve_lower_rev", "c2_solve_upper_rev", "c2_matmul_lower_rev", "c2_matmul_upper_rev",
    "c2_get_celerite_matrices", "c2_kernel_values", "c2_colsumsq_over_d", "c2_loglik", "c2_loglik_grad_workspace_bytes", "c2_loglik_grad", "c2_condition", "c2_dot_tril",
    "c2_kron_loglik_workspace_bytes", "c2_kron_loglik", "c2_kron_loglik_grad",
    "c2_loglik_terms_workspace_bytes", "c2_loglik_terms", "c2_loglik_terms_grad",
    "c2h_factor", "c2h_solve_lower", "c2h_solve_upper", "c2h_matmul_lower", "c2h_matmul_upper",
    "c2h_general_matmul_lower", "c2h_general_matmul_upper", "c2h_factor_rev",
    "c2h_solve_lower_rev", "c2h_solve_upper_rev", "c2h_matmul_lower_rev", "c2h_matmul_upper_rev",
    "c2h_get_celerite_matrices", "c2h_release_thread_cache",
    "c2_set_option", "c2_get_option", "c2_option_count", "c2_option_info", "c2_options_reload_env",
]

_lib = None
_env_names = ()      # environment variables of the option table (c2_option_info)
_env_seen = None     # their values when the table was last (re)loaded


class BackendError(RuntimeError):
    pass


def _sync_env(lib):
    """The C library reads its dispatch options from the environment ONCE, when it is loaded, and never calls getenv
    afterwards (c2_dispatch.hpp).  This ctypes layer additionally follows the process environment when it SEES one of the
    table's variables change between two calls -- what the test-suite and the A/B tools rely on when they edit os.environ
    at run time -- and applies exactly the variables that changed (c2_set_option each), so values an application set
    through set_option() for OTHER options survive an unrelated monkeypatch.setenv.  Applications use set_option()."""
    global _env_seen
    now = tuple(os.environ.get(n) for n in _env_names)
    if now != _env_seen:
        for name, old, new in zip(_env_names, _env_seen or (None,) * len(_env_names), now):
            if old != new:
                # unset or unparsable -> back to the table's default, as at load time
                if lib.c2_set_option(name.encode(), None if not new else new.encode()) != C2_OK:
                    lib.c2_set_option(name.encode(), None)
        _env_seen = now


def set_option(name, value=None):
    """c2_set_option: `name` = option name or its environment variable; value None = back to the default / automatic."""
    lib = load()
    rc = lib.c2_set_option(str(name).encode(), None if value is None else str(value).encode())
    if rc != C2_OK:
        raise ValueError("celerite2_amd: unknown option %r or unparsable value %r" % (name, value))


def options():
    """The dispatch table: a list of dicts (name, env, default, switch, value, is_set, doc, measured)."""
    lib = load()
    out = []
    for i in range(lib.c2_option_count()):
        name, env, doc, src = (ctypes.c_char_p() for _ in range(4))
        dflt, sw = ctypes.c_double(), ctypes.c_int()
        lib.c2_option_info(i, ctypes.byref(name), ctypes.byref(env), ctypes.byref(dflt), ctypes.byref(sw), ctypes.byref(doc),
                           ctypes.byref(src))
        val, st = ctypes.c_double(), ctypes.c_int()
        lib.c2_get_option(name.value, ctypes.byref(val), ctypes.byref(st))
        out.append(dict(name=name.value.decode(), env=env.value.decode(), default=dflt.value, switch=bool(sw.value),
                        value=val.value, is_set=bool(st.value), doc=doc.value.decode(), measured=src.value.decode()))
    return out


def load():
    """Load libcelerite2_amd.so (after torch, so both share one HIP runtime)."""
    global _lib, _env_names, _env_seen
    if _lib is not None:
        _sync_env(_lib)
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
    lib.c2_set_option.argtypes = [ctypes.c_char_p, ctypes.c_char_p]
    lib.c2_get_option.argtypes = [ctypes.c_char_p, ctypes.POINTER(ctypes.c_double), ctypes.POINTER(ctypes.c_int)]
    lib.c2_option_count.restype = ctypes.c_int
    lib.c2_option_info.argtypes = [ctypes.c_int, ctypes.POINTER(ctypes.c_char_p), ctypes.POINTER(ctypes.c_char_p),
                                   ctypes.POINTER(ctypes.c_double), ctypes.POINTER(ctypes.c_int),
                                   ctypes.POINTER(ctypes.c_char_p), ctypes.POINTER(ctypes.c_char_p)]
    lib.c2_options_reload_env.restype = None
    _lib = lib
    names = []
    for i in range(lib.c2_option_count()):
        env = ctypes.c_char_p()
        lib.c2_option_info(i, None, ctypes.byref(env), None, None, None, None)
        names.append(env.value.decode())
    _env_names = tuple(names)
    _env_seen = tuple(os.environ.get(n) for n in _env_names)
    return lib


def check(rc, what):
    if rc == C2_OK:
        return
    lib = load()
    if rc == C2_ERR_INVALID:
        raise ValueError("celerite2_amd.%s: invalid shape / null argument" % what)
    if rc == C2_ERR_UNSUPPORTED:
        raise ValueError("celerite2_amd.%s: width not supported (J <= %d; the 2-D and coefficient-level extensions: J <= 32)"
                         % (what, C2_MAX_WIDTH))
    raise BackendError("celerite2_amd.%s: HIP error: %s" % (what, lib.c2_last_error().decode()))

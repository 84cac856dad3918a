# -*- coding: utf-8 -*-
"""oracle/cpu.py -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.

ctypes loader for libc2_oracle.so (the CPU restatement in c2_oracle.cpp) with
numpy-level wrappers that mirror the reference's `celerite2.driver` /
`celerite2.backprop` signatures (python/celerite2/driver.cpp:13-499,
python/celerite2/backprop.cpp:12-926): caller-allocated outputs, in-place
aliasing allowed, outputs returned as the same array objects.
"""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "libc2_oracle.so")

_i64 = ctypes.c_int64
_dp = ctypes.c_void_p


def build(force=False):
    src = os.path.join(_HERE, "c2_oracle.cpp")
    if force or not os.path.exists(_SO) or os.path.getmtime(_SO) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "-B" if force else "-s"])
    return _SO


_lib = None
_so_in_use = _SO
_flags_in_use = "-O3 -march=x86-64-v3 -fopenmp"


def build_native():
    """Timing build for bench.py's cpu_baseline leg: the same source with -march=native, compiled ON the machine
    that times it, into oracle/libc2_oracle_native.so.  Falls back to the portable build
    (x86-64-v3) when no compiler is there.  Must be called before the first lib()."""
    global _so_in_use, _flags_in_use
    src, out = os.path.join(_HERE, "c2_oracle.cpp"), os.path.join(_HERE, "libc2_oracle_native.so")
    flags = ["-O3", "-march=native", "-std=c++17", "-fPIC", "-fopenmp"]
    try:
        if not os.path.exists(out) or os.path.getmtime(out) < os.path.getmtime(src):
            subprocess.check_call(["g++"] + flags + ["-shared", "-o", out, src])
        assert _lib is None, "build_native() must precede the first use of the oracle"
        _so_in_use, _flags_in_use = out, "-O3 -march=native -fopenmp"
    except Exception:
        build()
    return _so_in_use


def build_flags():
    return _flags_in_use


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_so_in_use):
            build()
        _lib = ctypes.CDLL(_so_in_use)
        _lib.c2o_factor.restype = _i64
        _lib.c2o_loglik.restype = _i64
        _lib.c2o_loglik_grad.restype = _i64
        _lib.c2o_loglik_grad_work_size.restype = _i64
        _lib.c2o_num_threads.restype = ctypes.c_int
    return _lib


class LinAlgError(Exception):
    pass


def _p(x):
    if x is None:
        return _dp(0)
    assert x.dtype == np.float64 and x.flags.c_contiguous, "oracle wrappers need C-contiguous float64"
    return _dp(x.ctypes.data)


def _dims(t, c, Y=None):
    N, J = t.shape[0], c.shape[0]
    if Y is None:
        return _i64(N), _i64(J)
    assert Y.ndim == 2
    return _i64(N), _i64(J), _i64(Y.shape[1])


def factor(t, c, a, U, V, d, W, S=None):
    flag = lib().c2o_factor(*_dims(t, c), _p(t), _p(c), _p(a), _p(U), _p(V), _p(d), _p(W), _p(S))
    if flag:
        raise LinAlgError("failed to factorize or solve matrix (row %d)" % flag)
    return (d, W) if S is None else (d, W, S)


def factor_flag(t, c, a, U, V, d, W, S=None):
    return int(lib().c2o_factor(*_dims(t, c), _p(t), _p(c), _p(a), _p(U), _p(V), _p(d), _p(W), _p(S)))


def _sweep(name):
    def f(t, c, U, W, Y, Z, F=None):
        getattr(lib(), "c2o_" + name)(*_dims(t, c, Y), _p(t), _p(c), _p(U), _p(W), _p(Y), _p(Z), _p(F))
        return Z if F is None else (Z, F)
    f.__name__ = name
    return f


solve_lower = _sweep("solve_lower")
solve_upper = _sweep("solve_upper")
matmul_lower = _sweep("matmul_lower")
matmul_upper = _sweep("matmul_upper")


def _sweep_fwd(name):
    """backprop.*_fwd: Z is zeroed first (backprop.cpp:201,207,505,511), then the op with workspace."""
    def f(t, c, U, W, Y, Z, F):
        Z[...] = 0.0
        getattr(lib(), "c2o_" + name)(*_dims(t, c, Y), _p(t), _p(c), _p(U), _p(W), _p(Y), _p(Z), _p(F))
        return Z, F
    f.__name__ = name + "_fwd"
    return f


solve_lower_fwd = _sweep_fwd("solve_lower")
solve_upper_fwd = _sweep_fwd("solve_upper")
matmul_lower_fwd = _sweep_fwd("matmul_lower")
matmul_upper_fwd = _sweep_fwd("matmul_upper")


def factor_fwd(t, c, a, U, V, d, W, S):
    return factor(t, c, a, U, V, d, W, S)


def _general(name):
    def f(t1, t2, c, U, V, Y, Z, F=None):
        N, M, J, nrhs = t1.shape[0], t2.shape[0], c.shape[0], Y.shape[1]
        getattr(lib(), "c2o_" + name)(_i64(N), _i64(M), _i64(J), _i64(nrhs), _p(t1), _p(t2), _p(c), _p(U), _p(V),
                                      _p(Y), _p(Z), _p(F))
        return Z if F is None else (Z, F)
    f.__name__ = name
    return f


general_matmul_lower = _general("general_matmul_lower")
general_matmul_upper = _general("general_matmul_upper")


def general_matmul_lower_fwd(t1, t2, c, U, V, Y, Z, F):
    Z[...] = 0.0
    return general_matmul_lower(t1, t2, c, U, V, Y, Z, F)


def general_matmul_upper_fwd(t1, t2, c, U, V, Y, Z, F):
    Z[...] = 0.0
    return general_matmul_upper(t1, t2, c, U, V, Y, Z, F)


def factor_rev(t, c, a, U, V, d, W, S, bd, bW, bt, bc, ba, bU, bV):
    lib().c2o_factor_rev(*_dims(t, c), _p(t), _p(c), _p(a), _p(U), _p(V), _p(d), _p(W), _p(S), _p(bd), _p(bW),
                         _p(bt), _p(bc), _p(ba), _p(bU), _p(bV))
    return bt, bc, ba, bU, bV


def _sweep_rev(name):
    def f(t, c, U, W, Y, Z, F, bZ, bt, bc, bU, bW, bY):
        getattr(lib(), "c2o_" + name)(*_dims(t, c, Y), _p(t), _p(c), _p(U), _p(W), _p(Y), _p(Z), _p(F), _p(bZ),
                                      _p(bt), _p(bc), _p(bU), _p(bW), _p(bY))
        return bt, bc, bU, bW, bY
    f.__name__ = name
    return f


solve_lower_rev = _sweep_rev("solve_lower_rev")
solve_upper_rev = _sweep_rev("solve_upper_rev")
matmul_lower_rev = _sweep_rev("matmul_lower_rev")
matmul_upper_rev = _sweep_rev("matmul_upper_rev")


def get_celerite_matrices(ar, ac, bc, dc, x, diag, a, U, V):
    lib().c2o_get_celerite_matrices(_i64(len(ar)), _i64(len(ac)), _i64(len(x)), _p(ar), _p(ac), _p(bc), _p(dc),
                                    _p(x), _p(diag), _p(a), _p(U), _p(V))
    return a, U, V


def loglik(t, c, a, U, V, y):
    N, J = t.shape[0], c.shape[0]
    ll = np.zeros(1)
    d = np.empty(N); W = np.empty((N, J)); z = np.empty(N)
    flag = lib().c2o_loglik(_i64(N), _i64(J), _p(t), _p(c), _p(a), _p(U), _p(V), _p(y), _p(ll), _p(d), _p(W), _p(z))
    return float(ll[0]), int(flag)


def loglik_grad(t, c, a, U, V, y):
    N, J = t.shape[0], c.shape[0]
    ll = np.zeros(1)
    bt = np.empty(N); bc = np.empty(J); ba = np.empty(N); bU = np.empty((N, J)); bV = np.empty((N, J)); by = np.empty(N)
    work = np.empty(lib().c2o_loglik_grad_work_size(_i64(N), _i64(J)))
    flag = lib().c2o_loglik_grad(_i64(N), _i64(J), _p(t), _p(c), _p(a), _p(U), _p(V), _p(y), _p(ll), _p(bt), _p(bc),
                                 _p(ba), _p(bU), _p(bV), _p(by), _p(work))
    return float(ll[0]), (bt, bc, ba, bU, bV, by), int(flag)


def _bs(x, per):
    """batch stride (in elements) of a (B, per) or shared (per,) array."""
    return 0 if x.ndim == 1 else per


def loglik_batched(t, c, a, U, V, y, nthreads=0):
    B, N = y.shape
    J = c.shape[-1]
    ll = np.empty(B); flag = np.empty(B, dtype=np.int32)
    lib().c2o_loglik_batched(_i64(B), _i64(N), _i64(J), _p(t), _i64(_bs(t, N)), _p(c), _i64(_bs(c, J)), _p(a), _p(U),
                             _p(V), _p(y), _p(ll), ctypes.c_void_p(flag.ctypes.data), ctypes.c_int(nthreads))
    return ll, flag


def loglik_grad_batched(t, c, a, U, V, y, nthreads=0):
    B, N = y.shape
    J = c.shape[-1]
    ll = np.empty(B); flag = np.empty(B, dtype=np.int32)
    bt = np.empty((B, N)); bc = np.empty((B, J)); ba = np.empty((B, N))
    bU = np.empty((B, N, J)); bV = np.empty((B, N, J)); by = np.empty((B, N))
    lib().c2o_loglik_grad_batched(_i64(B), _i64(N), _i64(J), _p(t), _i64(_bs(t, N)), _p(c), _i64(_bs(c, J)), _p(a),
                                  _p(U), _p(V), _p(y), _p(ll), _p(bt), _p(bc), _p(ba), _p(bU), _p(bV), _p(by),
                                  ctypes.c_void_p(flag.ctypes.data), ctypes.c_int(nthreads))
    return ll, (bt, bc, ba, bU, bV, by), flag


_lib_ld = None


def loglik_grad_batched_ld(t, c, a, U, V, y, nthreads=0):
    """loglik_grad_batched evaluated in extended precision (long double: the same source, oracle/Makefile) on float64
    inputs; results rounded back to float64.  How far the float64 restatement is from THIS is the floor below which no
    float64 evaluation order can be asked to agree with it."""
    global _lib_ld
    so = os.path.join(_HERE, "libc2_oracle_ld.so")
    if _lib_ld is None:
        if not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(os.path.join(_HERE, "c2_oracle.cpp")):
            subprocess.check_call(["make", "-C", _HERE, "-s", "libc2_oracle_ld.so"])
        _lib_ld = ctypes.CDLL(so)
    B, N = y.shape
    J = c.shape[-1]
    L = np.longdouble
    assert np.dtype(L).itemsize == 16, "x86-64 long double"
    ld = lambda x: np.ascontiguousarray(x, dtype=L)
    p = lambda x: _dp(x.ctypes.data)
    tl, cl, al, Ul, Vl, yl = (ld(x) for x in (t, c, a, U, V, y))
    ll = np.empty(B, L); flag = np.empty(B, dtype=np.int32)
    bt = np.empty((B, N), L); bc = np.empty((B, J), L); ba = np.empty((B, N), L)
    bU = np.empty((B, N, J), L); bV = np.empty((B, N, J), L); by = np.empty((B, N), L)
    _lib_ld.c2o_ld_loglik_grad_batched(_i64(B), _i64(N), _i64(J), p(tl), _i64(_bs(t, N)), p(cl), _i64(_bs(c, J)), p(al),
                                       p(Ul), p(Vl), p(yl), p(ll), p(bt), p(bc), p(ba), p(bU), p(bV), p(by),
                                       ctypes.c_void_p(flag.ctypes.data), ctypes.c_int(nthreads))
    f = lambda x: np.asarray(x, dtype=np.float64)
    return f(ll), tuple(f(x) for x in (bt, bc, ba, bU, bV, by)), flag


def num_threads():
    return int(lib().c2o_num_threads())

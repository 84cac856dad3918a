# -*- coding: utf-8 -*-
"""oracle/ref_shim.py -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.  BUILD CONTAINER ONLY.

Makes the reference's own pure-Python modules importable IN THIS CONTAINER, straight from where they lie
(/root/reference/python/celerite2/{terms,core,numpy,testing}.py are READ at import time; nothing is copied into the
repo and nothing of this travels to the GPU box, which has no /root/reference):

  * an empty package object `celerite2` whose `__path__` is the reference's package directory (its `__init__.py`
    is NOT executed: it needs the generated `celerite2_version.py`),
  * `celerite2.driver` / `celerite2.backprop` as modules whose functions are the CPU restatement's wrappers
    (oracle/cpu.py: same names, argument order, same-object returns and in-place behaviour as
    python/celerite2/driver.cpp:13-499 / backprop.cpp:12-926) -- the only compiled pieces the Python files need
    (terms.py:22, numpy.py:9-11).

With that in place
  * `celerite2.terms` is REFERENCE code: `get_coefficients` (terms.py:515-521, 554-569, 658-691, 729-745, 791-812),
    `get_value` (:58-79), `to_dense` (:106-115) and the interleaved `c` of `get_celerite_matrices` (:171-173)
    run as written upstream and never enter the shim;
  * `celerite2.numpy.GaussianProcess` / `core.py` are REFERENCE callers (log-likelihood assembly numpy.py:66-109,
    ConditionalDistribution core.py:9-150) executing over the restatement;
  * the reference's acceptance suites python/test/test_driver.py / test_backprop.py run unmodified
    (tools/ref_acceptance.py).

Users: tests/golden/make_golden_ref.py, tools/ref_acceptance.py, tests/test_oracle.py (skipped where the reference is
absent).  The product never imports this file.
"""
import importlib
import os
import sys
import types

REF_PKG = "/root/reference/python/celerite2"
REF_TESTS = "/root/reference/python/test"


def available():
    return os.path.isfile(os.path.join(REF_PKG, "terms.py"))


def _driver_module(cpu):
    m = types.ModuleType("celerite2.driver")
    m.__doc__ = "shim: oracle/cpu.py behind the names of python/celerite2/driver.cpp:479-499"
    m.LinAlgError = cpu.LinAlgError
    m.BackpropLinAlgError = cpu.LinAlgError
    for name in ("factor", "solve_lower", "solve_upper", "matmul_lower", "matmul_upper", "general_matmul_lower",
                 "general_matmul_upper", "get_celerite_matrices"):
        setattr(m, name, getattr(cpu, name))
    return m


def _backprop_module(cpu):
    m = types.ModuleType("celerite2.backprop")
    m.__doc__ = "shim: oracle/cpu.py behind the names of python/celerite2/backprop.cpp:899-926"
    m.LinAlgError = cpu.LinAlgError
    for name in ("factor_fwd", "factor_rev", "solve_lower_fwd", "solve_lower_rev", "solve_upper_fwd", "solve_upper_rev",
                 "matmul_lower_fwd", "matmul_lower_rev", "matmul_upper_fwd", "matmul_upper_rev",
                 "general_matmul_lower_fwd", "general_matmul_upper_fwd"):
        setattr(m, name, getattr(cpu, name))
    return m


def install(driver=None, backprop=None):
    """Register the shim in sys.modules and return the reference's modules as a namespace:
    ns.terms, ns.core, ns.numpy (GaussianProcess), ns.testing, ns.driver, ns.backprop.
    `driver` / `backprop`: modules to bind instead of the oracle-backed ones (e.g. the product's pybind11 modules on a
    machine that has both a GPU and the reference)."""
    if not available():
        raise RuntimeError("the reference is not mounted at %s (this only works in the build container)" % REF_PKG)
    for k in [k for k in sys.modules if k == "celerite2" or k.startswith("celerite2.")]:
        del sys.modules[k]
    from oracle import cpu

    cpu.build()
    pkg = types.ModuleType("celerite2")
    pkg.__path__ = [REF_PKG]
    pkg.__version__ = "0.0.0+shim"
    sys.modules["celerite2"] = pkg
    drv = driver if driver is not None else _driver_module(cpu)
    bp = backprop if backprop is not None else _backprop_module(cpu)
    sys.modules["celerite2.driver"] = drv
    sys.modules["celerite2.backprop"] = bp
    pkg.driver, pkg.backprop = drv, bp
    ns = types.SimpleNamespace(driver=drv, backprop=bp)
    for name in ("terms", "core", "numpy", "testing"):
        mod = importlib.import_module("celerite2." + name)
        assert os.path.dirname(os.path.abspath(mod.__file__)) == REF_PKG, mod.__file__
        setattr(pkg, name, mod)
        setattr(ns, name, mod)
    pkg.GaussianProcess = ns.numpy.GaussianProcess
    return ns


def uninstall():
    for k in [k for k in sys.modules if k == "celerite2" or k.startswith("celerite2.")]:
        del sys.modules[k]

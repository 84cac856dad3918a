# -*- coding: utf-8 -*-
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with `-m gpu` on the GPU box)")


@pytest.fixture(scope="session")
def oracle():
    """The CPU restatement (test infrastructure only)."""
    from oracle import cpu

    cpu.build()
    return cpu


@pytest.fixture(scope="session")
def golden():
    import numpy as np

    path = os.path.join(ROOT, "tests", "golden", "golden.npz")
    return dict(np.load(path))


@pytest.fixture(scope="session")
def golden_kron():
    """Dense-Kronecker fixtures of the 2-D extension (tests/golden/make_golden_kron.py)."""
    import numpy as np

    return dict(np.load(os.path.join(ROOT, "tests", "golden", "kron.npz")))


# Counters the time-parallel-gradient fuzz fills in (tests/test_gpu_fuzz.py::_tpg_check): how many draws pass the PLAIN
# north_star criterion (1e-10 of the array's largest entry) and how many needed the extended-precision floor term of
# DESIGN.md section 5 -- printed with the run's summary so the log says it where people look.
TPG_STATS = {"draws": 0, "needed_floor": 0, "needed_floor_seeds": [], "worst_plain": 0.0, "floor_details": []}


def pytest_terminal_summary(terminalreporter):
    if TPG_STATS["draws"]:
        terminalreporter.write_line(
            "time-parallel gradient criterion: %d draw evaluations, %d needed the 4 x extended-precision-floor term "
            "(the rest pass 1e-10 of the largest entry outright); worst plain distance %.2e; seeds needing the floor: %s"
            % (TPG_STATS["draws"], TPG_STATS["needed_floor"], TPG_STATS["worst_plain"],
               sorted(set(TPG_STATS["needed_floor_seeds"]))[:40]))
        for form, seed, w, wx, ox in TPG_STATS["floor_details"][:20]:
            terminalreporter.write_line(
                "    %s, seed %d: %.2e from the float64 oracle; %.2e from its extended-precision evaluation, from which the "
                "float64 oracle itself is %.2e (%s)" % (form, seed, w, wx, ox,
                "the device is the closer of the two to the exact result" if wx < ox else "the oracle is closer"))

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

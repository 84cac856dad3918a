# -*- coding: utf-8 -*-
"""Run the reference's OWN acceptance suites, unmodified and where they lie, over the shim of oracle/ref_shim.py
(build container only; /root/reference is read, nothing is copied):

    python tools/ref_acceptance.py              # celerite2.driver / .backprop = the CPU restatement (oracle/cpu.py)
    python tools/ref_acceptance.py --product    # = the product's pybind11 modules (needs an MI355X AND the reference)

  /root/reference/python/test/test_driver.py    dense Cholesky / triangular products vs the 8 driver functions
  /root/reference/python/test/test_backprop.py  *_fwd == plain, every *_rev vs finite differences

test_celerite2.py / test_terms.py import the original `celerite` package at module level and cannot run here.
The pass counts of the last run are recorded in oracle/README.md.
"""
import os
import sys

sys.dont_write_bytecode = True
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    import pytest

    from oracle import ref_shim

    if "--product" in sys.argv:
        from celerite2_amd import backprop, driver
        ref_shim.install(driver=driver, backprop=backprop)
    else:
        ref_shim.install()
    tests = [os.path.join(ref_shim.REF_TESTS, f) for f in ("test_driver.py", "test_backprop.py")]
    # no cache / no bytecode: nothing is written under /root/reference
    return pytest.main(tests + ["-q", "-p", "no:cacheprovider", "--rootdir", "/tmp", "-c", "/dev/null"])


if __name__ == "__main__":
    sys.exit(main())

# -*- coding: utf-8 -*-
"""celerite2_amd -- MI355X-native (gfx950 HIP) backend for celerite2's O(N)
semiseparable GP linear algebra.

    celerite2_amd.driver / celerite2_amd.backprop   pybind11 drop-ins for celerite2.driver / .backprop
    celerite2_amd.ops                               batched device-resident ops on torch tensors
    celerite2_amd.gp                                thin batched GaussianProcess frontend
    celerite2_amd.terms                             kernel terms -> (c, a, U, V)

Nothing is imported eagerly: the compiled pieces fail loudly on first use if the
HIP library has not been built (`python -m celerite2_amd.build`).
"""
import ctypes as _ctypes
import importlib.util as _ilu
import os as _os


def _preload_hip_runtime():
    """One HIP runtime per process.  PyTorch-ROCm wheels bundle their own libamdhip64.so (soname
    libamdhip64.so.7, the same soname libcelerite2_amd.so needs).  If our library were loaded first it would
    bind /opt/rocm's copy and a later `import torch` would bring a second runtime into the process (the
    second one then sees no device).  Loading torch's copy first -- located WITHOUT importing torch --
    makes every later lookup of that soname resolve to the same object."""
    try:
        spec = _ilu.find_spec("torch")
    except Exception:  # pragma: no cover
        spec = None
    if spec is None or not spec.submodule_search_locations:
        return None
    path = _os.path.join(list(spec.submodule_search_locations)[0], "lib", "libamdhip64.so")
    if _os.path.exists(path):
        try:
            return _ctypes.CDLL(path, mode=_ctypes.RTLD_GLOBAL)
        except OSError:  # pragma: no cover
            return None
    return None


_hip_runtime = _preload_hip_runtime()

__version__ = "0.1.0"
__all__ = ["__version__"]

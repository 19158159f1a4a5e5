"""MI355X-native filtered vector search for a milli-derived engine.

The package holds only what the hot path needs: ``csrc/`` (HIP kernels and the
C ABI of ``libvs_amd.so``), ``abi`` (ctypes binding of that ABI) and ``engine``
(host-side mirror of the reference's Index / Search interface for this path).
"""
from . import abi  # noqa: F401  (raises ImportError if libvs_amd.so has not been built)

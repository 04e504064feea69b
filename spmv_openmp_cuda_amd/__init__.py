"""MI355X-native SpMV (fp64 y = A.x on CSR / ELL) behind the C driver surface of
andreadiiorio/SpMV_openMP_CUDA.

The compute path is ``lib/libspmvhip.so`` (hand-written gfx950 HIP kernels behind
the C-ABI of ``include/spmvHip.h``); this package is the thin Python mirror of
that interface used by ``bench.py`` and ``tests/``.  There is NO CPU fallback:
importing :mod:`spmv_openmp_cuda_amd.api` fails loudly when the library has not
been built (``make lib``, or ``__graft_entry__.build()``).
"""
from . import ctypes_defs  # noqa: F401  (struct mirrors; safe without the .so)

__all__ = ["ctypes_defs"]

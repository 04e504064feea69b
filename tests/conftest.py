import ctypes as C
import os
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def _ensure_built():
    """Build what is missing (no-op when the .so files travelled with the snapshot)."""
    need = [os.path.join(ROOT, "spmv_openmp_cuda_amd", "lib", "libspmvhip.so"),
            os.path.join(ROOT, "spmv_openmp_cuda_amd", "lib", "libspmvhost.so"),
            os.path.join(ROOT, "spmv_openmp_cuda_amd", "bin", "SpMV_HIP.elf"),
            os.path.join(ROOT, "oracle", "liboracle.so"),
            os.path.join(ROOT, "tests", "harness", "test_SpMV_HIP.elf")]
    if not all(os.path.exists(p) for p in need):
        subprocess.check_call(["make", "-C", ROOT, "lib", "host", "oracle", "harness"], stdout=subprocess.DEVNULL)


_ensure_built()


class Oracle:
    """ctypes view of oracle/liboracle.so -- the CHECKER (tests only)."""

    def __init__(self):
        self.lib = C.CDLL(os.path.join(ROOT, "oracle", "liboracle.so"))
        vp, ul = C.c_void_p, C.c_ulong
        self.lib.oracleCsrSerial.argtypes = [ul, vp, vp, vp, vp, vp]
        self.lib.oracleCsrSerial32.argtypes = [ul, vp, vp, vp, vp, vp]
        self.lib.oracleCsrSerial64_32.argtypes = [ul, vp, vp, vp, vp, vp]
        self.lib.oracleCsrOmp32.argtypes = [ul, vp, vp, vp, vp, vp]
        self.lib.synthFillCsrRef.argtypes = [C.c_uint64, C.c_uint64, C.c_uint64, vp, vp, vp, C.c_uint64, C.c_uint64,
                                            C.c_uint64]
        self.lib.oracleVectorsDiffRef.argtypes = [vp, vp, ul, C.POINTER(C.c_double)]
        self.lib.oracleSizeofSpmat.restype = C.c_size_t
        self.lib.oracleSizeofConfig.restype = C.c_size_t

    @staticmethod
    def _p(a):
        return a.ctypes.data_as(C.c_void_p)

    def csr_serial(self, IRP, JA, AS, x):
        """sgemvSerial on 64-bit host arrays."""
        IRP = np.ascontiguousarray(IRP, dtype=np.uint64)
        JA = np.ascontiguousarray(JA, dtype=np.uint64)
        AS = np.ascontiguousarray(AS, dtype=np.float64)
        x = np.ascontiguousarray(x, dtype=np.float64)
        y = np.empty(IRP.size - 1, dtype=np.float64)
        assert self.lib.oracleCsrSerial(IRP.size - 1, self._p(IRP), self._p(JA), self._p(AS), self._p(x), self._p(y)) == 0
        return y

    def csr_serial_dev(self, IRP, JA32, AS, x):
        """same walk on device-format arrays (u32 columns, u32/u64 row pointers)."""
        y = np.empty(IRP.size - 1, dtype=np.float64)
        fn = self.lib.oracleCsrSerial32 if IRP.dtype == np.uint32 else self.lib.oracleCsrSerial64_32
        assert fn(IRP.size - 1, self._p(IRP), self._p(JA32), self._p(AS), self._p(x), self._p(y)) == 0
        return y

    def synth_fill(self, N, row_offset, IRP64, seed_s, seed_v, band):
        IRP64 = np.ascontiguousarray(IRP64, dtype=np.uint64)
        nnz = int(IRP64[-1] - IRP64[0])
        JA = np.empty(nnz, dtype=np.uint32)
        AS = np.empty(nnz, dtype=np.float64)
        assert self.lib.synthFillCsrRef(IRP64.size - 1, N, row_offset, self._p(IRP64), self._p(JA), self._p(AS),
                                        seed_s, seed_v, band) == 0
        return JA, AS


@pytest.fixture(scope="session")
def oracle():
    return Oracle()


def random_csr(rng, M, N, lens):
    """CSR with the given row lengths, sorted distinct columns, values U(-1,1)."""
    lens = np.asarray(lens, dtype=np.int64)
    IRP = np.zeros(M + 1, dtype=np.uint64)
    IRP[1:] = np.cumsum(lens)
    JA = np.empty(int(IRP[-1]), dtype=np.uint64)
    for r in range(M):
        if lens[r]:
            JA[int(IRP[r]):int(IRP[r + 1])] = np.sort(rng.choice(N, size=int(lens[r]), replace=False))
    AS = rng.uniform(-1, 1, size=JA.size)
    return IRP, JA, AS


def tight_error(IRP, JA, AS, x, y_ref, y):
    """max_i |dy_i| / sum_j |a_ij x_j|  (SURVEY 8d's non-gating tight check)."""
    prod = np.abs(AS * x[JA.astype(np.int64)])
    scale = np.add.reduceat(np.concatenate([prod, [0.0]]), np.minimum(IRP[:-1].astype(np.int64), prod.size))
    lens = np.diff(IRP.astype(np.int64))
    scale = np.where(lens > 0, scale, 0.0)
    d = np.abs(y_ref - y)
    ok_zero = np.all(d[scale == 0] == 0)
    nz = scale > 0
    return (np.max(d[nz] / scale[nz]) if nz.any() else 0.0) if ok_zero else np.inf

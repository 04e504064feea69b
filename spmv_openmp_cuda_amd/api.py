"""Python mirror of the C driver surface (include/spmvHip.h, include/SpMV.h).

Same names, argument meaning and error behaviour (0 = EXIT_SUCCESS, 1 =
EXIT_FAILURE, diagnostics on stderr) as the C entry points, which in turn mirror
the reference (src/include/SpMV.h:119-142, src/include/cudaUtils.h:60-78).
Everything numeric happens in libspmvhip.so; numpy is only used to marshal host
arrays.  The library is REQUIRED: there is no fallback implementation.
"""
import ctypes as C
import os

import numpy as np

from .ctypes_defs import CONFIG, POISON_NAN, SPMAT_TAG_ELL_TRANSPOSED, spmat, spmvDim3

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("SPMV_LIB") or os.path.join(_HERE, "lib", "libspmvhip.so")      # SPMV_LIB: tuning builds only
HOSTLIB_PATH = os.path.join(_HERE, "lib", "libspmvhost.so")


class SpmvHipError(RuntimeError):
    pass


def _load(path):
    if not os.path.exists(path):
        raise SpmvHipError(
            f"{path} is missing: build it with `make lib host` (or __graft_entry__.build()). "
            "This package has no CPU fallback.")
    return C.CDLL(path, mode=C.RTLD_GLOBAL)


lib = _load(LIB_PATH)
hostlib = _load(HOSTLIB_PATH)

_vp, _sz, _u64, _i = C.c_void_p, C.c_size_t, C.c_uint64, C.c_int
_SPMV_ARGS = [C.POINTER(spmat), _vp, CONFIG, _vp]
_sigs = {
    "spmvHipInit": ([_i, _sz, _sz], _i), "spmvHipFinalize": ([], _i), "spmvHipDeviceCount": ([], _i),
    "spmvHipSetStream": ([_vp], _i), "spmvHipSetSync": ([_i], _i), "spmvHipProbeLdsAtomicOrder": ([], _i),
    "spmvHipLastKernelSeconds": ([], C.c_double),
    "spmvHipLastLaunch": ([C.POINTER(spmvDim3), C.POINTER(spmvDim3)], _i),
    "spmvHipDeviceSynchronize": ([], _i),
    "spmvHipVecAlloc": ([C.POINTER(_vp), _sz], _i), "spmvHipVecFree": ([_vp], _i),
    "spmvHipVecUp": ([_vp, _vp, _sz], _i), "spmvHipVecDown": ([_vp, _vp, _sz], _i),
    "spmvHipVecFill": ([_vp, _sz, _u64], _i),
    "spmvHipMalloc": ([C.POINTER(_vp), _sz], _i), "spmvHipFree": ([_vp], _i),
    "spmvHipMemcpyUp": ([_vp, _vp, _sz], _i), "spmvHipMemcpyDown": ([_vp, _vp, _sz], _i),
    "spMatCpyCSR": ([C.POINTER(spmat), C.POINTER(spmat)], _i),
    "spMatCpyELL": ([C.POINTER(spmat), C.POINTER(spmat)], _i),
    "spMatCpyELLTransposed": ([C.POINTER(spmat), C.POINTER(spmat)], _i),
    "hipFreeSpmat": ([C.POINTER(spmat)], _i),
    "spmvHipCsrToEll": ([C.POINTER(spmat), _i, C.POINTER(spmat)], _i),
    "spmvHipAdoptCSR": ([C.POINTER(spmat), C.c_ulong, C.c_ulong, C.c_ulong, _vp, _i, _vp, _vp, _vp], _i),
    "hipSpMVRowsCSR": (_SPMV_ARGS, _i), "hipSpMVWarpPerRowCSR": (_SPMV_ARGS, _i),
    "hipSpMVRowsSELL": (_SPMV_ARGS, _i), "spmvHipBuildSell": ([C.POINTER(spmat)], _i),
    "spmvHipSellBytes": ([C.POINTER(spmat)], _sz),
    "hipSpMVTilesCSR": (_SPMV_ARGS, _i), "spmvHipBuildTiles": ([C.POINTER(spmat)], _i),
    "spmvHipTilesBytes": ([C.POINTER(spmat)], _sz),
    "hipSpMVStripesCSR": (_SPMV_ARGS, _i), "spmvHipBuildStripes": ([C.POINTER(spmat)], _i),
    "hipSpMVAutoCSR": (_SPMV_ARGS, _i), "spmvHipAutoChoice": ([C.POINTER(spmat), C.POINTER(C.c_double)], C.c_char_p),
    "spmvHipStripesBytes": ([C.POINTER(spmat)], _sz),
    "spmvHipStripesShape": ([C.POINTER(spmat), C.POINTER(C.c_uint), C.POINTER(C.c_uint), C.POINTER(_i), C.POINTER(C.c_double)], _i),
    "spmvHipEnqueueCSR": ([C.POINTER(spmat), _i, _vp, _vp, _vp], _i),
    "spmvHipEnqueueAuto": ([C.POINTER(spmat), _vp, _vp, _vp], _i),
    "spmvHipEnqueueAutoRows": ([C.POINTER(spmat), _vp, _vp, _vp], _i),
    "spmvHipAutoChoiceRows": ([C.POINTER(spmat), C.POINTER(C.c_double)], C.c_char_p),
    "spmvHipBuildStripesOpt": ([C.POINTER(spmat), _vp], _i), "spmvHipStripesInfo": ([C.POINTER(spmat), _vp], _i),
    "spmvHipShardCSR": ([C.POINTER(spmat), _i, C.POINTER(_vp)], _i),
    "spmvHipShardCSRGroups": ([C.POINTER(spmat), _i, _i, C.POINTER(_vp)], _i),
    "spmvHipSpMVSharded": ([_vp, _vp, _i, _vp, C.POINTER(C.c_double), C.POINTER(C.c_double)], _i),
    "spmvHipShardFree": ([_vp], _i),
    "hipSpMVRowsELL": (_SPMV_ARGS, _i), "hipSpMVRowsELLNNTransposed": (_SPMV_ARGS, _i),
    "hipSpMVWarpsPerRowELLNTrasposed": (_SPMV_ARGS, _i),
    "spmvHipSetVariant": ([C.c_char_p, _i], _i), "spmvHipSetEllRowLens": ([_i], _i),
    "spmvHipSetUnitValues": ([_i], _i), "spmvHipUnitValue": ([C.POINTER(spmat), C.POINTER(C.c_double)], _i),
    "spmvHipRowsCSR": ([C.POINTER(spmat), _vp, C.POINTER(CONFIG), _vp], _i),
    "spmvHipWarpPerRowCSR": ([C.POINTER(spmat), _vp, C.POINTER(CONFIG), _vp], _i),
    "spmvHipRowsELL": ([C.POINTER(spmat), _vp, C.POINTER(CONFIG), _vp], _i),
    "spmvHipWarpsPerRowELL": ([C.POINTER(spmat), _vp, C.POINTER(CONFIG), _vp], _i),
    "spmvHipDropCache": ([], _i),
    "spmvHipEventCreate": ([C.POINTER(_vp)], _i), "spmvHipEventDestroy": ([_vp], _i),
    "spmvHipEventRecord": ([_vp], _i), "spmvHipEventElapsedMs": ([_vp, _vp, C.POINTER(C.c_float)], _i),
    "spmvHipPartitionRows": ([_vp, C.c_ulong, _i, _vp], _i),
    "spmvHipRowBlockCSR": ([C.POINTER(spmat), C.c_ulong, C.c_ulong], C.POINTER(spmat)),
    "spmvHipSynthFillCSR": ([C.c_ulong, C.c_ulong, C.c_ulong, _vp, _i, _vp, _vp, _u64, _u64, C.c_ulong], _i),
    "spmvHipCompactRows": ([_vp, _vp, _vp, _i, C.c_ulong], _i),
    "spmvHipWindowCreate": ([_sz, C.POINTER(_vp), _vp], _i), "spmvHipWindowFree": ([_vp], _i),
    "spmvHipWindowOpen": ([_vp, _i, C.POINTER(_vp)], _i), "spmvHipWindowClose": ([_vp], _i),
    "spmvHipPeerPush": ([_vp, _sz, _sz, _i, _vp], _i), "spmvHipPeerPushJoin": ([], _i),
    "spmvHipTilesShape": ([C.POINTER(spmat), C.POINTER(C.c_uint), C.POINTER(C.c_uint)], _i),
    "hipSpMVTilesExpand": ([C.POINTER(spmat), _vp], _i),
    "hipSpMVTilesReduce": ([C.POINTER(spmat), C.c_uint, C.c_uint, _vp, _i, _vp], _i),
    "hipSpMVTilesReducePush": ([C.POINTER(spmat), _vp, _i, _vp], _i), "spmvHipTilesPushFailed": ([C.POINTER(spmat)], _i),
    "spmvHipTilesPushJoin": ([], _i),
    "spmvHipBuildTilesOpt": ([C.POINTER(spmat), _vp], _i), "spmvHipTilesInfo": ([C.POINTER(spmat), _vp], _i),
    "spmvHipTilesBinRow": ([C.POINTER(spmat), C.c_uint, C.POINTER(C.c_ulong)], _i),
}


class spmvTilesOpts(C.Structure):
    """include/spmvHip.h `spmvTilesOpts` (0 / 0 / -1 / 0 = automatic)."""
    _fields_ = [("rowsPerBin", C.c_uint), ("taper", _i), ("ntStore", _i), ("chunk", C.c_uint), ("deterministic", _i)]

    def __init__(self, rowsPerBin=0, taper=0, ntStore=-1, chunk=0, deterministic=0):
        super().__init__(rowsPerBin, taper, ntStore, chunk, deterministic)


class spmvTilesInfo(C.Structure):
    _fields_ = [("nBins", C.c_uint), ("rowsPerBin", C.c_uint), ("nSlices", C.c_uint), ("taper", _i), ("ntStore", _i),
                ("chunk", C.c_uint), ("buildMs", C.c_double), ("bytes", _sz), ("allocMs", C.c_double), ("tempBytes", _sz),
                ("deterministic", _i)]


class spmvStripesOpts(C.Structure):
    """include/spmvHip.h `spmvStripesOpts` (0 / 0 / -1 / -1 / 0 = automatic)."""
    _fields_ = [("rowsPerBin", C.c_uint), ("grid", C.c_uint), ("spread", _i), ("wide", _i), ("deterministic", _i)]

    def __init__(self, rowsPerBin=0, grid=0, spread=-1, wide=-1, deterministic=0):
        super().__init__(rowsPerBin, grid, spread, wide, deterministic)


class spmvStripesInfo(C.Structure):
    _fields_ = [("nBins", C.c_uint), ("rowsPerBin", C.c_uint), ("grid", C.c_uint), ("spread", C.c_uint), ("wide", _i),
                ("deterministic", _i), ("buildMs", C.c_double), ("bytes", _sz)]


IPC_HANDLE_BYTES = 64
MAX_PEERS = 15
for _name, (_args, _res) in _sigs.items():
    _f = getattr(lib, _name, None)
    if _f is None and os.environ.get("SPMV_LIB"):        # an older tuning build for an A/B may lack the newest entry points
        continue
    if _f is None:
        raise SpmvHipError(f"{LIB_PATH} does not export {_name}: rebuild it (`make lib`)")
    _f.argtypes, _f.restype = _args, _res

_hsigs = {
    "MMtoCSR": ([C.c_char_p], C.POINTER(spmat)), "MMtoELL": ([C.c_char_p], C.POINTER(spmat)),
    "freeSpmat": ([C.POINTER(spmat)], None),
    "ellTranspose": ([C.POINTER(spmat)], C.POINTER(spmat)),
    "csrToEll": ([C.POINTER(spmat)], C.POINTER(spmat)),
    "doubleVectorsDiff": ([_vp, _vp, C.c_ulong, C.POINTER(C.c_double)], _i),
    "statsAvgVar": ([_vp, C.c_uint, _vp], None),
    "fillRndVector": ([C.c_ulong, _vp], _i),
    "spmvModeFromString": ([C.c_char_p], _i),
    "writeDoubleVector": ([C.c_char_p, _vp, C.c_ulong], _i),
    "readDoubleVector": ([C.c_char_p, C.POINTER(C.c_ulong)], C.POINTER(C.c_double)),
    "spmvSynthPowerLawLengths": ([_u64, _u64, C.c_uint32, _u64, _vp, C.POINTER(C.c_double)], _i),
    "spmvSynthPrefix": ([_vp, _u64, _vp], _u64),
    "spmvSynthMakeX": ([_u64, _u64, _vp], None),
    "spmvSynthPerm": ([_u64, _u64, _u64], _u64),
    "spmvSynthWriteMtx": ([C.c_char_p, _i, C.c_ulong, C.c_ulong, C.c_ulong, _u64, C.POINTER(C.c_ulong), C.POINTER(C.c_ulong),
                           C.POINTER(C.c_ulong)], _i),
    "spmvSynthStructuredValue": ([_u64, C.c_ulong, C.c_ulong], C.c_double),
}
for _name, (_args, _res) in _hsigs.items():
    _f = getattr(hostlib, _name)
    _f.argtypes, _f.restype = _args, _res

SPMV_LAUNCHERS = {
    "hipSpMVRowsCSR": lib.hipSpMVRowsCSR,
    "hipSpMVWarpPerRowCSR": lib.hipSpMVWarpPerRowCSR,
    "hipSpMVTilesCSR": lib.hipSpMVTilesCSR,
    "hipSpMVStripesCSR": lib.hipSpMVStripesCSR,
    "hipSpMVAutoCSR": lib.hipSpMVAutoCSR,
    "hipSpMVRowsSELL": lib.hipSpMVRowsSELL,
    "hipSpMVRowsELL": lib.hipSpMVRowsELL,
    "hipSpMVRowsELLNNTransposed": lib.hipSpMVRowsELLNNTransposed,
    "hipSpMVWarpsPerRowELLNTrasposed": lib.hipSpMVWarpsPerRowELLNTrasposed,
}


def _check(rc, what):
    if rc != 0:
        raise SpmvHipError(f"{what} failed (EXIT_FAILURE) -- see stderr")


def _ptr(a):
    return a.ctypes.data_as(C.c_void_p) if a is not None else None


# ----------------------------------------------------------------- lifecycle
def spmvHipInit(dev=0):
    _check(lib.spmvHipInit(int(dev), C.sizeof(spmat), C.sizeof(CONFIG)), "spmvHipInit")


def spmvHipFinalize():
    lib.spmvHipFinalize()


# ----------------------------------------------------------------- host matrices
class HostCSR:
    """A host `spmat` in CSR form over numpy arrays (64-bit indices like the
    reference's loader output).  Keeps the arrays alive."""

    def __init__(self, M, N, IRP, JA, AS, with_row_lens=True):
        self.IRP = np.ascontiguousarray(IRP, dtype=np.uint64)
        self.JA = np.ascontiguousarray(JA, dtype=np.uint64)
        self.AS = np.ascontiguousarray(AS, dtype=np.float64)
        assert self.IRP.shape == (M + 1,) and self.JA.shape == self.AS.shape
        self.RL = np.diff(self.IRP).astype(np.uint64) if with_row_lens else None
        s = spmat()
        s.M, s.N, s.NZ = M, N, int(self.JA.size)
        s.IRP = self.IRP.ctypes.data_as(C.POINTER(C.c_ulong))
        s.JA = self.JA.ctypes.data_as(C.POINTER(C.c_ulong))
        s.AS = self.AS.ctypes.data_as(C.POINTER(C.c_double))
        if self.RL is not None:
            s.RL = self.RL.ctypes.data_as(C.POINTER(C.c_ulong))
        self.struct = s

    @property
    def M(self):
        return self.struct.M

    @property
    def N(self):
        return self.struct.N

    @property
    def NZ(self):
        return self.struct.NZ

    def to_ell(self, with_row_lens=True):
        """Row-major ELL with {JA=0, AS=0} padding (parser.c:246-253 semantics)."""
        M = self.M
        lens = np.diff(self.IRP).astype(np.int64)
        K = int(lens.max()) if M else 0
        JA = np.zeros((M, K), dtype=np.uint64)
        AS = np.zeros((M, K), dtype=np.float64)
        if self.NZ:
            rows = np.repeat(np.arange(M), lens)
            pos = np.arange(self.NZ) - np.repeat(self.IRP[:-1].astype(np.int64), lens)
            JA[rows, pos] = self.JA
            AS[rows, pos] = self.AS
        return HostELL(M, self.N, self.NZ, K, JA, AS, lens.astype(np.uint64) if with_row_lens else None)


class HostELL:
    """A host `spmat` in ELL form.  `transposed` follows the reference's
    ellTranspose field convention (sparseUtils.c:168-171)."""

    def __init__(self, M, N, NZ, K, JA, AS, RL=None, transposed=False):
        self.JA = np.ascontiguousarray(JA, dtype=np.uint64)
        self.AS = np.ascontiguousarray(AS, dtype=np.float64)
        self.RL = None if RL is None else np.ascontiguousarray(RL, dtype=np.uint64)
        self.rows, self.slots, self.transposed = M, K, transposed
        s = spmat()
        s.NZ = NZ
        if transposed:
            s.M, s.N, s.MAX_ROW_NZ = K, M, M
            s.dev = SPMAT_TAG_ELL_TRANSPOSED
            s.pitchJA = N                       # column count for the upload's range check (see ellTranspose)
        else:
            s.M, s.N, s.MAX_ROW_NZ = M, N, K
        s.JA = self.JA.ctypes.data_as(C.POINTER(C.c_ulong))
        s.AS = self.AS.ctypes.data_as(C.POINTER(C.c_double))
        if self.RL is not None:
            s.RL = self.RL.ctypes.data_as(C.POINTER(C.c_ulong))
        self.struct = s
        self._N = N

    def transpose(self):
        """ellTranspose (sparseUtils.c:145-185) on numpy arrays."""
        assert not self.transposed
        return HostELL(self.rows, self._N, self.struct.NZ, self.slots,
                       self.JA.reshape(self.rows, self.slots).T.copy(),
                       self.AS.reshape(self.rows, self.slots).T.copy(), self.RL, transposed=True)


# ----------------------------------------------------------------- device objects
class DeviceVector:
    def __init__(self, n):
        self.n = int(n)
        p = C.c_void_p()
        _check(lib.spmvHipVecAlloc(C.byref(p), self.n), "spmvHipVecAlloc")
        self.ptr = p

    def up(self, host):
        host = np.ascontiguousarray(host, dtype=np.float64)
        assert host.size == self.n
        _check(lib.spmvHipVecUp(self.ptr, _ptr(host), self.n), "spmvHipVecUp")
        return self

    def down(self):
        out = np.empty(self.n, dtype=np.float64)
        _check(lib.spmvHipVecDown(_ptr(out), self.ptr, self.n), "spmvHipVecDown")
        return out

    def poison(self):
        _check(lib.spmvHipVecFill(self.ptr, self.n, POISON_NAN), "spmvHipVecFill")

    def free(self):
        if self.ptr:
            lib.spmvHipVecFree(self.ptr)
            self.ptr = None

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


class DeviceBuffer:
    """Raw device bytes (for device-format matrices built on the GPU)."""

    def __init__(self, nbytes):
        self.nbytes = int(nbytes)
        p = C.c_void_p()
        _check(lib.spmvHipMalloc(C.byref(p), max(self.nbytes, 1)), "spmvHipMalloc")
        self.ptr = p

    def up(self, host):
        host = np.ascontiguousarray(host)
        assert host.nbytes == self.nbytes
        _check(lib.spmvHipMemcpyUp(self.ptr, _ptr(host), self.nbytes), "spmvHipMemcpyUp")
        return self

    def down(self, dtype):
        out = np.empty(self.nbytes // np.dtype(dtype).itemsize, dtype=dtype)
        _check(lib.spmvHipMemcpyDown(_ptr(out), self.ptr, self.nbytes), "spmvHipMemcpyDown")
        return out

    def free(self):
        if self.ptr:
            lib.spmvHipFree(self.ptr)
            self.ptr = None

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


class DeviceMatrix:
    """A device handle (`spmat` whose `dev` is set) plus what keeps it alive."""

    def __init__(self):
        self.handle = spmat()
        self.keep = []
        self.rows = 0

    def free(self):
        if self.handle.dev:
            lib.hipFreeSpmat(C.byref(self.handle))
        for k in self.keep:
            if hasattr(k, "free"):
                k.free()
        self.keep = []

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


def spMatCpyCSR(host: HostCSR) -> DeviceMatrix:
    d = DeviceMatrix()
    _check(lib.spMatCpyCSR(C.byref(host.struct), C.byref(d.handle)), "spMatCpyCSR")
    d.rows = host.M
    return d


def spMatCpyELL(host: HostELL) -> DeviceMatrix:
    d = DeviceMatrix()
    _check(lib.spMatCpyELL(C.byref(host.struct), C.byref(d.handle)), "spMatCpyELL")
    d.rows = host.rows
    return d


def csr_to_ell_device(dcsr: DeviceMatrix, transposed: bool) -> DeviceMatrix:
    d = DeviceMatrix()
    _check(lib.spmvHipCsrToEll(C.byref(dcsr.handle), 1 if transposed else 0, C.byref(d.handle)), "spmvHipCsrToEll")
    d.rows = dcsr.rows
    return d


def spmv(launcher: str, dmat: DeviceMatrix, dx: DeviceVector, dy: DeviceVector, cfg: CONFIG = None):
    """Run one of the five HIP launchers; raises on EXIT_FAILURE."""
    fn = SPMV_LAUNCHERS[launcher]
    _check(fn(C.byref(dmat.handle), dx.ptr, cfg if cfg is not None else CONFIG(), dy.ptr), launcher)


def build_tiles(dmat: DeviceMatrix, rowsPerBin=0, taper=False, ntStore=-1, chunk=0, deterministic=False):
    """spmvHipBuildTilesOpt: (re)build the two-phase format of this handle with explicit options."""
    o = spmvTilesOpts(int(rowsPerBin), 1 if taper else 0, int(ntStore), int(chunk), 1 if deterministic else 0)     # (no second form for this format)
    _check(lib.spmvHipBuildTilesOpt(C.byref(dmat.handle), C.byref(o)), "spmvHipBuildTilesOpt")


def tiles_info(dmat: DeviceMatrix) -> spmvTilesInfo:
    info = spmvTilesInfo()
    _check(lib.spmvHipTilesInfo(C.byref(dmat.handle), C.byref(info)), "spmvHipTilesInfo")
    return info


def build_stripes(dmat: DeviceMatrix, rowsPerBin=0, grid=0, spread=-1, wide=-1, deterministic=False):
    """spmvHipBuildStripesOpt: (re)build the stripes format of this handle with explicit options."""
    o = spmvStripesOpts(int(rowsPerBin), int(grid), int(spread), int(wide), int(deterministic))
    _check(lib.spmvHipBuildStripesOpt(C.byref(dmat.handle), C.byref(o)), "spmvHipBuildStripesOpt")


def stripes_info(dmat: DeviceMatrix) -> spmvStripesInfo:
    info = spmvStripesInfo()
    _check(lib.spmvHipStripesInfo(C.byref(dmat.handle), C.byref(info)), "spmvHipStripesInfo")
    return info


def set_variant(launcher: str, variant: int):
    _check(lib.spmvHipSetVariant(launcher.encode(), int(variant)), "spmvHipSetVariant")


def last_launch():
    g, b = spmvDim3(), spmvDim3()
    lib.spmvHipLastLaunch(C.byref(g), C.byref(b))
    return (g.x, g.y, g.z), (b.x, b.y, b.z)


def partition_rows(IRP, n_parts):
    IRP = np.ascontiguousarray(IRP, dtype=np.uint64)
    bounds = np.zeros(n_parts + 1, dtype=np.uint64)
    _check(lib.spmvHipPartitionRows(_ptr(IRP), IRP.size - 1, n_parts, _ptr(bounds)), "spmvHipPartitionRows")
    return bounds

"""ctypes mirrors of the C structs in include/spmv_types.h (spmat, CONFIG) and of
the reference's own layout (for the compiled reference under oracle/_ref only).

Reference definitions: src/include/sparseMatrix.h:25-42, src/include/config.h:21-32.
"""
import ctypes as C

c_ulong_p = C.POINTER(C.c_ulong)
c_double_p = C.POINTER(C.c_double)


class spmat(C.Structure):
    """include/spmv_types.h `spmat` (layout independent of compile flags)."""
    _fields_ = [
        ("NZ", C.c_ulong), ("M", C.c_ulong), ("N", C.c_ulong),
        ("JA", c_ulong_p),
        ("RL", c_ulong_p),
        ("IRP", c_ulong_p),
        ("MAX_ROW_NZ", C.c_ulong),
        ("AS", c_double_p),
        ("pitchJA", C.c_size_t), ("pitchAS", C.c_size_t),
        ("dev", C.c_void_p),
    ]


class spmvDim3(C.Structure):
    _fields_ = [("x", C.c_uint), ("y", C.c_uint), ("z", C.c_uint)]


class CONFIG(C.Structure):
    """include/spmv_types.h `CONFIG`."""
    _fields_ = [
        ("gridRows", C.c_ushort), ("gridCols", C.c_ushort),
        ("threadNum", C.c_uint),
        ("chunkDistrbFunc", C.c_void_p),
        ("gridSize", spmvDim3), ("blockSize", spmvDim3),
        ("sharedMemSize", C.c_size_t),
    ]


class ref_spmat(C.Structure):
    """The REFERENCE's spmat as compiled by gcc with -DROWLENS and without
    __CUDACC__ (sparseMatrix.h:25-42) -- only for calling oracle/_ref."""
    _fields_ = [
        ("NZ", C.c_ulong), ("M", C.c_ulong), ("N", C.c_ulong),
        ("JA", c_ulong_p),
        ("RL", c_ulong_p),
        ("IRP", c_ulong_p),
        ("MAX_ROW_NZ", C.c_ulong),
        ("AS", c_double_p),
    ]


class ref_CONFIG(C.Structure):
    """The REFERENCE's CONFIG without __CUDACC__ (config.h:21-32)."""
    _fields_ = [
        ("gridRows", C.c_ushort), ("gridCols", C.c_ushort),
        ("threadNum", C.c_uint),
        ("chunkDistrbFunc", C.c_void_p),
    ]


SPMAT_TAG_ELL_TRANSPOSED = 0x454C4C54
POISON_NAN = 0x7FF8DEADDEADDEAD
DOUBLE_DIFF_THREASH = 7e-4
MAXRND = 3e-5

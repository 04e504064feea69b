"""Synthetic benchmark matrices (DESIGN.md "Synthetic inputs").

The reference has no generator -- it only reads MatrixMarket files -- so the
recipe is defined here and printed with every result as
(N, nnz, avg / min / max row length, seeds).  Row lengths and x come from the
host library (csrc/host/synth_host.c), the per-entry columns/values are filled
on the device (csrc/hip/synth.hip) directly in device format, so the 1.6 G-nnz
matrix never exists on the host.  CPU twin for tests: oracle/synth_ref.c.
"""
import ctypes as C
from dataclasses import dataclass

import numpy as np

from . import api
from .api import DeviceBuffer, DeviceMatrix, _check, _ptr, hostlib, lib

SEED_STRUCT = 0x5EED0000
SEED_VAL = 0xA5A50000
SEED_VEC = 0xC0FFEE00
SEED_PERM = 0x9E3779B9


@dataclass
class Workload:
    name: str
    N: int              # rows = cols
    nnz: int
    law: str            # "uniform" | "powerlaw"
    max_row: int = 50000
    cfg: int = 0        # seed offset (BASELINE.json config index)
    band: int = 0       # 0 = columns stratified-uniform over [0,N); >0 = banded
    clip: int = 0       # >0: clip row lengths to this maximum (ELL comparison)


WORKLOADS = {
    # BASELINE.json configs[1..4]
    "c2": Workload("c2-uniform-1M-32", 1_000_000, 32_000_000, "uniform", cfg=2),
    "c3": Workload("c3-powerlaw-10M-200M", 10_000_000, 200_000_000, "powerlaw", cfg=3),
    "c4": Workload("c4-powerlaw-10M-clip64", 10_000_000, 200_000_000, "powerlaw", cfg=3, clip=64),
    "c5": Workload("c5-powerlaw-80M-1.6G", 80_000_000, 1_600_000_000, "powerlaw", cfg=5),
    # cache-friendly counterparts (columns within +-2^14 of the row)
    "c2b": Workload("c2b-uniform-1M-32-band", 1_000_000, 32_000_000, "uniform", cfg=2, band=1 << 14),
    "c3b": Workload("c3b-powerlaw-10M-200M-band", 10_000_000, 200_000_000, "powerlaw", cfg=3, band=1 << 14),
    # narrow band (x window of a row block fits L1/LDS): the kernel-bound regime
    "c3n": Workload("c3n-powerlaw-10M-200M-band512", 10_000_000, 200_000_000, "powerlaw", cfg=3, band=512),
    # small shapes for tests
    "tiny": Workload("tiny-powerlaw-20k", 20_000, 400_000, "powerlaw", max_row=5000, cfg=9),
    "tinyu": Workload("tiny-uniform-4k-32", 4096, 4096 * 32, "uniform", cfg=8),
}


def scaled(w: Workload, factor: float) -> Workload:
    """Same law at `factor` x the size (rows and nnz)."""
    return Workload(f"{w.name}-x{factor:g}", max(1, int(w.N * factor)), max(1, int(w.nnz * factor)), w.law,
                    w.max_row, w.cfg, w.band, w.clip)


def row_lengths(w: Workload) -> np.ndarray:
    """uint32 row lengths of the whole matrix (host)."""
    if w.law == "uniform":
        per = w.nnz // w.N
        lens = np.full(w.N, per, dtype=np.uint32)
    elif w.law == "powerlaw":
        lens = np.empty(w.N, dtype=np.uint32)
        s = C.c_double()
        _check(hostlib.spmvSynthPowerLawLengths(w.N, w.nnz, min(w.max_row, w.N), SEED_PERM + w.cfg, _ptr(lens),
                                                C.byref(s)), "spmvSynthPowerLawLengths")
    else:
        raise ValueError(w.law)
    if w.clip:
        np.minimum(lens, w.clip, out=lens)
    return lens


def prefix(lens: np.ndarray) -> np.ndarray:
    irp = np.empty(lens.size + 1, dtype=np.uint64)
    hostlib.spmvSynthPrefix(_ptr(lens), lens.size, _ptr(irp))
    return irp


def make_x(n: int, cfg: int) -> np.ndarray:
    x = np.empty(n, dtype=np.float64)
    hostlib.spmvSynthMakeX(n, SEED_VEC + cfg, _ptr(x))
    return x


def describe(w: Workload, lens: np.ndarray) -> dict:
    return {"workload": w.name, "N": int(w.N), "nnz": int(lens.sum(dtype=np.uint64)),
            "row_len_avg": float(lens.mean()), "row_len_min": int(lens.min()), "row_len_max": int(lens.max()),
            "columns": "banded+-%d" % w.band if w.band else "stratified-uniform",
            "seeds": [SEED_STRUCT + w.cfg, SEED_VAL + w.cfg, SEED_VEC + w.cfg]}


def device_csr(w: Workload, irp_global: np.ndarray, r0: int, r1: int) -> DeviceMatrix:
    """Rows [r0, r1) of workload `w` generated on the device in device format and
    adopted as a CSR handle.  `irp_global` is the whole matrix' row pointer."""
    rows = r1 - r0
    irp_local = (irp_global[r0:r1 + 1] - irp_global[r0]).astype(np.uint64)
    nnz = int(irp_local[-1])
    irp_bytes = 4 if nnz < (1 << 32) - 65536 else 8
    irp_dev_host = irp_local.astype(np.uint32 if irp_bytes == 4 else np.uint64)
    d_irp = DeviceBuffer(irp_dev_host.nbytes).up(irp_dev_host)
    d_ja = DeviceBuffer(4 * nnz)
    d_as = DeviceBuffer(8 * nnz)
    _check(lib.spmvHipSynthFillCSR(rows, w.N, r0, d_irp.ptr, irp_bytes, d_ja.ptr, d_as.ptr,
                                   SEED_STRUCT + w.cfg, SEED_VAL + w.cfg, w.band), "spmvHipSynthFillCSR")
    dm = DeviceMatrix()
    _check(lib.spmvHipAdoptCSR(C.byref(dm.handle), rows, w.N, nnz, d_irp.ptr, irp_bytes, d_ja.ptr, d_as.ptr,
                               _ptr(irp_dev_host)), "spmvHipAdoptCSR")
    dm.keep = [d_irp, d_ja, d_as]
    dm.rows = rows
    dm.nnz = nnz
    dm.irp_bytes = irp_bytes
    dm.buffers = {"IRP": d_irp, "JA": d_ja, "AS": d_as}
    return dm


def algorithmic_bytes_csr(nnz: int, M: int, N: int) -> int:
    """values+32-bit cols per nnz; 32-bit row ptr + fp64 store per row; x once."""
    return nnz * 12 + M * 12 + N * 8

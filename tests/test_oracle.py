"""Pin the oracle (oracle/spmv_oracle.c) BEFORE trusting it:
  (a) against the golden vectors produced by the reference's own CLI
      (tests/golden/make_golden.py -> y_*_csr.bin / y_*_ell.bin),
  (b) against oracle/_ref/libspmvref.so -- the reference's own C sources compiled
      where they lie -- on random matrices, whenever that file is present.
Bit-exact for the serial walk; the OpenMP variants (omp simd reduction may
re-associate) are held to 1e-15 relative to sum|a x|.
"""
import ctypes as C
import json
import os

import numpy as np
import pytest

from conftest import ROOT, random_csr, tight_error
from spmv_openmp_cuda_amd.ctypes_defs import CONFIG, ref_CONFIG, ref_spmat, spmat

GOLD = os.path.join(ROOT, "tests", "golden")
NAMES = ["cage4like", "sym6", "pattern8x5", "int5x7", "skew12x40", "rand300"]
REF_LIB = os.path.join(ROOT, "oracle", "_ref", "libspmvref.so")


def load_golden(name):
    with open(os.path.join(GOLD, name + ".parsed.json")) as f:
        p = json.load(f)
    g = {k: p[k] for k in ("M", "N", "NZ")}
    g["IRP"] = np.array(p["IRP"], dtype=np.uint64)
    g["JA"] = np.array(p["JA"], dtype=np.uint64)
    g["AS"] = np.array(p["AS"], dtype=np.float64)
    g["RL"] = np.array(p["RL"], dtype=np.uint64)
    g["K"] = p["ELL_MAX_ROW_NZ"]
    g["ELL_JA"] = np.array(p["ELL_JA"], dtype=np.uint64)
    g["ELL_AS"] = np.array(p["ELL_AS"], dtype=np.float64)
    g["x"] = np.fromfile(os.path.join(GOLD, f"x_{name}.bin"))
    g["y_csr"] = np.fromfile(os.path.join(GOLD, f"y_{name}_csr.bin"))
    g["y_ell"] = np.fromfile(os.path.join(GOLD, f"y_{name}_ell.bin"))
    return g


def _struct(cls, M, N, NZ, IRP=None, JA=None, AS=None, RL=None, K=0):
    s = cls()
    s.M, s.N, s.NZ, s.MAX_ROW_NZ = M, N, NZ, K
    for f, a, t in (("IRP", IRP, C.c_ulong), ("JA", JA, C.c_ulong), ("AS", AS, C.c_double), ("RL", RL, C.c_ulong)):
        if a is not None:
            setattr(s, f, a.ctypes.data_as(C.POINTER(t)))
    return s


def _cfg(cls, lib_fn_ptr, threads):
    c = cls()
    c.gridRows = c.gridCols = 8
    c.threadNum = threads
    c.chunkDistrbFunc = lib_fn_ptr
    return c


ALL_CSR = ("sgemvSerial", "spmvRowsBasicCSR", "spmvRowsBlocksCSR", "spmvTilesCSR", "spmvTilesAllocdCSR")
ALL_ELL = ("spmvRowsBasicELL", "spmvRowsBlocksELL", "spmvTilesELL")


@pytest.fixture(scope="module")
def olib(oracle):
    lib = oracle.lib
    for fn in ALL_CSR + ALL_ELL:
        getattr(lib, fn).argtypes = [C.POINTER(spmat), C.c_void_p, C.POINTER(CONFIG), C.c_void_p]
    return lib


def _fn_addr(lib, name):
    return C.cast(getattr(lib, name), C.c_void_p).value


def _run_oracle(olib, fn, m, x, grid=(8, 8)):
    y = np.full(m.M, np.nan)
    cfg = _cfg(CONFIG, _fn_addr(olib, "chunksNOOP"), 1)
    cfg.gridRows, cfg.gridCols = grid
    assert getattr(olib, fn)(C.byref(m), x.ctypes.data_as(C.c_void_p), C.byref(cfg), y.ctypes.data_as(C.c_void_p)) == 0
    return y


def test_layout_matches_headers(oracle):
    assert oracle.lib.oracleSizeofSpmat() == C.sizeof(spmat)
    assert oracle.lib.oracleSizeofConfig() == C.sizeof(CONFIG)


@pytest.mark.parametrize("name", NAMES)
def test_serial_oracle_vs_reference_cli_golden(olib, name):
    g = load_golden(name)
    m = _struct(spmat, g["M"], g["N"], g["NZ"], g["IRP"], g["JA"], g["AS"], g["RL"])
    y = _run_oracle(olib, "sgemvSerial", m, g["x"])
    # golden y was produced by the reference's spmvRowsBasicCSR (omp simd): identical up to re-association
    assert tight_error(g["IRP"], g["JA"], g["AS"], g["x"], g["y_csr"], y) <= 1e-15
    assert np.max(np.abs(y - g["y_csr"]), initial=0) <= 7e-4
    # numpy restatement of the definition, as a third opinion
    dense = np.zeros((g["M"], g["N"]))
    rows = np.repeat(np.arange(g["M"]), np.diff(g["IRP"].astype(np.int64)))
    dense[rows, g["JA"].astype(np.int64)] = g["AS"]
    assert np.allclose(dense @ g["x"], y, rtol=0, atol=1e-18 + 1e-13 * np.abs(dense).dot(np.abs(g["x"])).max())


@pytest.mark.parametrize("name", NAMES)
def test_omp_csr_and_ell_vs_golden(olib, name):
    g = load_golden(name)
    m = _struct(spmat, g["M"], g["N"], g["NZ"], g["IRP"], g["JA"], g["AS"], g["RL"])
    y = _run_oracle(olib, "spmvRowsBasicCSR", m, g["x"])
    assert np.array_equal(y, g["y_csr"])            # same loop nest, same compiler flags -> same bits
    for rl in (g["RL"], None):                      # with and without the row-lens early exit
        e = _struct(spmat, g["M"], g["N"], g["NZ"], None, g["ELL_JA"], g["ELL_AS"], rl, g["K"])
        y = _run_oracle(olib, "spmvRowsBasicELL", e, g["x"])
        assert tight_error(g["IRP"], g["JA"], g["AS"], g["x"], g["y_ell"], y) <= 1e-15
        if rl is not None:
            assert np.array_equal(y, g["y_ell"])


@pytest.mark.parametrize("name", NAMES)
def test_all_cpu_variants_vs_golden(olib, name):
    g = load_golden(name)
    m = _struct(spmat, g["M"], g["N"], g["NZ"], g["IRP"], g["JA"], g["AS"], g["RL"])
    e = _struct(spmat, g["M"], g["N"], g["NZ"], None, g["ELL_JA"], g["ELL_AS"], g["RL"], g["K"])
    for grid in ((8, 8), (2, 3)):
        for fn in ALL_CSR:
            y = _run_oracle(olib, fn, m, g["x"], grid)
            assert tight_error(g["IRP"], g["JA"], g["AS"], g["x"], g["y_csr"], y) <= 1e-15, (fn, grid)
        for fn in ALL_ELL:
            y = _run_oracle(olib, fn, e, g["x"], grid)
            assert tight_error(g["IRP"], g["JA"], g["AS"], g["x"], g["y_ell"], y) <= 1e-15, (fn, grid)


def test_reference_gate_semantics(oracle):
    """oracleVectorsDiffRef restates utils.c:362-393 including its NaN blind spot."""
    a = np.array([1.0, 2.0, 3.0])
    d = C.c_double()
    f = oracle.lib.oracleVectorsDiffRef
    p = lambda v: v.ctypes.data_as(C.c_void_p)
    assert f(p(a), p(a + 6e-4), 3, C.byref(d)) == 0
    assert f(p(a), p(a + 8e-4), 3, C.byref(d)) == 1
    nan = np.array([1.0, np.nan, 3.0])
    assert f(p(a), p(nan), 3, C.byref(d)) == 0      # the reference lets NaN through


@pytest.mark.skipif(not os.path.exists(REF_LIB), reason="oracle/_ref not built (no /root/reference here)")
def test_restatement_equals_compiled_reference(olib):
    ref = C.CDLL(REF_LIB)
    assert ref.refSizeofSpmat() == C.sizeof(ref_spmat) and ref.refSizeofConfig() == C.sizeof(ref_CONFIG)
    ref.refChunksNOOP.restype = C.c_void_p
    for fn in ALL_CSR + ALL_ELL:
        getattr(ref, fn).argtypes = [C.POINTER(ref_spmat), C.c_void_p, C.POINTER(ref_CONFIG), C.c_void_p]
    rng = np.random.default_rng(99)
    for trial in range(6):
        M, N = int(rng.integers(1, 400)), int(rng.integers(1, 500))
        lens = rng.integers(0, min(N, 60) + 1, size=M)
        if trial == 0:
            lens[:] = 0
        IRP, JA, AS = random_csr(rng, M, N, lens)
        RL = np.diff(IRP).astype(np.uint64)
        x = np.sin(rng.uniform(0, 7, N)) * 3e-5
        mr = _struct(ref_spmat, M, N, JA.size, IRP, JA, AS, RL)
        mo = _struct(spmat, M, N, JA.size, IRP, JA, AS, RL)
        cr = _cfg(ref_CONFIG, ref.refChunksNOOP(), ref.refMaxThreads())
        grids = [(8, 8), (3, 5), (1, 1), (40, 2)] if trial < 3 else [(8, 8)]
        for grid in grids:
            cr.gridRows, cr.gridCols = grid
            # the reference's spmvTilesAllocdCSR double-frees when a column partition holds no entry
            # (realloc(p, 0) == NULL is treated as an error, sparseUtils.c:118-127): only compare where it survives
            gc = grid[1]
            cb, rem = N // gc, N % gc
            starts = [g * cb + min(g, rem) for g in range(gc + 1)]
            per_part = np.histogram(JA.astype(np.int64), bins=starts)[0] if JA.size else np.zeros(gc)
            for fn in ALL_CSR:
                yo = _run_oracle(olib, fn, mo, x, grid)
                if fn == "spmvTilesAllocdCSR" and (per_part == 0).any():
                    assert np.array_equal(yo, _run_oracle(olib, "spmvTilesCSR", mo, x, grid))   # same tiles, same sums
                    continue
                yr = np.full(M, np.nan)
                assert getattr(ref, fn)(C.byref(mr), x.ctypes.data_as(C.c_void_p), C.byref(cr), yr.ctypes.data_as(C.c_void_p)) == 0
                assert np.array_equal(yr, yo), (fn, grid)
        # ELL through each side's own layout
        K = int(lens.max()) if M else 0
        EJ = np.zeros((M, max(K, 1)), dtype=np.uint64)[:, :K].copy()
        EA = np.zeros((M, max(K, 1)))[:, :K].copy()
        for r in range(M):
            b, l = int(IRP[r]), int(lens[r])
            EJ[r, :l], EA[r, :l] = JA[b:b + l], AS[b:b + l]
        er = _struct(ref_spmat, M, N, JA.size, None, EJ, EA, RL, K)
        eo = _struct(spmat, M, N, JA.size, None, EJ, EA, RL, K)
        for grid in grids:
            cr.gridRows, cr.gridCols = grid
            for fn in ALL_ELL:
                yr = np.full(M, np.nan)
                assert getattr(ref, fn)(C.byref(er), x.ctypes.data_as(C.c_void_p), C.byref(cr), yr.ctypes.data_as(C.c_void_p)) == 0
                assert np.array_equal(yr, _run_oracle(olib, fn, eo, x, grid)), (fn, grid)


def test_synth_twin_properties(oracle):
    """CPU twin of the device generator: sorted distinct in-range columns, values in [-1,1)."""
    from spmv_openmp_cuda_amd import synth
    for band in (0, 50):
        w = synth.Workload("t", 5000, 100000, "powerlaw", 2000, 9, band)
        lens = synth.row_lengths(w)
        assert lens.sum() == 100000 and lens.max() == 2000 and lens.min() >= 1
        irp = synth.prefix(lens)
        ja, as_ = oracle.synth_fill(w.N, 0, irp, 1, 2, band)
        for r in (0, 1, 17, 4999, int(np.argmax(lens))):
            seg = ja[int(irp[r]):int(irp[r + 1])].astype(np.int64)
            assert (np.diff(seg) > 0).all() and seg.min() >= 0 and seg.max() < w.N
            if band and lens[r] <= band:
                assert np.abs(seg - r).max() <= 2 * max(band, int(lens[r])) + 1      # window is shifted at the edges
        assert as_.min() >= -1 and as_.max() < 1 and abs(as_.mean()) < 0.02
        # shard consistency: rows [a,b) generated with an offset equal the same rows of the whole
        a, b = 1234, 2345
        ja2, as2 = oracle.synth_fill(w.N, a, irp[a:b + 1], 1, 2, band)
        assert np.array_equal(ja2, ja[int(irp[a]):int(irp[b])]) and np.array_equal(as2, as_[int(irp[a]):int(irp[b])])

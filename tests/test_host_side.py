"""Host-side C (csrc/host): MatrixMarket loader, reshaping, parity gate, stats,
vector I/O, mode strings -- against the reference's own loader output
(tests/golden/*.parsed.json) and its documented semantics."""
import ctypes as C
import json
import os

import numpy as np
import pytest

from conftest import ROOT
from test_oracle import GOLD, NAMES, load_golden

from spmv_openmp_cuda_amd import api
from spmv_openmp_cuda_amd.ctypes_defs import SPMAT_TAG_ELL_TRANSPOSED

H = api.hostlib


def _arr(ptr, n, dtype):
    if n == 0:
        return np.zeros(0, dtype=dtype)
    return np.ctypeslib.as_array(ptr, shape=(n,)).astype(dtype)


@pytest.mark.parametrize("name", NAMES)
def test_loader_matches_reference_loader(name):
    g = load_golden(name)
    path = os.path.join(GOLD, name + ".mtx").encode()
    csr = H.MMtoCSR(path)
    assert csr, "MMtoCSR failed"
    m = csr.contents
    assert (m.M, m.N, m.NZ) == (g["M"], g["N"], g["NZ"])
    assert np.array_equal(_arr(m.IRP, m.M + 1, np.uint64), g["IRP"])
    assert np.array_equal(_arr(m.JA, m.NZ, np.uint64), g["JA"])
    assert np.array_equal(_arr(m.AS, m.NZ, np.float64), g["AS"])
    assert np.array_equal(_arr(m.RL, m.M, np.uint64), g["RL"])
    ell = H.MMtoELL(path)
    e = ell.contents
    assert e.MAX_ROW_NZ == g["K"] and not e.IRP
    assert np.array_equal(_arr(e.JA, e.M * e.MAX_ROW_NZ, np.uint64), g["ELL_JA"])
    assert np.array_equal(_arr(e.AS, e.M * e.MAX_ROW_NZ, np.float64), g["ELL_AS"])
    # csrToEll builds the same padded matrix from the CSR form
    e2 = H.csrToEll(csr).contents
    assert e2.MAX_ROW_NZ == g["K"]
    assert np.array_equal(_arr(e2.JA, e2.M * e2.MAX_ROW_NZ, np.uint64), g["ELL_JA"])
    # ellTranspose: field convention of sparseUtils.c:168-171 + column-major data
    t = H.ellTranspose(ell).contents
    assert (t.M, t.N, t.MAX_ROW_NZ) == (g["K"], g["M"], g["M"]) and t.dev == SPMAT_TAG_ELL_TRANSPOSED
    assert np.array_equal(_arr(t.JA, t.M * t.MAX_ROW_NZ, np.uint64).reshape(g["K"], g["M"]),
                          g["ELL_JA"].reshape(g["M"], g["K"]).T)
    assert np.array_equal(_arr(t.RL, g["M"], np.uint64), g["RL"])
    for p in (csr, ell):
        H.freeSpmat(p)


def _write(tmp_path, text, name="m.mtx"):
    p = tmp_path / name
    p.write_text(text)
    return str(p).encode()


def test_loader_rejects_what_it_must(tmp_path, capfd):
    hdr = "%%MatrixMarket matrix coordinate real general\n"
    bad = {
        "unsorted row": hdr + "2 3 2\n1 3 1.0\n1 2 2.0\n",                 # columns must ascend within a row
        "duplicate": hdr + "2 2 2\n1 1 1.0\n1 1 2.0\n",
        "out of range": hdr + "2 2 1\n3 1 1.0\n",
        "zero index": hdr + "2 2 1\n0 1 1.0\n",
        "too few entries": hdr + "2 2 3\n1 1 1.0\n",
        "too many entries": hdr + "2 2 1\n1 1 1.0\n2 2 1.0\n",
        "complex": "%%MatrixMarket matrix coordinate complex general\n1 1 1\n1 1 1.0 0.0\n",
        "array": "%%MatrixMarket matrix array real general\n1 1\n1.0\n",
        "no banner": "1 1 1\n1 1 1.0\n",
        "garbage value": hdr + "1 1 1\n1 1 abc\n",
    }
    for why, text in bad.items():
        path = _write(tmp_path, text)
        assert not H.MMtoCSR(path), why
        assert not H.MMtoELL(path), why
    assert not H.MMtoCSR(b"/nonexistent/file.mtx")
    capfd.readouterr()


def test_loader_details(tmp_path):
    # comments + blank line before the size line, rows interleaved (column-major file), symmetric diag handling
    text = ("%%MatrixMarket matrix coordinate real symmetric\n% c1\n%c2\n\n3 3 4\n"
            "1 1 1.5\n2 1 2.5\n3 1 -1\n3 3 4\n")
    m = H.MMtoCSR(_write(tmp_path, text)).contents
    assert (m.M, m.N, m.NZ) == (3, 3, 6)                       # 2*4 - 2 diagonal entries
    assert list(_arr(m.IRP, 4, np.uint64)) == [0, 3, 4, 6]
    assert list(_arr(m.JA, 6, np.uint64)) == [0, 1, 2, 0, 0, 2]
    assert list(_arr(m.AS, 6, np.float64)) == [1.5, 2.5, -1, 2.5, -1, 4]
    # empty matrix (no entries) is valid
    z = H.MMtoCSR(_write(tmp_path, "%%MatrixMarket matrix coordinate real general\n4 5 0\n", "z.mtx")).contents
    assert (z.M, z.N, z.NZ) == (4, 5, 0) and list(_arr(z.IRP, 5, np.uint64)) == [0] * 5


def test_ell_size_guard():
    """parser.c:223-232: 2*M*maxRow > 6<<27 entries is refused (BASELINE config 4 on the unclipped matrix)."""
    M = 1 << 20
    lens = np.ones(M, dtype=np.uint64)
    lens[0] = 1000                                             # 2*M*1000 > 805306368
    irp = np.zeros(M + 1, dtype=np.uint64)
    irp[1:] = np.cumsum(lens)
    nz = int(irp[-1])
    host = api.HostCSR(M, M, irp, np.zeros(nz, dtype=np.uint64), np.zeros(nz))
    assert not H.csrToEll(C.byref(host.struct))


def test_parity_gate_and_stats():
    a = np.array([0.0, 1.0, -2.0, 3.0])
    d = C.c_double()
    f = lambda b: H.doubleVectorsDiff(a.ctypes.data_as(C.c_void_p), b.ctypes.data_as(C.c_void_p), 4, C.byref(d))
    assert f(a + 6.9e-4) == 0 and abs(d.value + 6.9e-4) < 1e-12
    assert f(a + np.array([0, 0, 7.1e-4, 0])) == 1
    assert f(np.array([0.0, np.nan, -2.0, 3.0])) == 1          # NaN must FAIL (the reference passes it)
    assert f(np.array([0.0, np.inf, -2.0, 3.0])) == 1
    v = np.array([1.0, 2.0, 3.0, 6.0])
    out = np.zeros(2)
    H.statsAvgVar(v.ctypes.data_as(C.c_void_p), 4, out.ctypes.data_as(C.c_void_p))
    assert out[0] == 3.0 and abs(out[1] - v.var()) < 1e-15


def test_vector_io_and_rnd(tmp_path):
    v = np.sin(np.arange(1234.0))
    p = str(tmp_path / "v.bin").encode()
    assert H.writeDoubleVector(p, v.ctypes.data_as(C.c_void_p), v.size) == 0
    n = C.c_ulong(0)
    r = H.readDoubleVector(p, C.byref(n))
    assert n.value == v.size and np.array_equal(np.ctypeslib.as_array(r, shape=(v.size,)), v)
    x = np.empty(5000)
    assert H.fillRndVector(x.size, x.ctypes.data_as(C.c_void_p)) == 0
    assert np.isfinite(x).all() and np.abs(x).max() <= 3e-5 and x.std() > 1e-6


def test_mode_strings_are_exact_match():
    m = lambda s: H.spmvModeFromString(s.encode())
    assert m("CUDA_CSR_ROWS") == 8 and m("CUDA_CSR_ROWS_WARP") == 9 and m("HIP_CSR_ROWS_WARP") == 9
    assert m("CUDA_ELL_ROWS") == 10 and m("CUDA_ELL_ROWS_WARP_NN_TRANSPOSED") == 12
    assert m("CSR_ROWS") == 1 and m("ELL_TILES") == 7
    # appended modes keep the reference's values for everything before them (enum order of src/include/SpMV.h:42-59)
    assert m("CUDA_CSR_TILES") == m("HIP_CSR_TILES") == 14 and m("CUDA_SELL_ROWS") == 15
    assert m("CUDA_CSR_STRIPES") == m("HIP_CSR_STRIPES") == 16 and m("CUDA_CSR_AUTO") == m("HIP_CSR_AUTO") == 17
    assert m("CUDA_CSR_ROWS_WARPX") == -1 and m("CUDA_CSR") == -1 and m("") == -1   # no prefix matching


def test_partition_rows_balances_nnz():
    from spmv_openmp_cuda_amd import sharding, synth
    w = synth.WORKLOADS["tiny"]
    irp = synth.prefix(synth.row_lengths(w))
    for parts in (1, 2, 3, 8):
        b = api.partition_rows(irp, parts).astype(np.int64)
        assert np.array_equal(b, sharding.partition_by_nnz(irp, parts))
        assert b[0] == 0 and b[-1] == w.N and (np.diff(b) >= 0).all()
        share = np.diff(irp[b].astype(np.int64))
        assert share.sum() == w.nnz and np.abs(share - w.nnz / parts).max() <= 5000 + 1
    # equal-row snap: scattered heavy rows -> equal blocks are nnz-balanced within 1 % and are preferred
    w2 = synth.Workload("p", 64_000, 1_280_000, "powerlaw", 200, 9)
    irp2 = synth.prefix(synth.row_lengths(w2))
    plan = sharding.make_plan(irp2, 8)
    assert plan.equal_blocks and (np.diff(plan.bounds) == 8000).all()
    share = np.diff(irp2[plan.bounds].astype(np.int64))
    assert share.max() <= 1.01 * share.mean()
    assert not sharding.make_plan(irp2, 8, snap_tol=0).equal_blocks            # pure nnz split keeps its ragged blocks
    skew = np.concatenate([np.full(100, 1000), np.ones(63_900)]).astype(np.uint32)   # heavy rows clustered: no snap
    assert not sharding.make_plan(synth.prefix(skew), 8).equal_blocks
    # degenerate: more parts than rows, empty matrix
    irp2 = np.array([0, 5, 5, 9], dtype=np.uint64)
    b = api.partition_rows(irp2, 8)
    assert b[0] == 0 and b[-1] == 3 and (np.diff(b.astype(np.int64)) >= 0).all()
    b = api.partition_rows(np.zeros(1, dtype=np.uint64), 4)
    assert (b == 0).all()


def test_powerlaw_law_and_permutation():
    from spmv_openmp_cuda_amd import synth
    N = 100_000
    seen = np.array([H.spmvSynthPerm(N, 5, v) for v in range(0, N, 97)])
    assert seen.max() < N and np.unique(seen).size == seen.size
    full = np.array([H.spmvSynthPerm(1000, 3, v) for v in range(1000)])
    assert np.array_equal(np.sort(full), np.arange(1000))      # bijection
    w = synth.Workload("t", N, 2_000_000, "powerlaw", 50000, 3)
    lens = synth.row_lengths(w)
    assert lens.sum() == 2_000_000 and lens.max() == 50000 and lens.min() >= 1
    srt = np.sort(lens)[::-1].astype(np.float64)
    # heavy tail: top 1% of the rows hold far more than 1% of the nnz
    assert srt[: N // 100].sum() > 0.15 * 2_000_000
    # heavy rows are scattered, not clustered at the start
    heavy = np.nonzero(lens > 1000)[0]
    assert heavy.size > 5 and heavy.max() - heavy.min() > N // 2


def test_compressed_inputs(tmp_path):
    """.gz / .bz2 / .xz / .zip MatrixMarket files are inflated in-process (the reference shells out to gzip, xz, bzip2 and
    unzip, utils.c:433-462): zlib, libbz2 and liblzma (both loaded at run time), and a zip reader over zlib's raw inflate
    (first member; stored and deflated)."""
    import bz2
    import gzip
    import io
    import lzma
    import zipfile
    H.extractInTmpFS.argtypes = [C.c_char_p, C.c_char_p]
    raw = open(os.path.join(GOLD, "cage4like.mtx"), "rb").read()
    big = open(os.path.join(GOLD, "rand300.mtx"), "rb").read()        # larger than the first guess of the xz output buffer / 8
    g = load_golden("cage4like")

    def zipped(data, method):
        b = io.BytesIO()
        with zipfile.ZipFile(b, "w", method) as z:
            z.writestr("inner.mtx", data)
        return b.getvalue()
    cases = [("m.mtx.gz", gzip.compress(raw), raw), ("m.mtx.bz2", bz2.compress(raw), raw), ("m.mtx.xz", lzma.compress(raw), raw),
             ("b.mtx.xz", lzma.compress(big * 40), big * 40), ("m.mtx.zip", zipped(raw, zipfile.ZIP_DEFLATED), raw),
             ("s.mtx.zip", zipped(raw, zipfile.ZIP_STORED), raw), ("b.mtx.zip", zipped(big, zipfile.ZIP_DEFLATED), big)]
    for name, data, want in cases:
        src = tmp_path / name
        src.write_bytes(data)
        dst = str(tmp_path / "extracted").encode()
        assert H.extractInTmpFS(str(src).encode(), dst) == 0, name
        assert open(dst, "rb").read() == want, name
        if want is raw:
            m = H.MMtoCSR(dst).contents
            assert m.NZ == g["NZ"] and np.array_equal(_arr(m.JA, m.NZ, np.uint64), g["JA"])
    for name, data in (("bad.mtx.xz", b"not an xz stream"), ("bad.mtx.zip", b"PK not a zip"), ("trunc.mtx.zip", zipped(raw, zipfile.ZIP_DEFLATED)[:60])):
        src = tmp_path / name
        src.write_bytes(data)
        assert H.extractInTmpFS(str(src).encode(), str(tmp_path / "x").encode()) == 1, name
    plain = tmp_path / "p.mtx"
    plain.write_bytes(raw)
    assert H.extractInTmpFS(str(plain).encode(), str(tmp_path / "x").encode()) == -1      # not compressed
    assert H.extractInTmpFS(b"/nonexistent/whatever.mtx.xz", str(tmp_path / "x").encode()) == 1
    assert H.extractInTmpFS(b"/nonexistent/file.mtx.gz", str(tmp_path / "x").encode()) == 1

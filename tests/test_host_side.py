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


def test_parallel_parser_reads_what_fscanf_reads(tmp_path):
    """The in-memory parser (csrc/host/mmfast.c; every regular file goes through it): values in every notation -- short
    decimals and exponents (exact fast path), 16-17 significant digits, subnormals, huge exponents (strtod) -- must be
    the correctly rounded doubles fscanf's %lf gives, i.e. Python's float() of the same token; several chunks (the file
    is cut at line ends and parsed by all cores) must come back in file order."""
    rng = np.random.default_rng(11)
    M, N, per = 30_000, 900, 6
    cols = np.sort(np.stack([rng.choice(N, per, replace=False) for _ in range(M)]), axis=1)
    tokens = []
    specials = ["1e22", "1e23", "123456789012345", "1234567890123456", "0.1e-310", "4.9e-324", "1.7976931348623157e308", "-0.0",
                "+5", "007.250", ".5", "5.", "1E+2", "2.2250738585072014e-308", "9007199254740993", "0.30000000000000004"]
    for i in range(M * per):
        k = i % 7
        if i < len(specials):
            tokens.append(specials[i])
        elif k == 0:
            tokens.append(repr(float(rng.standard_normal() * 10.0 ** rng.integers(-30, 30))))     # 17 significant digits
        elif k == 1:
            tokens.append(f"{rng.integers(-99999, 99999) / 1000:.3f}")
        elif k == 2:
            tokens.append(f"{rng.integers(1, 10**9)}e{rng.integers(-25, 25)}")
        elif k == 3:
            tokens.append(str(int(rng.integers(-10**6, 10**6))))
        elif k == 4:
            tokens.append(f"{rng.random():.15g}")
        elif k == 5:
            tokens.append(f"{rng.random() * 1e-5:.12e}")
        else:
            tokens.append(f"{rng.integers(1, 2047) / 1024}")
    lines = ["%%MatrixMarket matrix coordinate real general", f"{M} {N} {M * per}"]
    i = 0
    for r in range(M):
        for c in cols[r]:
            lines.append(f"{r + 1} {c + 1} {tokens[i]}" if i % 5 else f"  {r + 1}\t{c + 1}   {tokens[i]}  ")     # odd spacing too
            i += 1
    path = _write(tmp_path, "\n".join(lines) + "\n")
    assert os.path.getsize(path) > 4 * 65536                    # several chunks
    m = H.MMtoCSR(path).contents
    assert (m.M, m.N, m.NZ) == (M, N, M * per)
    assert np.array_equal(_arr(m.JA, m.NZ, np.uint64), cols.reshape(-1).astype(np.uint64))
    got = _arr(m.AS, m.NZ, np.float64)
    want = np.array([float(t) for t in tokens])
    assert np.array_equal(got.view(np.uint64), want.view(np.uint64))            # bit for bit, -0.0 and subnormals included


def test_parallel_parser_symmetric_pattern_and_fallback(tmp_path):
    """Symmetric + pattern over several chunks (mirrored entries right behind their originals, so every row still sees
    ascending columns); a file whose entries straddle line ends -- legal for fscanf, not for a line-cut parallel parse --
    is still read (serial fallback); errors found by the parallel parser are reported as errors."""
    n = 20_000
    ent = []                                                   # lower triangle, column-major: what symmetric files look like
    rng = np.random.default_rng(5)
    for c in range(n):
        ent.append((c, c))
        for r in sorted(set(int(v) for v in rng.integers(c + 1, min(n, c + 40), size=3)) if c + 1 < n else []):
            ent.append((r, c))
    text = "%%MatrixMarket matrix coordinate pattern symmetric\n" + f"{n} {n} {len(ent)}\n" + "".join(f"{r + 1} {c + 1}\n" for r, c in ent)
    path = _write(tmp_path, text, "sym.mtx")
    assert os.path.getsize(path) > 4 * 65536
    m = H.MMtoCSR(path).contents
    import scipy.sparse as sp
    rows = np.array([e[0] for e in ent] + [e[1] for e in ent if e[0] != e[1]])
    colsv = np.array([e[1] for e in ent] + [e[0] for e in ent if e[0] != e[1]])
    ref = sp.csr_matrix((np.ones(rows.size), (rows, colsv)), shape=(n, n))
    ref.sort_indices()
    assert m.NZ == ref.nnz and np.array_equal(_arr(m.IRP, n + 1, np.uint64), ref.indptr.astype(np.uint64))
    assert np.array_equal(_arr(m.JA, m.NZ, np.uint64), ref.indices.astype(np.uint64)) and np.all(_arr(m.AS, m.NZ, np.float64) == 1.0)
    # entries straddling lines, > 64 KiB so that a chunk boundary falls inside the odd layout
    body = "".join(f"{r + 1}\n{r + 1} {0.5 + r}\n" for r in range(12_000))
    odd = H.MMtoCSR(_write(tmp_path, "%%MatrixMarket matrix coordinate real general\n12000 12000 12000\n" + body, "odd.mtx"))
    assert odd, "the serial fallback must read a file whose entries span lines"
    o = odd.contents
    assert o.NZ == 12000 and np.array_equal(_arr(o.AS, 12000, np.float64), 0.5 + np.arange(12000))
    # errors deep inside a large file
    big = ["%%MatrixMarket matrix coordinate real general", "50000 50000 50000"] + [f"{r + 1} {r + 1} 1.5" for r in range(50_000)]
    for lineno, bad in ((40_000, "60000 1 1.0"), (30_000, "0 5 1.0"), (45_000, "7 7 x1")):
        t = list(big)
        t[lineno] = bad
        assert not H.MMtoCSR(_write(tmp_path, "\n".join(t) + "\n", "bad.mtx")), bad
    assert not H.MMtoCSR(_write(tmp_path, "\n".join(big[:-1]) + "\n", "short.mtx"))
    assert not H.MMtoCSR(_write(tmp_path, "\n".join(big + ["3 4 1.0"]) + "\n", "long.mtx"))


def test_structured_generators_have_the_published_shapes(tmp_path):
    """spmvSynthWriteMtx: the three stand-ins for the matrices of the reference's report at reduced size,
    written as .mtx and read back by the loader: rows / longest row / locality as stated, columns ascending and distinct,
    values reproducible from (seed, row, column)."""
    def gen(kind, p0, p1, p2, name):
        Mv, NZv, mx = C.c_ulong(), C.c_ulong(), C.c_ulong()
        path = str(tmp_path / name).encode()
        assert H.spmvSynthWriteMtx(path, kind, p0, p1, p2, 77, C.byref(Mv), C.byref(NZv), C.byref(mx)) == 0
        m = H.MMtoCSR(path)
        assert m
        c = m.contents
        assert (c.M, c.N, c.NZ) == (Mv.value, Mv.value, NZv.value)
        irp, ja, as_ = _arr(c.IRP, c.M + 1, np.int64), _arr(c.JA, c.NZ, np.int64), _arr(c.AS, c.NZ, np.float64)
        lens = np.diff(irp)
        assert lens.max() == mx.value
        rows = np.repeat(np.arange(c.M), lens)
        inner = np.ones(c.NZ, dtype=bool)
        inner[irp[1:-1][irp[1:-1] < c.NZ]] = False
        assert np.all(np.diff(ja)[inner[1:]] > 0)               # ascending, distinct inside every row
        for i in (0, c.NZ // 3, c.NZ - 1):
            assert as_[i] == H.spmvSynthStructuredValue(77, int(rows[i]), int(ja[i])) and as_[i] != 0
        H.freeSpmat(m)
        return lens, np.abs(ja - rows)
    lens, dist = gen(0, 40, 20, 10, "stencil.mtx")              # 3-D stencil, 18 neighbours
    assert lens.max() == 18 and lens.min() == 6 and lens.mean() > 14 and dist.max() == 40 * 20 + 40
    lens, dist = gen(1, 200_000, 0, 0, "road.mtx")              # road network: short rows, near-diagonal columns
    assert lens.max() <= 9 and 1.9 < lens.mean() < 2.4 and np.median(dist) <= 2 and lens.min() >= 1
    lens, dist = gen(2, 36417, 24, 0, "block.mtx")              # dense blocks along the diagonal
    assert 150 < lens.max() <= 204 and 100 < lens.mean() < 140
    assert H.spmvSynthWriteMtx(b"/nonexistent/dir/x.mtx", 0, 4, 4, 4, 1, None, None, None) != 0
    assert H.spmvSynthWriteMtx(str(tmp_path / "k.mtx").encode(), 9, 4, 4, 4, 1, None, None, None) != 0


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


def _archives(tmp_path):
    import bz2
    import gzip
    import lzma
    import zipfile
    raw = open(os.path.join(GOLD, "cage4like.mtx"), "rb").read()
    big = open(os.path.join(GOLD, "rand300.mtx"), "rb").read()
    out = []
    for name, data in (("a.mtx.gz", gzip.compress(raw)), ("a.mtx.bz2", bz2.compress(raw)), ("a.mtx.xz", lzma.compress(big))):
        (tmp_path / name).write_bytes(data)
        out.append(str(tmp_path / name))
    for name, method, data in (("a.mtx.zip", zipfile.ZIP_DEFLATED, big), ("s.mtx.zip", zipfile.ZIP_STORED, raw)):
        with zipfile.ZipFile(tmp_path / name, "w", method) as z:
            z.writestr("inner.mtx", data)
        out.append(str(tmp_path / name))
    return out


def test_loader_survives_damaged_files_under_sanitizers(tmp_path):
    """tests/harness/fuzz_loader.c: thousands of damaged MatrixMarket files and damaged .gz/.bz2/.xz/.zip archives through the
    host loader built with AddressSanitizer + UBSan; the in-memory parser and the serial fscanf steps must agree on every
    file (both reject, or the same entries in the same order)."""
    import subprocess
    probe = tmp_path / "probe.c"
    probe.write_text("int main(void) { return 0; }\n")
    if subprocess.run(["gcc", "-fsanitize=address,undefined", str(probe), "-o", str(tmp_path / "probe")], capture_output=True).returncode:
        pytest.skip("this gcc has no sanitizer runtimes")
    r = subprocess.run(["make", "-s", "fuzz"], cwd=ROOT, capture_output=True, text=True)
    assert r.returncode == 0, r.stdout + r.stderr
    work = tmp_path / "work"
    work.mkdir()
    env = dict(os.environ, ASAN_OPTIONS="allocator_may_return_null=1", OMP_NUM_THREADS="4")
    r = subprocess.run([os.path.join(ROOT, "tests/harness/fuzz_loader.elf"), str(work), "6000", "20261005"] + _archives(tmp_path),
                       capture_output=True, text=True, env=env, timeout=600)
    assert r.returncode == 0, r.stdout[-4000:] + r.stderr[-4000:]
    assert "ERROR: AddressSanitizer" not in r.stderr and "runtime error" not in r.stderr
    line = r.stdout.strip().splitlines()[-1]
    nums = [int(t) for t in line.replace(",", " ").replace(";", " ").split() if t.isdigit()]
    files, accepted, rejected, csr, ell, archives, inflated = nums
    assert files == 4500 and accepted + rejected == files and accepted > 500 and rejected > 500 and csr == ell > 300
    assert archives == 1500 and 0 < inflated < archives


def test_size_line_that_wraps_and_truncated_xz(tmp_path):
    """Found by the fuzzer: `%lu` reads "-29" as 2^64-29, and 24 bytes x such a count wraps to a small allocation (now refused
    before any product is formed); a truncated .xz made the decoder loop double its output buffer up to 64 GiB (liblzma answers
    BUF_ERROR for missing INPUT as well)."""
    import lzma
    import time
    for count in ("-29", "768614336404564651", "18446744073709551615"):
        p = tmp_path / "wrap.mtx"
        p.write_text("%%MatrixMarket matrix coordinate real general\n3 3 " + count + "\n1 1 1.0\n2 2 2.0\n")
        assert not H.MMtoCSR(str(p).encode())
        fifo = tmp_path / "wrap.fifo"                                  # the serial path alone (not a regular file)
        os.mkfifo(fifo)
        import threading
        t = threading.Thread(target=lambda: open(fifo, "w").write(p.read_text()))
        t.start()
        assert not H.MMtoCSR(str(fifo).encode())
        t.join()
        os.remove(fifo)
    H.extractInTmpFS.argtypes = [C.c_char_p, C.c_char_p]
    data = lzma.compress(open(os.path.join(GOLD, "rand300.mtx"), "rb").read())
    src = tmp_path / "cut.mtx.xz"
    src.write_bytes(data[: len(data) // 2])
    t0 = time.time()
    assert H.extractInTmpFS(str(src).encode(), str(tmp_path / "cut.mtx").encode()) != 0
    assert time.time() - t0 < 5


def test_structured_generators_write_pattern_files(tmp_path):
    """kind | 16: the same structure as a `pattern` file (what the DIMACS10 graphs of the reference's report are): no value
    column, and the loader gives every entry 1.0 (parser.c:59-61)."""
    Mv, NZv, mx = C.c_ulong(), C.c_ulong(), C.c_ulong()
    valued, pat = str(tmp_path / "v.mtx").encode(), str(tmp_path / "p.mtx").encode()
    assert H.spmvSynthWriteMtx(valued, 1, 50_000, 0, 0, 5, C.byref(Mv), C.byref(NZv), C.byref(mx)) == 0
    assert H.spmvSynthWriteMtx(pat, 1 | 16, 50_000, 0, 0, 5, C.byref(Mv), C.byref(NZv), C.byref(mx)) == 0
    assert open(pat, "rb").readline() == b"%%MatrixMarket matrix coordinate pattern general\n"
    assert os.path.getsize(pat) < os.path.getsize(valued)
    a, b = H.MMtoCSR(valued).contents, H.MMtoCSR(pat).contents
    assert (a.M, a.NZ) == (b.M, b.NZ) == (Mv.value, NZv.value)
    assert np.array_equal(_arr(a.IRP, a.M + 1, np.int64), _arr(b.IRP, b.M + 1, np.int64))
    assert np.array_equal(_arr(a.JA, a.NZ, np.int64), _arr(b.JA, b.NZ, np.int64))
    assert np.all(_arr(b.AS, b.NZ, np.float64) == 1.0) and not np.all(_arr(a.AS, a.NZ, np.float64) == 1.0)

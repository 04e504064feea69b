"""GPU parity: every HIP launcher (through the C-ABI) against the serial oracle.

Gate = the reference's own: |y_gpu - y_oracle| <= 7e-4 per element
(src/include/config.h:113, src/commons/utils.c:362-393) with NaN = failure, y
poisoned before every launch.  On top of it:
  * kernels that add a row's products in ascending-j order must be BIT-IDENTICAL
    to sgemvSerial (hipSpMVRowsCSR both variants, hipSpMVRowsELL,
    hipSpMVRowsELLNNTransposed);
  * shuffle-tree kernels must satisfy |dy| <= 1e-13 * sum|a_ij x_j|.
"""
import numpy as np
import pytest

from conftest import random_csr, tight_error

pytestmark = pytest.mark.gpu

GATE = 7e-4
TIGHT = 1e-13


@pytest.fixture(scope="module")
def api():
    from spmv_openmp_cuda_amd import api as a
    a.spmvHipInit(0)
    yield a
    a.spmvHipFinalize()


def _cases():
    rng = np.random.default_rng(1234)
    cases = {}
    # irregular: empty rows, single entries, rows straddling the 2048-nnz block size, long rows
    M, N = 3000, 5000
    lens = rng.integers(0, 40, size=M)
    lens[::7] = 0
    lens[5] = 2048
    lens[6] = 2049          # first row of the long-row path
    lens[100] = 4999
    lens[2999] = 3000
    cases["irregular"] = (M, N) + random_csr(rng, M, N, lens)
    # cage4-shaped: 9x9, 49 nnz, max row 6
    lens = np.array([5, 6, 5, 6, 5, 6, 5, 6, 5])
    cases["cage4-shaped"] = (9, 9) + random_csr(rng, 9, 9, lens)
    # all-empty and single-row
    cases["empty-rows"] = (70, 10) + random_csr(rng, 70, 10, np.zeros(70, dtype=int))
    cases["one-long-row"] = (1, 20000) + random_csr(rng, 1, 20000, np.array([20000]))
    # many 1-nnz rows (more than STREAM_MAX_ROWS per block)
    cases["ones"] = (10000, 300) + random_csr(rng, 10000, 300, np.ones(10000, dtype=int))
    # uniform 32/row
    cases["uniform32"] = (4096, 4096) + random_csr(rng, 4096, 4096, np.full(4096, 32))
    # banded: every row block's columns fit the 2 Ki-column LDS x tile (incl. the clipped edges)
    Mb = 6000
    lb = rng.integers(1, 30, size=Mb)
    IRPb = np.zeros(Mb + 1, dtype=np.uint64)
    IRPb[1:] = np.cumsum(lb)
    JAb = np.empty(int(IRPb[-1]), dtype=np.uint64)
    for r in range(Mb):
        lo, hi = max(0, r - 300), min(Mb, r + 300)
        JAb[int(IRPb[r]):int(IRPb[r + 1])] = np.sort(rng.choice(np.arange(lo, hi), size=int(lb[r]), replace=False))
    cases["banded"] = (Mb, Mb, IRPb, JAb, rng.uniform(-1, 1, size=JAb.size))
    # wide and tall enough for several 16 Ki-column slices and several row bins (tiles kernel)
    Mw, Nw = 40000, 70000
    lw = rng.integers(0, 12, size=Mw)
    lw[123] = 30000
    lw[39999] = 5000
    cases["wide"] = (Mw, Nw) + random_csr(rng, Mw, Nw, lw)
    # very wide and nearly empty: 128 column-sorted entries of a row bin span more than 2^17 columns (the stripes
    # kernel's 32-bit-column encoding); column ids up to 4 M
    Mh, Nh = 3000, 4_000_000
    lh = rng.integers(0, 4, size=Mh)
    IRPh = np.zeros(Mh + 1, dtype=np.uint64)
    IRPh[1:] = np.cumsum(lh)
    JAh = np.empty(int(IRPh[-1]), dtype=np.uint64)
    for r in range(Mh):
        JAh[int(IRPh[r]):int(IRPh[r + 1])] = np.sort(rng.choice(Nh, size=int(lh[r]), replace=False))
    cases["hyper-sparse"] = (Mh, Nh, IRPh, JAh, rng.uniform(-1, 1, size=JAh.size))
    return cases


CASES = _cases()


def _x(rng, n):
    return np.sin(rng.uniform(0, 2 * np.pi, size=n)) * 3e-5


def _run(api, launcher, dmat, x, rows):
    dx = api.DeviceVector(x.size).up(x)
    dy = api.DeviceVector(rows)
    dy.poison()
    api.spmv(launcher, dmat, dx, dy)
    y = dy.down()
    dx.free()
    dy.free()
    return y


@pytest.fixture(autouse=True)
def _default_variants(api):
    """every test starts (and leaves) with the default kernel variants"""
    api.set_variant("hipSpMVRowsCSR", 2)
    api.set_variant("hipSpMVWarpPerRowCSR", 2)
    api.set_variant("hipSpMVRowsELLNNTransposed", 1)
    yield
    api.set_variant("hipSpMVRowsCSR", 2)
    api.set_variant("hipSpMVWarpPerRowCSR", 2)
    api.set_variant("hipSpMVRowsELLNNTransposed", 1)


@pytest.mark.parametrize("name", list(CASES))
@pytest.mark.parametrize("launcher,variant,exact", [
    ("hipSpMVRowsCSR", 0, True), ("hipSpMVRowsCSR", 1, True), ("hipSpMVRowsCSR", 2, True),
    ("hipSpMVWarpPerRowCSR", 0, False), ("hipSpMVWarpPerRowCSR", 1, False), ("hipSpMVWarpPerRowCSR", 2, False),
    ("hipSpMVTilesCSR", -1, False), ("hipSpMVStripesCSR", -1, False), ("hipSpMVRowsSELL", -1, False)])
def test_csr(api, oracle, name, launcher, variant, exact):
    M, N, IRP, JA, AS = CASES[name]
    x = _x(np.random.default_rng(7), N)
    y_ref = oracle.csr_serial(IRP, JA, AS, x)
    host = api.HostCSR(M, N, IRP, JA, AS)
    dmat = api.spMatCpyCSR(host)
    if variant >= 0:
        api.set_variant(launcher, variant)
    y = _run(api, launcher, dmat, x, M)
    if launcher == "hipSpMVRowsSELL":                   # one lane per row, ascending j: exact for rows <= 256 entries
        short = np.diff(IRP.astype(np.int64)) <= 256
        assert np.array_equal(y[short], y_ref[short])
    if launcher in ("hipSpMVTilesCSR", "hipSpMVStripesCSR") and JA.size:       # second call re-uses the built format
        assert np.max(np.abs(_run(api, launcher, dmat, x, M) - y)) <= 1e-15    # arrival-order sums differ in the last bits
    if launcher == "hipSpMVStripesCSR" and JA.size:
        import ctypes as C
        wide = C.c_int(-1)
        assert api.lib.spmvHipStripesShape(C.byref(dmat.handle), None, None, C.byref(wide), None) == 0
        assert wide.value == (1 if name == "hyper-sparse" else 0)
        # what spmvHipLastLaunch reports is what ran: min(bins, CUs) persistent workgroups of 256 threads
        info = api.stripes_info(dmat)
        assert api.last_launch() == ((info.grid, 1, 1), (256, 1, 1)) and 1 <= info.grid <= info.nBins
    dmat.free()
    assert not np.isnan(y).any(), "rows left unwritten (poison survived)"
    assert np.max(np.abs(y - y_ref), initial=0.0) <= GATE
    if exact:
        assert np.array_equal(y, y_ref)
    else:
        assert tight_error(IRP, JA, AS, x, y_ref, y) <= TIGHT


@pytest.mark.parametrize("name", ["irregular", "cage4-shaped", "ones", "uniform32", "empty-rows"])
@pytest.mark.parametrize("rowlens", [True, False])
@pytest.mark.parametrize("launcher,transposed,exact", [
    ("hipSpMVRowsELL", True, True),
    ("hipSpMVRowsELLNNTransposed", False, True),          # default variant 1: LDS-stream kernel, one thread sums its row
    ("hipSpMVRowsELLNNTransposed:0", False, True),        # variant 0: a thread walks its row in global memory
    ("hipSpMVWarpsPerRowELLNTrasposed", False, False)])
def test_ell(api, oracle, name, rowlens, launcher, transposed, exact):
    if launcher.endswith(":0"):
        launcher = launcher[:-2]
        api.set_variant(launcher, 0)
    M, N, IRP, JA, AS = CASES[name]
    x = _x(np.random.default_rng(8), N)
    y_ref = oracle.csr_serial(IRP, JA, AS, x)
    ell = api.HostCSR(M, N, IRP, JA, AS).to_ell(with_row_lens=rowlens)
    dmat = api.spMatCpyELL(ell.transpose() if transposed else ell)
    api.lib.spmvHipSetEllRowLens(1 if rowlens else 0)
    y = _run(api, launcher, dmat, x, M)
    dmat.free()
    api.lib.spmvHipSetEllRowLens(1)
    assert not np.isnan(y).any()
    assert np.max(np.abs(y - y_ref), initial=0.0) <= GATE
    if exact:
        # padding adds +0.0*x[0] terms: exact unless the row sum is -0.0
        assert np.array_equal(y, y_ref + 0.0)
    else:
        assert tight_error(IRP, JA, AS, x, y_ref, y) <= TIGHT


def test_wrong_handle_kind_fails_loudly(api):
    M, N, IRP, JA, AS = CASES["cage4-shaped"]
    host = api.HostCSR(M, N, IRP, JA, AS)
    dmat = api.spMatCpyCSR(host)
    dx, dy = api.DeviceVector(N), api.DeviceVector(M)
    with pytest.raises(api.SpmvHipError):
        api.spmv("hipSpMVRowsELL", dmat, dx, dy)
    dmat.free()


def test_synth_device_matches_twin_and_oracle(api, oracle):
    """Device generator == CPU twin bit-for-bit; SpMV on it == oracle."""
    from spmv_openmp_cuda_amd import synth
    for key in ("tiny", "tinyu"):
        for band in (0, 64):
            w = synth.WORKLOADS[key]
            w = synth.Workload(w.name, w.N, w.nnz, w.law, w.max_row, w.cfg, band)
            lens = synth.row_lengths(w)
            irp = synth.prefix(lens)
            dm = synth.device_csr(w, irp, 0, w.N)
            ja_dev = dm.buffers["JA"].down(np.uint32)
            as_dev = dm.buffers["AS"].down(np.float64)
            ja_ref, as_ref = oracle.synth_fill(w.N, 0, irp, synth.SEED_STRUCT + w.cfg, synth.SEED_VAL + w.cfg, band)
            assert np.array_equal(ja_dev, ja_ref) and np.array_equal(as_dev, as_ref)
            # columns ascending and distinct within each row, inside [0,N)
            d = np.diff(ja_ref.astype(np.int64))
            row_starts = irp[1:-1].astype(np.int64)
            mask = np.ones(d.size, dtype=bool)
            mask[row_starts[(row_starts > 0) & (row_starts < ja_ref.size)] - 1] = False
            assert (d[mask] > 0).all() and ja_ref.max() < w.N
            x = synth.make_x(w.N, w.cfg)
            y_ref = oracle.csr_serial_dev(irp.astype(np.uint32), ja_ref, as_ref, x)
            for launcher, exact in (("hipSpMVRowsCSR", True), ("hipSpMVWarpPerRowCSR", False), ("hipSpMVTilesCSR", False),
                                        ("hipSpMVStripesCSR", False), ("hipSpMVRowsSELL", False)):
                y = _run(api, launcher, dm, x, w.N)
                assert np.max(np.abs(y - y_ref)) <= GATE
                if exact:
                    assert np.array_equal(y, y_ref)
            # a shard generated with a row offset equals the same rows of the whole matrix
            r0, r1 = w.N // 3, w.N // 3 + 1000
            sh = synth.device_csr(w, irp, r0, r1)
            b0, b1 = int(irp[r0]), int(irp[r1])
            assert np.array_equal(sh.buffers["JA"].down(np.uint32), ja_ref[b0:b1])
            assert np.array_equal(sh.buffers["AS"].down(np.float64), as_ref[b0:b1])
            sh.free()
            dm.free()


@pytest.mark.parametrize("key", ["c2", "c3", "c5"])
def test_full_size_spot_checks(api, oracle, key):
    """BASELINE.json's full sizes (1 M x 32, 10 M / 200 M and 80 M / 1.6 G): the whole y must be
    written (no poison left) and rows at the start, middle, end and around the
    heaviest row must match the oracle.  (A launch whose blocks*threads wrapped the
    32-bit work-item count once left most of c5 unfilled -- this is its regression test.)
    hipSpMVAutoCSR runs FIRST, on the fresh handle: its choice must be the stripes kernel on c2 and c3 (x fits the
    Infinity Cache) and the two-phase kernel on c5, and the losers must leave no format behind; hipSpMVWarpPerRowCSR in
    its default variant must then run that same choice."""
    import ctypes as C
    from spmv_openmp_cuda_amd import synth
    w = synth.WORKLOADS[key]
    lens = synth.row_lengths(w)
    irp = synth.prefix(lens)
    dm = synth.device_csr(w, irp, 0, w.N)
    x = synth.make_x(w.N, w.cfg)
    dx = api.DeviceVector(w.N).up(x)
    dy = api.DeviceVector(w.N)
    S = 100_000
    heavy = int(np.argmax(lens))
    ranges = [(0, S), (w.N // 2, w.N // 2 + S), (w.N - S, w.N), (max(0, heavy - 10), min(w.N, heavy + 10))]
    refs = []
    for r0, r1 in ranges:
        ja, as_ = oracle.synth_fill(w.N, r0, irp[r0:r1 + 1], synth.SEED_STRUCT + w.cfg, synth.SEED_VAL + w.cfg, w.band)
        # the device generator wrote the same entries
        b0, b1 = int(irp[r0]), int(irp[r1])
        ja_dev = np.empty(b1 - b0, dtype=np.uint32)
        api.lib.spmvHipMemcpyDown(ja_dev.ctypes.data_as(C.c_void_p), dm.buffers["JA"].ptr.value + 4 * b0, 4 * (b1 - b0))
        assert np.array_equal(ja_dev, ja)
        refs.append(oracle.csr_serial_dev((irp[r0:r1 + 1] - irp[r0]).astype(np.uint32), ja, as_, x))
    expect = b"hipSpMVTilesCSR" if key == "c5" else b"hipSpMVStripesCSR"
    for launcher, variant, exact in (("hipSpMVAutoCSR", -1, False), ("hipSpMVWarpPerRowCSR", 2, False), ("hipSpMVWarpPerRowCSR", 1, False),
                                     ("hipSpMVRowsCSR", 1, True), ("hipSpMVRowsCSR", 2, True), ("hipSpMVTilesCSR", -1, False),
                                     ("hipSpMVStripesCSR", -1, False), ("hipSpMVRowsSELL", -1, False)):
        if variant >= 0:
            api.set_variant(launcher, variant)
        dy.poison()
        api.spmv(launcher, dm, dx, dy)
        y = dy.down()
        assert not np.isnan(y).any(), "poison survived: some rows were never written"
        for (r0, r1), yr in zip(ranges, refs):
            assert np.max(np.abs(y[r0:r1] - yr)) <= GATE
            if exact:
                assert np.array_equal(y[r0:r1], yr)
        arrival_order = launcher in ("hipSpMVTilesCSR", "hipSpMVStripesCSR", "hipSpMVAutoCSR") or (launcher, variant) == ("hipSpMVWarpPerRowCSR", 2)
        if launcher == "hipSpMVAutoCSR" or (launcher, variant) == ("hipSpMVWarpPerRowCSR", 2):
            assert api.lib.spmvHipAutoChoice(C.byref(dm.handle), None) == expect
            tb, sb = api.lib.spmvHipTilesBytes(C.byref(dm.handle)), api.lib.spmvHipStripesBytes(C.byref(dm.handle))
            assert (tb > 0) == (expect == b"hipSpMVTilesCSR") and (sb > 0) == (expect == b"hipSpMVStripesCSR")
        if (launcher, variant) == ("hipSpMVRowsCSR", 2):    # the serial-order selection: measured, a format kernel wins at this size
            ms = (C.c_double * 4)()
            pick = api.lib.spmvHipAutoChoiceRows(C.byref(dm.handle), ms)
            names = (b"hipSpMVRowsCSR", b"hipSpMVTilesCSR(deterministic)", b"hipSpMVStripesCSR(owner wavefronts)", b"hipSpMVStripesCSR(ordered tickets)")
            assert pick in names[1:], pick
            assert ms[0] > 0 and min(t for t in ms if t > 0) == ms[names.index(pick)]
        # linearity (size-independent property): A(2x) == 2 A(x) exactly in binary fp
        dx2 = api.DeviceVector(w.N).up(2.0 * x)
        dy.poison()
        api.spmv(launcher, dm, dx2, dy)
        if arrival_order:                                   # arrival-order sums: linear up to rounding only
            assert np.max(np.abs(dy.down() - 2.0 * y)) <= 1e-15
        else:
            assert np.array_equal(dy.down(), 2.0 * y)
        dx2.free()
    dm.free()


@pytest.mark.parametrize("name", ["irregular", "cage4-shaped", "empty-rows", "uniform32"])
def test_device_csr_to_ell(api, oracle, name):
    """spmvHipCsrToEll (device-side conversion) gives the same results as host to_ell/ellTranspose + spMatCpyELL."""
    M, N, IRP, JA, AS = CASES[name]
    x = _x(np.random.default_rng(11), N)
    y_ref = oracle.csr_serial(IRP, JA, AS, x)
    dcsr = api.spMatCpyCSR(api.HostCSR(M, N, IRP, JA, AS))
    for transposed, launcher in ((True, "hipSpMVRowsELL"), (False, "hipSpMVRowsELLNNTransposed"),
                                 (False, "hipSpMVWarpsPerRowELLNTrasposed")):
        dell = api.csr_to_ell_device(dcsr, transposed)
        for rl in (1, 0):
            api.lib.spmvHipSetEllRowLens(rl)
            y = _run(api, launcher, dell, x, M)
            assert not np.isnan(y).any() and np.max(np.abs(y - y_ref), initial=0.0) <= GATE
            if launcher != "hipSpMVWarpsPerRowELLNTrasposed":
                assert np.array_equal(y, y_ref + 0.0)
        dell.free()
    api.lib.spmvHipSetEllRowLens(1)
    dcsr.free()


def test_64bit_row_pointers_small(api, oracle):
    """The uint64_t row-pointer instantiations of every CSR kernel, on a small adopted matrix."""
    import ctypes as C
    from spmv_openmp_cuda_amd import synth
    M, N, IRP, JA, AS = CASES["irregular"]
    x = _x(np.random.default_rng(3), N)
    y_ref = oracle.csr_serial(IRP, JA, AS, x)
    d_irp = api.DeviceBuffer(IRP.nbytes).up(IRP.astype(np.uint64))
    d_ja = api.DeviceBuffer(4 * JA.size).up(JA.astype(np.uint32))
    d_as = api.DeviceBuffer(8 * AS.size).up(AS)
    dm = api.DeviceMatrix()
    assert api.lib.spmvHipAdoptCSR(C.byref(dm.handle), M, N, JA.size, d_irp.ptr, 8, d_ja.ptr, d_as.ptr, None) == 0
    dm.keep = [d_irp, d_ja, d_as]
    for launcher, variants, exact in (("hipSpMVRowsCSR", (0, 1, 2), True), ("hipSpMVWarpPerRowCSR", (0, 1, 2), False),
                                      ("hipSpMVTilesCSR", (-1,), False), ("hipSpMVStripesCSR", (-1,), False),
                                      ("hipSpMVRowsSELL", (-1,), False)):
        for v in variants:
            if v >= 0:
                api.set_variant(launcher, v)
            y = _run(api, launcher, dm, x, M)
            assert not np.isnan(y).any() and np.max(np.abs(y - y_ref)) <= GATE
            if exact:
                assert np.array_equal(y, y_ref)
    api.set_variant("hipSpMVRowsCSR", 1)
    api.set_variant("hipSpMVWarpPerRowCSR", 2)
    dm.free()


def test_more_than_4g_nnz(api, oracle):
    """Maximum-size edge: nnz >= 2^32 (4.4 G entries, ~53 GB of HBM) forces 64-bit row pointers end to end.
    Rows at the head, across the 2^32-entry boundary and at the tail must match the oracle; the
    two-phase kernel (32-bit positions) must refuse cleanly."""
    import ctypes as C
    from spmv_openmp_cuda_amd import synth
    w = synth.Workload("huge-uniform-100M-44", 100_000_000, 4_400_000_000, "uniform", cfg=7)
    lens = synth.row_lengths(w)
    irp = synth.prefix(lens)
    assert int(irp[-1]) >= 1 << 32
    dm = synth.device_csr(w, irp, 0, w.N)
    assert dm.irp_bytes == 8
    x = synth.make_x(w.N, w.cfg)
    dx = api.DeviceVector(w.N).up(x)
    dy = api.DeviceVector(w.N)
    cross = int(np.searchsorted(irp, 1 << 32)) - 50_000
    ranges = [(0, 100_000), (cross, cross + 100_000), (w.N - 100_000, w.N)]
    for launcher, exact in (("hipSpMVWarpPerRowCSR", False), ("hipSpMVRowsCSR", True)):
        dy.poison()
        api.spmv(launcher, dm, dx, dy)
        y = dy.down()
        assert not np.isnan(y).any()
        for r0, r1 in ranges:
            ja, as_ = oracle.synth_fill(w.N, r0, irp[r0:r1 + 1], synth.SEED_STRUCT + w.cfg, synth.SEED_VAL + w.cfg, 0)
            yr = oracle.csr_serial_dev((irp[r0:r1 + 1] - irp[r0]).astype(np.uint32), ja, as_, x)
            assert np.max(np.abs(y[r0:r1] - yr)) <= GATE
            if exact:
                assert np.array_equal(y[r0:r1], yr)
    for launcher in ("hipSpMVTilesCSR", "hipSpMVStripesCSR"):
        with pytest.raises(api.SpmvHipError):
            api.spmv(launcher, dm, dx, dy)
    dm.free()


def test_upload_validation_and_handle_lifecycle(api, capfd):
    """Bad inputs are refused with EXIT_FAILURE (never a crash), freed handles are rejected."""
    import ctypes as C
    M, N, IRP, JA, AS = CASES["cage4-shaped"]
    bad_col = JA.copy()
    bad_col[3] = N + 5                                    # column outside the matrix
    with pytest.raises(api.SpmvHipError):
        api.spMatCpyCSR(api.HostCSR(M, N, IRP, bad_col, AS))
    bad_irp = IRP.copy()
    bad_irp[-1] += 1                                      # IRP[M] != NZ
    h = api.HostCSR(M, N, IRP, JA, AS)
    h.IRP[-1] += 1
    with pytest.raises(api.SpmvHipError):
        api.spMatCpyCSR(h)
    h.IRP[-1] -= 1
    dmat = api.spMatCpyCSR(h)
    dx, dy = api.DeviceVector(N), api.DeviceVector(M)
    api.spmv("hipSpMVRowsCSR", dmat, dx.up(np.ones(N)), dy)
    assert api.lib.hipFreeSpmat(C.byref(dmat.handle)) == 0
    assert api.lib.hipFreeSpmat(C.byref(dmat.handle)) == 0          # idempotent
    with pytest.raises(api.SpmvHipError):
        api.spmv("hipSpMVRowsCSR", dmat, dx, dy)                    # freed handle
    # a block size that is not a multiple of the 64-lane wavefront falls back to the default (with a message)
    dmat = api.spMatCpyCSR(h)
    cfg = api.CONFIG()
    cfg.blockSize.x = 100
    api.set_variant("hipSpMVRowsCSR", 0)
    api.spmv("hipSpMVRowsCSR", dmat, dx, dy, cfg)
    assert api.last_launch()[1][0] == 256
    cfg.blockSize.x = 128
    api.spmv("hipSpMVRowsCSR", dmat, dx, dy, cfg)
    assert api.last_launch()[1][0] == 128
    api.set_variant("hipSpMVRowsCSR", 1)
    dmat.free()
    assert "not a multiple of the 64-lane wavefront" in capfd.readouterr().err


def test_config_block_sizes(api, oracle):
    """CONFIG.blockSize (the reference's BLOCKS_1D knob, config.h:102-105) up to 1024 threads on the
    kernels that honour it."""
    M, N, IRP, JA, AS = CASES["irregular"]
    x = _x(np.random.default_rng(21), N)
    y_ref = oracle.csr_serial(IRP, JA, AS, x)
    host = api.HostCSR(M, N, IRP, JA, AS)
    dcsr = api.spMatCpyCSR(host)
    ell = host.to_ell()
    dell_t, dell = api.spMatCpyELL(ell.transpose()), api.spMatCpyELL(ell)
    dx, dy = api.DeviceVector(N).up(x), api.DeviceVector(M)
    api.set_variant("hipSpMVRowsCSR", 0)
    api.set_variant("hipSpMVWarpPerRowCSR", 0)
    for bx in (64, 192, 512, 1024):
        cfg = api.CONFIG()
        cfg.blockSize.x = bx
        for launcher, mat in (("hipSpMVRowsCSR", dcsr), ("hipSpMVWarpPerRowCSR", dcsr), ("hipSpMVRowsELL", dell_t),
                              ("hipSpMVRowsELLNNTransposed", dell)):
            dy.poison()
            api.spmv(launcher, mat, dx, dy, cfg)
            assert api.last_launch()[1][0] == bx
            y = dy.down()
            assert not np.isnan(y).any() and np.max(np.abs(y - y_ref)) <= GATE, (launcher, bx)
    api.set_variant("hipSpMVRowsCSR", 1)
    api.set_variant("hipSpMVWarpPerRowCSR", 2)
    for m in (dcsr, dell_t, dell):
        m.free()


def test_degenerate_shapes_all_launchers(api, oracle):
    """1x1, a single empty row, a single column, rows >> cols: every CSR-upload launcher, no crash, exact zeros
    where the row is empty."""
    rng = np.random.default_rng(77)
    shapes = {
        "1x1": (1, 1, np.array([0, 1], dtype=np.uint64), np.array([0], dtype=np.uint64), np.array([2.5])),
        "1x1-empty": (1, 1, np.array([0, 0], dtype=np.uint64), np.zeros(0, dtype=np.uint64), np.zeros(0)),
        "one-column": (500, 1) + random_csr(rng, 500, 1, rng.integers(0, 2, size=500)),
        "tall": (70000, 3) + random_csr(rng, 70000, 3, rng.integers(0, 4, size=70000)),
    }
    for name, (M, N, IRP, JA, AS) in shapes.items():
        x = np.linspace(1e-5, 3e-5, N)
        y_ref = oracle.csr_serial(IRP, JA, AS, x)
        dm = api.spMatCpyCSR(api.HostCSR(M, N, IRP, JA, AS))
        for launcher in ("hipSpMVRowsCSR", "hipSpMVWarpPerRowCSR", "hipSpMVTilesCSR", "hipSpMVStripesCSR", "hipSpMVRowsSELL"):
            y = _run(api, launcher, dm, x, M)
            assert not np.isnan(y).any(), (name, launcher)
            assert np.max(np.abs(y - y_ref), initial=0.0) <= 1e-18, (name, launcher)
        for transposed, launcher in ((True, "hipSpMVRowsELL"), (False, "hipSpMVWarpsPerRowELLNTrasposed")):
            de = api.csr_to_ell_device(dm, transposed)
            y = _run(api, launcher, de, x, M)
            assert not np.isnan(y).any() and np.max(np.abs(y - y_ref), initial=0.0) <= 1e-18, (name, launcher)
            de.free()
        dm.free()


def test_config4_full_size_ell_vs_csr(api, oracle):
    """BASELINE config 4 at full size: the 10 M-row power-law matrix clipped to 64 slots, in ELL (transposed + pitched
    thread-per-row; row-major lanes-per-row), each with and without the row-length early exit, against CSR on the SAME
    clipped matrix -- all through the C-ABI, checked on oracle windows (head, middle, tail, around the longest rows);
    the thread-per-row ELL kernel must equal the serial-order CSR kernel bit for bit.  And the ELL size guard: the
    UNCLIPPED matrix (50 k slots = 6 TB) is refused by spmvHipCsrToEll before any allocation (the reference's loader
    refuses it too, src/lib/parser.c:223-232)."""
    import ctypes as C
    from spmv_openmp_cuda_amd import synth
    w = synth.WORKLOADS["c4"]
    lens = synth.row_lengths(w)
    assert int(lens.max()) == 64
    irp = synth.prefix(lens)
    dm = synth.device_csr(w, irp, 0, w.N)
    x = synth.make_x(w.N, w.cfg)
    dx, dy = api.DeviceVector(w.N).up(x), api.DeviceVector(w.N)
    S = 100_000
    full = int(np.argmax(lens))
    ranges = [(0, S), (w.N // 2, w.N // 2 + S), (w.N - S, w.N), (max(0, full - 50), min(w.N, full + 50))]
    refs, scales = [], []
    for r0, r1 in ranges:
        ja, as_ = oracle.synth_fill(w.N, r0, irp[r0:r1 + 1], synth.SEED_STRUCT + w.cfg, synth.SEED_VAL + w.cfg, w.band)
        il = (irp[r0:r1 + 1] - irp[r0])
        refs.append(oracle.csr_serial_dev(il.astype(np.uint32), ja, as_, x))
        scales.append(np.add.reduceat(np.abs(as_ * x[ja]), il[:-1].astype(np.int64)))       # no empty rows in this workload
    api.spmv("hipSpMVRowsCSR", dm, dx, dy)
    y_csr = dy.down()
    for (r0, r1), yr in zip(ranges, refs):
        assert np.array_equal(y_csr[r0:r1], yr)
    for transposed, launcher, exact in ((True, "hipSpMVRowsELL", True), (False, "hipSpMVWarpsPerRowELLNTrasposed", False)):
        dell = api.csr_to_ell_device(dm, transposed)
        for rl in (1, 0):
            api.lib.spmvHipSetEllRowLens(rl)
            dy.poison()
            api.spmv(launcher, dell, dx, dy)
            y = dy.down()
            assert not np.isnan(y).any(), (launcher, rl)
            if exact:
                assert np.array_equal(y, y_csr + 0.0), (launcher, rl)         # padding adds +0.0 * x[0] terms
            for (r0, r1), yr, sc in zip(ranges, refs, scales):
                assert np.max(np.abs(y[r0:r1] - yr)) <= GATE
                assert np.all(np.abs(y[r0:r1] - yr) <= TIGHT * sc), (launcher, rl)
        dell.free()
    api.lib.spmvHipSetEllRowLens(1)
    dm.free()
    # size guard on the unclipped matrix: only its row pointers are needed to see that it cannot be ELL
    w3 = synth.WORKLOADS["c3"]
    lens3 = synth.row_lengths(w3)
    irp3 = synth.prefix(lens3)
    dm3 = synth.device_csr(w3, irp3, 0, w3.N)
    bad = api.DeviceMatrix()
    for transposed in (0, 1):
        assert api.lib.spmvHipCsrToEll(C.byref(dm3.handle), transposed, C.byref(bad.handle)) == 1
        assert not bad.handle.dev
    dm3.free()


def test_host_pointer_wrappers_and_cache(api, oracle):
    """The SPMV_INTERF-style wrappers (host vectors in and out, matrix uploaded and cached): all four against the oracle;
    a host struct that is freed and re-allocated at the SAME address with another matrix must not get the old device
    copy (the cache remembers shape and array pointers), and spmvHipDropCache() empties it."""
    import ctypes as C
    lib = api.lib
    M, N, IRP, JA, AS = CASES["irregular"]
    x = _x(np.random.default_rng(31), N)
    y_ref = oracle.csr_serial(IRP, JA, AS, x)
    host = api.HostCSR(M, N, IRP, JA, AS)
    ell = host.to_ell()
    y = np.empty(M)
    vp = C.c_void_p
    for fn, mat, exact in ((lib.spmvHipRowsCSR, host, True), (lib.spmvHipWarpPerRowCSR, host, False),
                           (lib.spmvHipRowsELL, ell, True), (lib.spmvHipWarpsPerRowELL, ell, False)):
        for _ in range(2):                                  # second call: cached device copy
            y[:] = np.nan
            assert fn(C.byref(mat.struct), x.ctypes.data_as(vp), None, y.ctypes.data_as(vp)) == 0
            assert not np.isnan(y).any() and np.max(np.abs(y - y_ref)) <= GATE
            if exact:
                assert np.array_equal(y, y_ref + 0.0)
    # same struct address, other matrix (as after free + malloc): shape and pointers differ -> re-upload
    M2, N2, IRP2, JA2, AS2 = CASES["uniform32"]
    host2 = api.HostCSR(M2, N2, IRP2, JA2, AS2)
    x2 = _x(np.random.default_rng(32), N2)
    y2_ref = oracle.csr_serial(IRP2, JA2, AS2, x2)
    C.memmove(C.byref(host.struct), C.byref(host2.struct), C.sizeof(host.struct))
    y2 = np.full(M2, np.nan)
    assert lib.spmvHipRowsCSR(C.byref(host.struct), x2.ctypes.data_as(vp), None, y2.ctypes.data_as(vp)) == 0
    assert np.array_equal(y2, y2_ref)
    assert lib.spmvHipDropCache() == 0
    y2[:] = np.nan
    assert lib.spmvHipRowsCSR(C.byref(host.struct), x2.ctypes.data_as(vp), None, y2.ctypes.data_as(vp)) == 0
    assert np.array_equal(y2, y2_ref)
    assert lib.spmvHipDropCache() == 0


def test_upload_rejects_decreasing_row_pointers_and_bad_transposed_columns(api):
    """IRP = {0, 100, 5} with NZ = 5 passes an ends-only check and would make the kernels read past JA/AS; a transposed
    ELL upload checks its column ids against the column count ellTranspose() keeps."""
    M, N, IRP, JA, AS = CASES["cage4-shaped"]
    bad = IRP.copy()
    bad[3] = bad[4] + 7                                   # decreases at row 3 -> 4, ends still consistent
    with pytest.raises(api.SpmvHipError):
        api.spMatCpyCSR(api.HostCSR(M, N, bad, JA, AS))
    ell_t = api.HostCSR(M, N, IRP, JA, AS).to_ell().transpose()
    ell_t.JA[0, 0] = N + 3                                # column outside the matrix, in the transposed layout
    with pytest.raises(api.SpmvHipError):
        api.spMatCpyELL(ell_t)


@pytest.mark.parametrize("name", ["irregular", "wide", "banded", "ones"])
def test_stripes_many_small_bins(api, oracle, name):
    """The stripes kernel with bins of at most 64 rows (spmvHipBuildStripesOpt): hundreds of bins on a small matrix,
    i.e. several bins per persistent workgroup, bins without entries, single-row bins holding a long row."""
    M, N, IRP, JA, AS = CASES[name]
    x = _x(np.random.default_rng(17), N)
    y_ref = oracle.csr_serial(IRP, JA, AS, x)
    dmat = api.spMatCpyCSR(api.HostCSR(M, N, IRP, JA, AS))
    api.build_stripes(dmat, rowsPerBin=64)
    y = _run(api, "hipSpMVStripesCSR", dmat, x, M)
    info = api.stripes_info(dmat)
    assert info.rowsPerBin <= 64 and info.nBins >= (M + 63) // 64 and info.buildMs > 0 and info.bytes > 0
    dmat.free()
    assert not np.isnan(y).any()
    assert np.max(np.abs(y - y_ref), initial=0.0) <= GATE
    assert tight_error(IRP, JA, AS, x, y_ref, y) <= TIGHT


def test_auto_launcher_picks_and_remembers(api, oracle):
    """hipSpMVAutoCSR: small matrices go to the LDS-stream kernel without a measurement; from 2^18 entries on the
    eligible candidates are timed on the caller's x, the choice is reported and remembered, the losers' formats are
    released, and y equals the oracle on the first call and on every later one."""
    import ctypes as C
    # small: no private format is ever built
    M, N, IRP, JA, AS = CASES["irregular"]
    x = _x(np.random.default_rng(31), N)
    y_ref = oracle.csr_serial(IRP, JA, AS, x)
    dmat = api.spMatCpyCSR(api.HostCSR(M, N, IRP, JA, AS))
    assert api.lib.spmvHipAutoChoice(C.byref(dmat.handle), None) is None
    y = _run(api, "hipSpMVAutoCSR", dmat, x, M)
    assert tight_error(IRP, JA, AS, x, y_ref, y) <= TIGHT and not np.isnan(y).any()
    assert api.lib.spmvHipAutoChoice(C.byref(dmat.handle), None) == b"hipSpMVWarpPerRowCSR"
    assert api.lib.spmvHipTilesBytes(C.byref(dmat.handle)) == 0 and api.lib.spmvHipStripesBytes(C.byref(dmat.handle)) == 0
    dmat.free()
    # large enough to be measured: 2^18+ entries, uniform columns
    rng = np.random.default_rng(32)
    M2 = N2 = 40_000
    IRP2, JA2, AS2 = random_csr(rng, M2, N2, np.full(M2, 8))
    x2 = _x(rng, N2)
    y2_ref = oracle.csr_serial(IRP2, JA2, AS2, x2)
    d2 = api.spMatCpyCSR(api.HostCSR(M2, N2, IRP2, JA2, AS2))
    ms = (C.c_double * 4)()
    for call in range(3):
        y2 = _run(api, "hipSpMVAutoCSR", d2, x2, M2)
        assert not np.isnan(y2).any() and tight_error(IRP2, JA2, AS2, x2, y2_ref, y2) <= TIGHT, call
        name = api.lib.spmvHipAutoChoice(C.byref(d2.handle), ms)
        assert name in (b"hipSpMVWarpPerRowCSR", b"hipSpMVTilesCSR", b"hipSpMVStripesCSR")
        assert all(t > 0 for t in ms[:3]) and ms[3] == 0, list(ms)    # all three were eligible and measured
        assert ms[(b"hipSpMVWarpPerRowCSR", b"hipSpMVTilesCSR", b"hipSpMVStripesCSR").index(name)] == min(ms[:3])
    # only the winner keeps a private copy of the matrix
    tb, sb = api.lib.spmvHipTilesBytes(C.byref(d2.handle)), api.lib.spmvHipStripesBytes(C.byref(d2.handle))
    assert (tb > 0) == (name == b"hipSpMVTilesCSR") and (sb > 0) == (name == b"hipSpMVStripesCSR")
    # the serial-order selection (hipSpMVRowsCSR, default variant) is made separately, keeps its own winner beside the
    # other selection's, and whatever it picks gives the bits of the serial oracle -- on the first call and later
    assert api.lib.spmvHipAutoChoiceRows(C.byref(d2.handle), None) is None
    for call in range(3):
        yr = _run(api, "hipSpMVRowsCSR", d2, x2, M2)
        assert np.array_equal(yr, y2_ref), call
        rname = api.lib.spmvHipAutoChoiceRows(C.byref(d2.handle), ms)
        assert rname in (b"hipSpMVRowsCSR", b"hipSpMVTilesCSR(deterministic)", b"hipSpMVStripesCSR(owner wavefronts)",
                         b"hipSpMVStripesCSR(ordered tickets)") and all(t > 0 for t in ms)
    assert api.lib.spmvHipAutoChoice(C.byref(d2.handle), None) == name              # untouched
    y2 = _run(api, "hipSpMVAutoCSR", d2, x2, M2)                                    # and still runs its own format
    assert tight_error(IRP2, JA2, AS2, x2, y2_ref, y2) <= TIGHT
    tb2, sb2 = api.lib.spmvHipTilesBytes(C.byref(d2.handle)), api.lib.spmvHipStripesBytes(C.byref(d2.handle))
    assert tb2 >= tb and sb2 >= sb and (tb2 + sb2 > tb + sb) == (rname != b"hipSpMVRowsCSR")
    d2.free()
    # an ELL handle is refused
    Me, Ne, IRPe, JAe, ASe = CASES["cage4-shaped"]
    de = api.spMatCpyELL(api.HostCSR(Me, Ne, IRPe, JAe, ASe).to_ell())
    dx, dy = api.DeviceVector(Ne), api.DeviceVector(Me)
    with pytest.raises(api.SpmvHipError):
        api.spmv("hipSpMVAutoCSR", de, dx, dy)
    de.free()


def test_lds_atomics_add_in_lane_order_on_this_device(api):
    """The property the deterministic kernels rest on, probed directly (and consulted by the serial-order selection)."""
    assert api.lib.spmvHipProbeLdsAtomicOrder() == 1


def test_serial_order_contract_on_rows_with_unsorted_columns(api, oracle):
    """hipSpMVRowsCSR promises ascending-j sums.  Its deterministic format kernels add in ascending COLUMN order, which is
    the same only when the columns of every row ascend; a caller's CSR with shuffled rows must therefore be served by the
    kernel that walks j (no format is built), and y must still be the oracle's bits -- while the reduction-order selection
    is free to use any kernel."""
    import ctypes as C
    rng = np.random.default_rng(77)
    M = N = 50_000
    IRP, JA, AS = random_csr(rng, M, N, np.full(M, 8))               # 400 k entries: large enough for the selections to measure
    JA, AS = JA.copy(), AS.copy()
    for r in range(0, M, 3):                                          # shuffle the entries of every third row
        b, e = int(IRP[r]), int(IRP[r + 1])
        p = rng.permutation(e - b)
        JA[b:e], AS[b:e] = JA[b:e][p], AS[b:e][p]
    x = _x(rng, N)
    y_ref = oracle.csr_serial(IRP, JA, AS, x)                         # ascending j, whatever the columns
    dm = api.spMatCpyCSR(api.HostCSR(M, N, IRP, JA, AS))
    for _ in range(2):
        assert np.array_equal(_run(api, "hipSpMVRowsCSR", dm, x, M), y_ref)
    assert api.lib.spmvHipAutoChoiceRows(C.byref(dm.handle), None) == b"hipSpMVRowsCSR"
    assert api.lib.spmvHipTilesBytes(C.byref(dm.handle)) == 0 and api.lib.spmvHipStripesBytes(C.byref(dm.handle)) == 0
    y = _run(api, "hipSpMVWarpPerRowCSR", dm, x, M)
    assert tight_error(IRP, JA, AS, x, y_ref, y) <= TIGHT
    dm.free()


def test_launchers_capture_into_a_hip_graph(api, oracle):
    """With spmvHipSetSync(0) a launcher only enqueues kernels on the library stream -- no event, no allocation, no
    synchronisation once its format exists -- so a solver's inner loop can be captured into a HIP graph (torch's
    CUDAGraph on ROCm) and replayed.  Captured: all five CSR launchers, one after the other; nothing runs at capture
    time; one replay computes all five y."""
    import ctypes as C
    torch = pytest.importorskip("torch")
    M, N, IRP, JA, AS = CASES["irregular"]
    x_host = _x(np.random.default_rng(41), N)
    y_ref = oracle.csr_serial(IRP, JA, AS, x_host)
    dmat = api.spMatCpyCSR(api.HostCSR(M, N, IRP, JA, AS))
    names = ["hipSpMVRowsCSR", "hipSpMVWarpPerRowCSR", "hipSpMVTilesCSR", "hipSpMVStripesCSR", "hipSpMVRowsSELL"]
    cfg = api.CONFIG()
    stream = torch.cuda.Stream()
    try:
        with torch.cuda.stream(stream):
            x = torch.from_numpy(x_host).cuda()
            ys = [torch.full((M,), float("nan"), dtype=torch.float64, device="cuda") for _ in names]
            api.lib.spmvHipSetStream(C.c_void_p(stream.cuda_stream))
            api.lib.spmvHipSetSync(0)
            for n, y in zip(names, ys):                      # first calls build the private formats: not capturable
                assert api.SPMV_LAUNCHERS[n](C.byref(dmat.handle), x.data_ptr(), cfg, y.data_ptr()) == 0
            torch.cuda.synchronize()
            for y in ys:
                y.fill_(float("nan"))
            graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(graph, stream=stream):
                for n, y in zip(names, ys):
                    assert api.SPMV_LAUNCHERS[n](C.byref(dmat.handle), x.data_ptr(), cfg, y.data_ptr()) == 0
            torch.cuda.synchronize()
            assert all(bool(torch.isnan(y).all()) for y in ys)            # capture ran nothing
            for _ in range(2):
                graph.replay()
            torch.cuda.synchronize()
            for n, y in zip(names, ys):
                yh = y.cpu().numpy()
                assert not np.isnan(yh).any(), n
                assert tight_error(IRP, JA, AS, x_host, y_ref, yh) <= TIGHT, n
            assert np.array_equal(ys[0].cpu().numpy(), y_ref)
    finally:
        api.lib.spmvHipSetStream(None)
        api.lib.spmvHipSetSync(1)
        dmat.free()


def _fuzz_case(seed):
    """Seeded random shape: 1..4000 rows, 1..200 000 columns, row lengths from a mix of laws (many empty rows, a
    geometric body, a few rows as long as the matrix allows), sorted distinct columns."""
    rng = np.random.default_rng(seed)
    M = int(rng.choice([1, 2, 63, 64, 65, int(rng.integers(1, 4000))]))
    N = int(rng.choice([1, 2, 17, 4096, int(rng.integers(1, 200_000))]))
    law = seed % 4
    if law == 0:
        lens = rng.geometric(0.15, size=M) - 1
    elif law == 1:
        lens = np.where(rng.random(M) < 0.6, 0, rng.integers(0, 50, size=M))
    elif law == 2:
        lens = np.full(M, int(rng.integers(0, 70)))
    else:
        lens = (rng.pareto(1.2, size=M) * 3).astype(np.int64)
    lens = np.minimum(lens, N).astype(np.int64)
    for _ in range(3):                                     # a few rows as long as the matrix allows (capped)
        lens[int(rng.integers(0, M))] = min(N, int(rng.integers(1, 6000)))
    return (M, N) + random_csr(rng, M, N, lens)


@pytest.mark.parametrize("seed", list(range(24)))
def test_fuzz_random_shapes(api, oracle, seed):
    """Every CSR launcher, and the ELL launchers when the longest row is short enough for a padded copy, on seeded random
    shapes (sizes that are no multiple of anything, one row, one column, mostly empty, uniform, heavy-tailed)."""
    M, N, IRP, JA, AS = _fuzz_case(seed)
    x = _x(np.random.default_rng(1000 + seed), N)
    y_ref = oracle.csr_serial(IRP, JA, AS, x)
    dmat = api.spMatCpyCSR(api.HostCSR(M, N, IRP, JA, AS))
    for launcher in ("hipSpMVRowsCSR", "hipSpMVWarpPerRowCSR", "hipSpMVTilesCSR", "hipSpMVStripesCSR", "hipSpMVRowsSELL", "hipSpMVAutoCSR"):
        if launcher in ("hipSpMVTilesCSR", "hipSpMVStripesCSR") and JA.size == 0:
            continue                                      # these formats need at least one entry (refused with a message)
        y = _run(api, launcher, dmat, x, M)
        assert not np.isnan(y).any(), (seed, launcher)
        assert np.max(np.abs(y - y_ref), initial=0.0) <= GATE, (seed, launcher)
        if launcher == "hipSpMVRowsCSR":
            assert np.array_equal(y, y_ref), (seed, launcher)
        else:
            assert tight_error(IRP, JA, AS, x, y_ref, y) <= TIGHT, (seed, launcher)
    if JA.size:
        # the serial-order forms of the two format kernels on the same odd shapes: the oracle's bits, every one
        for build, launcher, forms in ((api.build_tiles, "hipSpMVTilesCSR", (1,)), (api.build_stripes, "hipSpMVStripesCSR", (1, 2))):
            for form in forms:
                build(dmat, deterministic=form)
                y = _run(api, launcher, dmat, x, M)
                assert np.array_equal(y, y_ref), (seed, launcher, form)
    dmat.free()
    if JA.size and int(np.diff(IRP.astype(np.int64)).max()) * M <= 4_000_000:
        for rowlens in (True, False):
            ell = api.HostCSR(M, N, IRP, JA, AS).to_ell(with_row_lens=rowlens)
            api.lib.spmvHipSetEllRowLens(1 if rowlens else 0)
            for launcher, transposed in (("hipSpMVRowsELL", True), ("hipSpMVRowsELLNNTransposed", False),
                                         ("hipSpMVWarpsPerRowELLNTrasposed", False)):
                d = api.spMatCpyELL(ell.transpose() if transposed else ell)
                y = _run(api, launcher, d, x, M)
                d.free()
                assert not np.isnan(y).any(), (seed, launcher, rowlens)
                assert tight_error(IRP, JA, AS, x, y_ref, y) <= TIGHT, (seed, launcher, rowlens)
        api.lib.spmvHipSetEllRowLens(1)


@pytest.mark.parametrize("spread,grid,rows,wide", [(0, 0, 0, -1), (1024, 0, 0, -1), (1024, 3, 64, -1), (517, 5, 200, 1), (6, 1, 64, -1)])
def test_stripes_rotated_sweeps(api, oracle, spread, grid, rows, wide):
    """The stripes kernel starts every workgroup's sweep somewhere inside the bin and wraps around (spmvStripesOpts.spread:
    1/1024ths of the bin over which one XCD's workgroups are spread).  Every entry must still be added exactly once: no
    spread, the full bin, bins of one or two batches, persistent grids of 1, 3 and 5 workgroups (opts.grid) walking many
    bins each, and the 32-bit-column encoding forced on matrices that do not need it (opts.wide).  Options are arguments
    of the build: an explicit build replaces the format, out-of-range options are refused, a handle built afterwards
    without options gets the automatic format."""
    for name in ("irregular", "wide", "banded", "ones", "uniform32"):
        M, N, IRP, JA, AS = CASES[name]
        x = _x(np.random.default_rng(23), N)
        y_ref = oracle.csr_serial(IRP, JA, AS, x)
        dmat = api.spMatCpyCSR(api.HostCSR(M, N, IRP, JA, AS))
        api.build_stripes(dmat, rowsPerBin=rows, grid=grid, spread=spread, wide=wide)
        info = api.stripes_info(dmat)
        assert info.spread == spread and (not grid or info.grid <= grid) and (not rows or info.rowsPerBin <= rows)
        assert info.wide == (1 if wide == 1 else 0) and info.deterministic == 0
        y = _run(api, "hipSpMVStripesCSR", dmat, x, M)
        dmat.free()
        assert not np.isnan(y).any(), name
        assert np.max(np.abs(y - y_ref), initial=0.0) <= GATE, name
        assert tight_error(IRP, JA, AS, x, y_ref, y) <= TIGHT, name
    M, N, IRP, JA, AS = CASES["irregular"]
    dmat = api.spMatCpyCSR(api.HostCSR(M, N, IRP, JA, AS))
    for bad in (dict(rowsPerBin=20001), dict(grid=100000), dict(spread=1025), dict(spread=-2)):
        with pytest.raises(api.SpmvHipError):
            api.build_stripes(dmat, **bad)
    assert api.stripes_info(dmat).nBins == 0                 # a refused build leaves nothing behind
    _run(api, "hipSpMVStripesCSR", dmat, _x(np.random.default_rng(1), N), M)
    assert api.stripes_info(dmat).spread == 6 and api.stripes_info(dmat).rowsPerBin <= 20000
    dmat.free()


@pytest.mark.parametrize("name", list(CASES))
def test_stripes_deterministic_form_is_the_serial_order(api, oracle, name):
    """spmvStripesOpts.deterministic (1: every row is added by ONE wavefront in ascending column order; 2: one shared stream
    whose batches add in ticket order): y is (a) the same
    bits in every run, (b) the same bits whatever the bin layout (bins of 64, 700 or 20 000 rows; 1, 3 or all
    workgroups), (c) the same bits when the rows are computed as three separate row blocks (the shape of a 3-rank run)
    and (d) the bits of the serial oracle -- the columns of every row ascend in these matrices, as the reference's loader
    guarantees (src/lib/parser.c:195-202)."""
    M, N, IRP, JA, AS = CASES[name]
    if JA.size == 0:
        pytest.skip("the stripes format needs at least one entry")
    x = _x(np.random.default_rng(29), N)
    y_ref = oracle.csr_serial(IRP, JA, AS, x)
    dmat = api.spMatCpyCSR(api.HostCSR(M, N, IRP, JA, AS))
    ys = []
    for form in (1, 2):                                      # owner wavefronts / ordered tickets
        for rows, grid in ((0, 0), (64, 3), (700, 1), (0, 0)):
            api.build_stripes(dmat, rowsPerBin=rows, grid=grid, deterministic=form)
            assert api.stripes_info(dmat).deterministic == form
            for _ in range(3):
                ys.append(_run(api, "hipSpMVStripesCSR", dmat, x, M))
    dmat.free()
    for y in ys:
        assert np.array_equal(y, ys[0])
    assert np.array_equal(ys[0], y_ref)
    # three row blocks, as three ranks would hold them
    cuts = [0, M // 3, 2 * M // 3, M]
    parts = []
    for r0, r1 in zip(cuts[:-1], cuts[1:]):
        if r1 == r0:
            continue
        b0, b1 = int(IRP[r0]), int(IRP[r1])
        if b1 == b0:
            parts.append(np.zeros(r1 - r0))
            continue
        blk = api.spMatCpyCSR(api.HostCSR(r1 - r0, N, IRP[r0:r1 + 1] - IRP[r0], JA[b0:b1], AS[b0:b1]))
        api.build_stripes(blk, deterministic=1 + len(parts) % 2)
        parts.append(_run(api, "hipSpMVStripesCSR", blk, x, r1 - r0))
        blk.free()
    assert np.array_equal(np.concatenate(parts), ys[0])


@pytest.mark.parametrize("name", list(CASES))
def test_tiles_deterministic_form_is_the_serial_order(api, oracle, name):
    """spmvTilesOpts.deterministic: a bin is four sub-bins, each walked by ONE wavefront in bin-major order, so a row's
    products are added in ascending column order: y is the same bits in every run, for every bin height (automatic, 256,
    5000 rows), when the rows are computed as three separate row blocks (the shape of a 3-rank run), through the
    expand / reduce pair with bin ranges, and it is the bits of the serial oracle."""
    import ctypes as C
    M, N, IRP, JA, AS = CASES[name]
    if JA.size == 0:
        pytest.skip("the two-phase format needs at least one entry")
    x = _x(np.random.default_rng(37), N)
    y_ref = oracle.csr_serial(IRP, JA, AS, x)
    dmat = api.spMatCpyCSR(api.HostCSR(M, N, IRP, JA, AS))
    ys = []
    for rows in (0, 256, 5000, 0):
        api.build_tiles(dmat, rowsPerBin=rows, deterministic=True)
        info = api.tiles_info(dmat)
        assert info.deterministic == 1 and info.rowsPerBin % 4 == 0 and info.nBins == (M + info.rowsPerBin - 1) // info.rowsPerBin
        for _ in range(3):
            ys.append(_run(api, "hipSpMVTilesCSR", dmat, x, M))
    # the two phases as separate launches, phase 2 in three bin ranges
    nb = api.tiles_info(dmat).nBins
    dx, dy = api.DeviceVector(N).up(x), api.DeviceVector(M)
    dy.poison()
    assert api.lib.hipSpMVTilesExpand(C.byref(dmat.handle), dx.ptr) == 0
    for b0, b1 in ((0, nb // 3), (nb // 3, nb // 3), (nb // 3, nb)):
        assert api.lib.hipSpMVTilesReduce(C.byref(dmat.handle), b0, b1, dy.ptr, 0, None) == 0
    ys.append(dy.down())
    with pytest.raises(api.SpmvHipError):
        api.build_tiles(dmat, taper=True, deterministic=True)
    dmat.free()
    for y in ys:
        assert np.array_equal(y, ys[0])
    assert np.array_equal(ys[0], y_ref)
    cuts = [0, M // 3, 2 * M // 3, M]
    parts = []
    for r0, r1 in zip(cuts[:-1], cuts[1:]):
        if r1 == r0:
            continue
        b0, b1 = int(IRP[r0]), int(IRP[r1])
        if b1 == b0:
            parts.append(np.zeros(r1 - r0))
            continue
        blk = api.spMatCpyCSR(api.HostCSR(r1 - r0, N, IRP[r0:r1 + 1] - IRP[r0], JA[b0:b1], AS[b0:b1]))
        api.build_tiles(blk, deterministic=True)
        parts.append(_run(api, "hipSpMVTilesCSR", blk, x, r1 - r0))
        blk.free()
    assert np.array_equal(np.concatenate(parts), ys[0])


@pytest.mark.parametrize("key", ["c3", "c5"])
def test_deterministic_forms_at_full_size(api, oracle, key):
    """BASELINE's full sizes through the deterministic forms of the two fast kernels: five runs give the same bits, the
    windows (head, middle, tail, heaviest row) are the serial oracle's bits, and the two kernels agree with each other on
    every row (both sum in ascending column order)."""
    from spmv_openmp_cuda_amd import synth
    w = synth.WORKLOADS[key]
    lens = synth.row_lengths(w)
    irp = synth.prefix(lens)
    dm = synth.device_csr(w, irp, 0, w.N)
    x = synth.make_x(w.N, w.cfg)
    dx, dy = api.DeviceVector(w.N).up(x), api.DeviceVector(w.N)
    S = 100_000
    heavy = int(np.argmax(lens))
    ranges = [(0, S), (w.N // 2, w.N // 2 + S), (w.N - S, w.N), (max(0, heavy - 10), min(w.N, heavy + 10))]
    refs = []
    for r0, r1 in ranges:
        ja, as_ = oracle.synth_fill(w.N, r0, irp[r0:r1 + 1], synth.SEED_STRUCT + w.cfg, synth.SEED_VAL + w.cfg, w.band)
        refs.append(oracle.csr_serial_dev((irp[r0:r1 + 1] - irp[r0]).astype(np.uint32), ja, as_, x))
    results = {}
    for launcher, build, form in (("hipSpMVTilesCSR", api.build_tiles, 1),) + ((("hipSpMVStripesCSR", api.build_stripes, 1),
                                                                                 ("hipSpMVStripesCSR", api.build_stripes, 2)) if key == "c3" else ()):
        build(dm, deterministic=form)
        first = None
        for _ in range(5):
            dy.poison()
            api.spmv(launcher, dm, dx, dy)
            y = dy.down()
            if first is None:
                first = y
                assert not np.isnan(y).any()
                for (r0, r1), yr in zip(ranges, refs):
                    assert np.array_equal(y[r0:r1], yr), (launcher, r0)
            else:
                assert np.array_equal(y, first), launcher
        if launcher in results:
            assert np.array_equal(results[launcher], first)
        results[launcher] = first
    if len(results) == 2:
        assert np.array_equal(results["hipSpMVTilesCSR"], results["hipSpMVStripesCSR"])
    dm.free()


def test_tiles_build_options(api, oracle):
    """spmvHipBuildTilesOpt: options are arguments of the build (no process state), an explicit build replaces the format,
    out-of-range options are refused, and spmvHipTilesInfo reports what was built."""
    import ctypes as C
    M, N, IRP, JA, AS = CASES["wide"]
    x = _x(np.random.default_rng(19), N)
    y_ref = oracle.csr_serial(IRP, JA, AS, x)
    dmat = api.spMatCpyCSR(api.HostCSR(M, N, IRP, JA, AS))
    assert api.tiles_info(dmat).nBins == 0                       # nothing built yet
    for rows, chunk, nt in ((0, 0, -1), (128, 4096, 0), (4096, 65536, 1), (0, 0, -1)):
        api.build_tiles(dmat, rowsPerBin=rows, chunk=chunk, ntStore=nt)
        info = api.tiles_info(dmat)
        assert info.nBins >= 1 and info.buildMs > 0 and info.bytes > 0 and info.deterministic == 0
        assert 0 < info.tempBytes <= 13 * JA.size + 24 * info.nSlices * info.nBins + (1 << 20)      # 12 B per entry + the tile tables
        if rows:
            assert info.rowsPerBin == rows and info.nBins == (M + rows - 1) // rows
        if chunk:
            assert info.chunk == chunk
        if nt >= 0:
            assert info.ntStore == nt
        y = _run(api, "hipSpMVTilesCSR", dmat, x, M)
        assert not np.isnan(y).any() and tight_error(IRP, JA, AS, x, y_ref, y) <= TIGHT
    for bad in (dict(rowsPerBin=63), dict(rowsPerBin=20001), dict(chunk=100)):
        with pytest.raises(api.SpmvHipError):
            api.build_tiles(dmat, **bad)
    # a second matrix built afterwards is NOT affected by the options of the first (there is no global toggle any more)
    d2 = api.spMatCpyCSR(api.HostCSR(M, N, IRP, JA, AS))
    _run(api, "hipSpMVTilesCSR", d2, x, M)
    auto = api.tiles_info(d2)
    api.build_tiles(dmat, rowsPerBin=128)
    d3 = api.spMatCpyCSR(api.HostCSR(M, N, IRP, JA, AS))
    _run(api, "hipSpMVTilesCSR", d3, x, M)
    assert api.tiles_info(d3).rowsPerBin == auto.rowsPerBin and api.tiles_info(d3).nBins == auto.nBins
    for d in (dmat, d2, d3):
        d.free()


def test_device_memory_comes_back(api, oracle):
    """Handle lifecycle on the device: upload -> both selections (which build, measure and discard formats) -> every format
    built explicitly in both of its forms -> SELL, ELL conversions -> host-pointer wrappers (cached device copies) -> free,
    drop the cache, finalize.  After the first round has loaded the code objects, free device memory must come back every
    round (the runtime keeps its own pools in 4 MiB pieces, so the number wobbles by a few of those; an array of the matrix
    left behind per round -- 3 MB at 2 B per entry, 20 MB for a copy of the CSR -- adds up past the tolerance)."""
    import ctypes as C
    import torch
    rng = np.random.default_rng(2026)
    M = N = 200_000
    IRP, JA, AS = random_csr(rng, M, N, rng.integers(0, 17, size=M))       # ~1.6 M entries: above the selections' threshold
    x = _x(rng, N)
    y_ref = oracle.csr_serial(IRP, JA, AS, x)
    vp = C.c_void_p
    free = []
    for round_ in range(7):
        host = api.HostCSR(M, N, IRP, JA, AS)
        d = api.spMatCpyCSR(host)
        assert np.array_equal(_run(api, "hipSpMVRowsCSR", d, x, M), y_ref)
        assert tight_error(IRP, JA, AS, x, y_ref, _run(api, "hipSpMVWarpPerRowCSR", d, x, M)) <= TIGHT
        for det in (False, True):
            api.build_tiles(d, deterministic=det)
            assert np.max(np.abs(_run(api, "hipSpMVTilesCSR", d, x, M) - y_ref)) <= GATE
        for mode in (0, 1, 2):
            api.build_stripes(d, deterministic=mode)
            assert np.max(np.abs(_run(api, "hipSpMVStripesCSR", d, x, M) - y_ref)) <= GATE
        assert np.max(np.abs(_run(api, "hipSpMVRowsSELL", d, x, M) - y_ref)) <= GATE
        for transposed in (False, True):
            e = api.csr_to_ell_device(d, transposed)
            assert np.array_equal(_run(api, "hipSpMVRowsELL" if transposed else "hipSpMVRowsELLNNTransposed", e, x, M), y_ref + 0.0)
            e.free()
        y = np.empty(M)
        assert api.lib.spmvHipRowsCSR(C.byref(host.struct), x.ctypes.data_as(vp), None, y.ctypes.data_as(vp)) == 0
        assert np.array_equal(y, y_ref)
        d.free()
        api.spmvHipFinalize()                                                # cache, workspace, events
        api.spmvHipInit(0)
        torch.cuda.synchronize()
        free.append(torch.cuda.mem_get_info()[0])
    assert min(free[2:]) >= free[1] - (8 << 20) and free[-1] >= free[1] - (8 << 20), free


@pytest.mark.parametrize("name", ["irregular", "ones", "uniform32", "banded", "wide"])
def test_non_finite_x_stays_in_its_rows(api, oracle, name):
    """x[0] = NaN, one +inf and one -inf further on.  A row is NaN / +inf / -inf exactly when the serial oracle says so
    (that class does not depend on the order of the sum) and every other row keeps its usual bound -- in particular no
    kernel may multiply a padding or filler entry with x[0] (the reference's CSR kernels have none; its ELL kernels do
    read {0.0, column 0} padding unless the row lengths stop them, which is why ELL is taken with row lengths here)."""
    M, N, IRP, JA, AS = CASES[name]
    x = _x(np.random.default_rng(70), N)
    x[0], x[N // 2], x[N - 1] = np.nan, np.inf, -np.inf
    with np.errstate(invalid="ignore"):
        y_ref = oracle.csr_serial(IRP, JA, AS, x)
    finite = np.isfinite(y_ref)
    assert (~finite).any() and finite.any()
    x0 = np.where(np.isfinite(x), x, 0.0)                               # the finite rows see only finite x: usual bounds
    y0_ref = oracle.csr_serial(IRP, JA, AS, x0)
    assert np.array_equal(y_ref[finite], y0_ref[finite])

    def check(y, exact, who):
        assert np.array_equal(np.isnan(y), np.isnan(y_ref)), who
        inf = np.isinf(y_ref)
        assert np.array_equal(y[inf], y_ref[inf]), who
        if exact:
            assert np.array_equal(y[finite], y_ref[finite]), who
        else:
            yz, rz = np.where(finite, y, 0.0), np.where(finite, y0_ref, 0.0)
            assert np.max(np.abs(yz - rz), initial=0.0) <= GATE and tight_error(IRP, JA, AS, x0, rz, yz) <= TIGHT, who

    host = api.HostCSR(M, N, IRP, JA, AS)
    d = api.spMatCpyCSR(host)
    for launcher, variant, exact in (("hipSpMVRowsCSR", 0, True), ("hipSpMVRowsCSR", 1, True), ("hipSpMVWarpPerRowCSR", 0, False),
                                     ("hipSpMVWarpPerRowCSR", 1, False), ("hipSpMVRowsSELL", -1, False)):
        if variant >= 0:
            api.set_variant(launcher, variant)
        check(_run(api, launcher, d, x, M), exact, (launcher, variant))
    for det in (False, True):
        api.build_tiles(d, deterministic=det)
        check(_run(api, "hipSpMVTilesCSR", d, x, M), det, ("tiles", det))
    for mode in (0, 1, 2):
        api.build_stripes(d, deterministic=mode)
        check(_run(api, "hipSpMVStripesCSR", d, x, M), mode != 0, ("stripes", mode))
    d.free()
    ell = host.to_ell(with_row_lens=True)
    api.lib.spmvHipSetEllRowLens(1)
    for launcher, mat, exact in (("hipSpMVRowsELL", ell.transpose(), True), ("hipSpMVRowsELLNNTransposed", ell, True),
                                 ("hipSpMVWarpsPerRowELLNTrasposed", ell, False)):
        de = api.spMatCpyELL(mat)
        check(_run(api, launcher, de, x, M), exact, launcher)
        de.free()


@pytest.mark.parametrize("name", ["irregular", "uniform32", "wide"])
def test_subnormal_products_and_sums(api, oracle, name):
    """Values scaled so that every product and every row sum is subnormal (|a x| ~ 1e-310): the serial-order kernels stay
    bit-identical to the oracle -- the fp64 units and the LDS adds (`ds_add_f64`, which the deterministic kernels sum
    with) keep subnormals -- and the others differ by a few units of the smallest subnormal."""
    M, N, IRP, JA, AS = CASES[name]
    AS = AS * 1e-305
    x = _x(np.random.default_rng(71), N)
    y_ref = oracle.csr_serial(IRP, JA, AS, x)
    tiny = np.finfo(np.float64).tiny
    assert np.max(np.abs(y_ref)) < tiny and np.count_nonzero(y_ref) > M // 2
    host = api.HostCSR(M, N, IRP, JA, AS)
    d = api.spMatCpyCSR(host)
    ulp = 5e-324
    rowlen = np.diff(IRP.astype(np.int64))

    def check(y, exact, who):
        if exact:
            assert np.array_equal(y, y_ref), who
        else:                                                            # every product exact in its last bit or rounded by <= 1 ulp
            assert np.all(np.abs(y - y_ref) <= ulp * (rowlen + 1)), who
    for launcher, variant, exact in (("hipSpMVRowsCSR", 0, True), ("hipSpMVRowsCSR", 1, True), ("hipSpMVWarpPerRowCSR", 0, False),
                                     ("hipSpMVWarpPerRowCSR", 1, False)):
        api.set_variant(launcher, variant)
        check(_run(api, launcher, d, x, M), exact, (launcher, variant))
    for det in (False, True):
        api.build_tiles(d, deterministic=det)
        check(_run(api, "hipSpMVTilesCSR", d, x, M), det, ("tiles", det))
    for mode in (0, 1, 2):
        api.build_stripes(d, deterministic=mode)
        check(_run(api, "hipSpMVStripesCSR", d, x, M), mode != 0, ("stripes", mode))
    d.free()
    ell = host.to_ell(with_row_lens=True)
    for launcher, mat in (("hipSpMVRowsELL", ell.transpose()), ("hipSpMVRowsELLNNTransposed", ell)):
        de = api.spMatCpyELL(mat)
        assert np.array_equal(_run(api, launcher, de, x, M), y_ref + 0.0), launcher
        de.free()


def test_under_device_memory_pressure(api, oracle, capfd):
    """All but 16 MiB of the device taken by someone else: the reference's two names still answer (their selections skip the
    formats that cannot be built and fall back to the kernel that needs no memory of its own -- serial order bit-identical
    as ever), the launchers that NEED a format refuse loudly, and a refused build leaves nothing behind: no device memory,
    and no sticky HIP error for the next, unrelated call to trip over."""
    import ctypes as C
    import torch
    rng = np.random.default_rng(5)
    M = N = 400_000
    IRP, JA, AS = random_csr(rng, M, N, rng.integers(0, 17, size=M))       # 3.2 M entries: a format needs 40-50 MB
    x = _x(rng, N)
    y_ref = oracle.csr_serial(IRP, JA, AS, x)
    host = api.HostCSR(M, N, IRP, JA, AS)
    d = api.spMatCpyCSR(host)
    dx, dy = api.DeviceVector(N).up(x), api.DeviceVector(M)
    torch.cuda.synchronize()
    free0 = torch.cuda.mem_get_info()[0]
    hog = C.c_void_p()
    assert api.lib.spmvHipMalloc(C.byref(hog), free0 - (16 << 20)) == 0
    try:
        for launcher, exact in (("hipSpMVWarpPerRowCSR", False), ("hipSpMVRowsCSR", True)):
            for _ in range(2):                                           # the selection, then the remembered choice
                dy.poison()
                api.spmv(launcher, d, dx, dy)
                y = dy.down()
                if exact:
                    assert np.array_equal(y, y_ref)
                else:
                    assert tight_error(IRP, JA, AS, x, y_ref, y) <= TIGHT
        assert api.lib.spmvHipAutoChoice(C.byref(d.handle), None) == b"hipSpMVWarpPerRowCSR"
        assert api.lib.spmvHipAutoChoiceRows(C.byref(d.handle), None) == b"hipSpMVRowsCSR"
        refused = 0
        for launcher in ("hipSpMVTilesCSR", "hipSpMVStripesCSR", "hipSpMVRowsSELL"):
            dy.poison()
            try:
                api.spmv(launcher, d, dx, dy)                            # (the runtime's own pools may still hold room for one of them)
                assert np.max(np.abs(dy.down() - y_ref)) <= GATE, launcher
            except api.SpmvHipError:
                refused += 1
            dy.poison()                                                  # a launch + hipGetLastError() inside: must not see the failed hipMalloc
            assert np.isnan(dy.down()).all()
        assert refused >= 1 and "allocation" in capfd.readouterr().err
    finally:
        api.lib.spmvHipFree(hog)
    for launcher in ("hipSpMVTilesCSR", "hipSpMVStripesCSR", "hipSpMVRowsSELL"):       # with room again the same handle builds them
        dy.poison()
        api.spmv(launcher, d, dx, dy)
        assert np.max(np.abs(dy.down() - y_ref)) <= GATE
    d.free(); dx.free(); dy.free()


def test_two_phase_products_are_handed_from_stream_to_stream(api, oracle):
    """Two matrices served by the two-phase kernel share the device's ONE product workspace.  Driven alternately from two
    non-blocking streams without any synchronisation by the caller, B's phase 1 must wait for A's phase 2 (an event recorded
    behind every phase 2, tiles.hip prodMark / prodHandover) -- every y of every round is checked.  Then the first stream is
    DESTROYED and the next call comes from the other one: the library must not touch the dead handle (the host-sanitizer
    run of round 3 caught exactly that), and a stream created afterwards -- possibly at the same address -- works too."""
    import ctypes as C
    hip = C.CDLL("libamdhip64.so")
    hip.hipStreamCreateWithFlags.argtypes = [C.POINTER(C.c_void_p), C.c_uint]
    hip.hipStreamDestroy.argtypes = [C.c_void_p]
    hip.hipStreamSynchronize.argtypes = [C.c_void_p]
    rng = np.random.default_rng(909)
    mats = []
    for M, N, per in ((300_000, 300_000, 24), (200_000, 500_000, 32)):       # 7.2 M and 6.4 M entries: phases of ~50-100 us
        JA = np.sort(rng.integers(0, N, size=(M, per)), axis=1).astype(np.uint64).ravel()
        IRP = (np.arange(M + 1, dtype=np.uint64) * per)
        AS = rng.uniform(-1, 1, size=JA.size)
        x = _x(rng, N)
        d = api.spMatCpyCSR(api.HostCSR(M, N, IRP, JA, AS))
        api.build_tiles(d)
        mats.append((M, IRP, JA, AS, x, oracle.csr_serial(IRP, JA, AS, x), d, api.DeviceVector(N).up(x), [api.DeviceVector(M) for _ in range(6)]))
    streams = [C.c_void_p(), C.c_void_p()]
    for s in streams:
        assert hip.hipStreamCreateWithFlags(C.byref(s), 1) == 0             # hipStreamNonBlocking
    cfg = api.CONFIG()
    tiles = api.SPMV_LAUNCHERS["hipSpMVTilesCSR"]

    def enqueue(k, stream, slot):
        M, IRP, JA, AS, x, y_ref, d, dx, dys = mats[k]
        api.lib.spmvHipSetStream(stream)
        assert tiles(C.byref(d.handle), dx.ptr, cfg, dys[slot].ptr) == 0

    def check(k, slots):
        M, IRP, JA, AS, x, y_ref, d, dx, dys = mats[k]
        for slot in slots:
            y = dys[slot].down()
            assert not np.isnan(y).any() and tight_error(IRP, JA, AS, x, y_ref, y) <= TIGHT, (k, slot)
            dys[slot].poison()
    try:
        for k in (0, 1):
            for dy in mats[k][8]:
                dy.poison()
        api.lib.spmvHipSetSync(0)
        for round_ in range(5):
            for slot in range(6):                                        # A on stream 0, B on stream 1, back to back, nobody waits
                enqueue(0, streams[0], slot)
                enqueue(1, streams[1], slot)
            assert api.lib.spmvHipDeviceSynchronize() == 0
            check(0, range(6)); check(1, range(6))
        enqueue(0, streams[0], 0)
        assert hip.hipStreamSynchronize(streams[0]) == 0 and hip.hipStreamDestroy(streams[0]) == 0
        dead, streams[0] = streams[0].value, C.c_void_p()
        enqueue(1, streams[1], 0)                                        # hand-over from a stream that no longer exists
        enqueue(0, streams[1], 1)
        assert api.lib.spmvHipDeviceSynchronize() == 0
        check(0, (0, 1)); check(1, (0,))
        assert hip.hipStreamCreateWithFlags(C.byref(streams[0]), 1) == 0    # often the address of the one just destroyed
        print("new stream at the dead one's address:", streams[0].value == dead)
        enqueue(0, streams[0], 2)
        enqueue(1, streams[1], 2)
        assert api.lib.spmvHipDeviceSynchronize() == 0
        check(0, (2,)); check(1, (2,))
    finally:
        api.lib.spmvHipSetStream(None)
        api.lib.spmvHipSetSync(1)
        api.lib.spmvHipDeviceSynchronize()
        for s in streams:
            if s:
                hip.hipStreamDestroy(s)
        for m in mats:
            m[6].free(); m[7].free()
            for dy in m[8]:
                dy.free()


def test_null_vectors_are_refused_on_the_host(api, capfd):
    """A NULL x or y would be dereferenced by every lane of a kernel (a GPU page fault); every launcher refuses it before
    anything is enqueued."""
    import ctypes as C
    M, N, IRP, JA, AS = CASES["cage4-shaped"]
    host = api.HostCSR(M, N, IRP, JA, AS)
    d = api.spMatCpyCSR(host)
    ell = host.to_ell()
    de, det = api.spMatCpyELL(ell), api.spMatCpyELL(ell.transpose())
    dx, dy = api.DeviceVector(N), api.DeviceVector(M)
    cfg = api.CONFIG()
    cases = [(n, d) for n in ("hipSpMVRowsCSR", "hipSpMVWarpPerRowCSR", "hipSpMVTilesCSR", "hipSpMVStripesCSR", "hipSpMVRowsSELL", "hipSpMVAutoCSR")]
    cases += [("hipSpMVRowsELL", det), ("hipSpMVRowsELLNNTransposed", de), ("hipSpMVWarpsPerRowELLNTrasposed", de)]
    for variant in (0, 1, 2):
        api.set_variant("hipSpMVRowsCSR", variant)
        api.set_variant("hipSpMVWarpPerRowCSR", variant)
        for name, mat in cases:
            fn = api.SPMV_LAUNCHERS[name]
            assert fn(C.byref(mat.handle), None, cfg, dy.ptr) != 0, name
            assert fn(C.byref(mat.handle), dx.ptr, cfg, None) != 0, name
    assert "is NULL" in capfd.readouterr().err
    for stream_fn in (api.lib.spmvHipEnqueueAuto, api.lib.spmvHipEnqueueAutoRows):
        assert stream_fn(C.byref(d.handle), None, dy.ptr, None) != 0
        assert stream_fn(C.byref(d.handle), dx.ptr, None, None) != 0
    d.free(); de.free(); det.free(); dx.free(); dy.free()


@pytest.mark.parametrize("name", ["irregular", "uniform32", "banded", "wide", "hyper-sparse", "ones"])
@pytest.mark.parametrize("value", [1.0, -2.5])
def test_matrices_whose_values_are_all_the_same(api, oracle, name, value):
    """MatrixMarket `pattern` files are loaded as all 1.0 (parser.c:59-61; the graphs of the reference's report are such
    files).  The upload recognises "every stored value is the same double"; the LDS-stream, stripes and two-phase kernels
    then keep the value in a register and stream no values.  y must not change by a bit: the serial-order kernels stay
    identical to the oracle, every kernel equals what it computes with the recognition switched off, and one differing
    value anywhere switches it off by itself."""
    import ctypes as C
    M, N, IRP, JA, AS = CASES[name]
    AS = np.full(AS.size, value)
    x = _x(np.random.default_rng(72), N)
    y_ref = oracle.csr_serial(IRP, JA, AS, x)
    c = C.c_double(0.0)

    def run_all(d):
        out = {}
        for launcher, variant in (("hipSpMVRowsCSR", 0), ("hipSpMVRowsCSR", 1), ("hipSpMVWarpPerRowCSR", 0), ("hipSpMVWarpPerRowCSR", 1)):
            api.set_variant(launcher, variant)
            out[(launcher, variant)] = _run(api, launcher, d, x, M)
        for det in (False, True):
            api.build_tiles(d, deterministic=det)
            out[("tiles", det)] = _run(api, "hipSpMVTilesCSR", d, x, M)
        for mode in (0, 1, 2):
            api.build_stripes(d, deterministic=mode)
            out[("stripes", mode)] = _run(api, "hipSpMVStripesCSR", d, x, M)
        api.set_variant("hipSpMVRowsCSR", 2)
        api.set_variant("hipSpMVWarpPerRowCSR", 2)
        out["CUDA_CSR_ROWS"] = _run(api, "hipSpMVRowsCSR", d, x, M)
        out["CUDA_CSR_ROWS_WARP"] = _run(api, "hipSpMVWarpPerRowCSR", d, x, M)
        return out
    exact = {("hipSpMVRowsCSR", 0), ("hipSpMVRowsCSR", 1), ("tiles", True), ("stripes", 1), ("stripes", 2), "CUDA_CSR_ROWS"}
    arrival = {("tiles", False), ("stripes", 0), "CUDA_CSR_ROWS_WARP"}          # sums in arrival order: equal to rounding, run to run
    d = api.spMatCpyCSR(api.HostCSR(M, N, IRP, JA, AS))
    assert api.lib.spmvHipUnitValue(C.byref(d.handle), C.byref(c)) == 1 and c.value == value
    sb0 = None
    fast = run_all(d)
    if JA.size >= 1:
        sb0 = api.lib.spmvHipStripesBytes(C.byref(d.handle))
    d.free()
    try:
        api.lib.spmvHipSetUnitValues(0)
        d = api.spMatCpyCSR(api.HostCSR(M, N, IRP, JA, AS))
        assert api.lib.spmvHipUnitValue(C.byref(d.handle), None) == 0
        plain = run_all(d)
        sb1 = api.lib.spmvHipStripesBytes(C.byref(d.handle))
        d.free()
    finally:
        api.lib.spmvHipSetUnitValues(1)
    assert sb0 < sb1                                                            # no value array in the stripes format
    for k, y in fast.items():
        assert not np.isnan(y).any(), k
        if k in exact:
            assert np.array_equal(y, y_ref), k
        else:
            assert tight_error(IRP, JA, AS, x, y_ref, y) <= TIGHT, k
        if k not in arrival:
            assert np.array_equal(y, plain[k]), k
    small = M * int(np.diff(IRP.astype(np.int64)).max(initial=0)) <= 20_000_000     # "wide": one 30 000-entry row would make 1.2 G ELL cells
    if small:
        # ELL with row lengths: the real cells are all `value`, the padding cells {0.0, column 0} are never touched
        host = api.HostCSR(M, N, IRP, JA, AS)
        ell = host.to_ell(with_row_lens=True)
        api.lib.spmvHipSetEllRowLens(1)
        dcsr = api.spMatCpyCSR(host)
        mats = [("hipSpMVRowsELL", api.spMatCpyELL(ell.transpose()), True), ("hipSpMVRowsELLNNTransposed", api.spMatCpyELL(ell), True),
                ("hipSpMVWarpsPerRowELLNTrasposed", api.spMatCpyELL(ell), False), ("hipSpMVRowsELL", api.csr_to_ell_device(dcsr, True), True),
                ("hipSpMVRowsELLNNTransposed", api.csr_to_ell_device(dcsr, False), True)]
        for launcher, de, bitwise in mats:
            assert api.lib.spmvHipUnitValue(C.byref(de.handle), C.byref(c)) == (1 if JA.size else 0), launcher
            y = _run(api, launcher, de, x, M)
            if bitwise:
                assert np.array_equal(y, y_ref + 0.0), launcher
            else:
                assert tight_error(IRP, JA, AS, x, y_ref, y) <= TIGHT, launcher
            api.lib.spmvHipSetEllRowLens(0)                                         # all slots: padding is part of the sum, values are read
            y = _run(api, launcher, de, x, M)
            api.lib.spmvHipSetEllRowLens(1)
            assert np.max(np.abs(y - y_ref), initial=0.0) <= GATE, launcher
            de.free()
        dcsr.free()
        de = api.spMatCpyELL(host.to_ell(with_row_lens=False))                       # no row lengths uploaded: nothing to recognise
        assert api.lib.spmvHipUnitValue(C.byref(de.handle), None) == 0
        de.free()
    AS2 = AS.copy()
    AS2[AS2.size // 2] = np.nextafter(value, 10.0)                              # one value off by one ulp
    d = api.spMatCpyCSR(api.HostCSR(M, N, IRP, JA, AS2))
    assert api.lib.spmvHipUnitValue(C.byref(d.handle), None) == 0
    y2_ref = oracle.csr_serial(IRP, JA, AS2, x)
    assert np.array_equal(_run(api, "hipSpMVRowsCSR", d, x, M), y2_ref)
    d.free()
    if small:
        de = api.spMatCpyELL(api.HostCSR(M, N, IRP, JA, AS2).to_ell(with_row_lens=True).transpose())
        assert api.lib.spmvHipUnitValue(C.byref(de.handle), None) == 0
        assert np.array_equal(_run(api, "hipSpMVRowsELL", de, x, M), y2_ref + 0.0)
        de.free()

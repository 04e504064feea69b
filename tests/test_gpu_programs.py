"""GPU tests of the plain-C programs and of the single-process sharded path:
  * bin/SpMV_HIP.elf (CLI, reference argv/dump-file/stdout conventions) on the
    golden MatrixMarket fixtures, output compared with the reference CLI's own y;
  * tests/harness/test_SpMV_HIP.elf (the reference's check+timing harness
    restated): every HIP and OpenMP implementation must pass the 7e-4 gate;
  * spmvHipShardCSR/spmvHipSpMVSharded with nDev = 1, with and without RCCL."""
import ctypes as C
import os
import subprocess

import numpy as np
import pytest

from conftest import ROOT, random_csr
from test_oracle import GOLD, NAMES, load_golden

pytestmark = pytest.mark.gpu
CLI = os.path.join(ROOT, "spmv_openmp_cuda_amd", "bin", "SpMV_HIP.elf")
HARNESS = os.path.join(ROOT, "tests", "harness", "test_SpMV_HIP.elf")


@pytest.mark.parametrize("name", NAMES)
@pytest.mark.parametrize("mode", ["CUDA_CSR_ROWS", "CUDA_CSR_ROWS_WARP", "HIP_CSR_TILES", "HIP_CSR_STRIPES", "CUDA_CSR_AUTO", "HIP_SELL_ROWS", "CUDA_ELL_ROWS",
                                  "HIP_ELL_ROWS_NN_TRANSPOSED", "CUDA_ELL_ROWS_WARP_NN_TRANSPOSED"])
def test_cli_against_reference_cli_output(name, mode):
    g = load_golden(name)
    r = subprocess.run([CLI, os.path.join(GOLD, name + ".mtx"), os.path.join(GOLD, f"x_{name}.bin"), mode],
                       capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stderr
    assert r.stdout.strip().splitlines()[-1].startswith("cmode:")      # last line keeps the reference grammar
    y = np.fromfile("/tmp/outVectorDumpRaw")
    assert y.size == g["M"] and not np.isnan(y).any()
    assert np.max(np.abs(y - g["y_csr"]), initial=0) <= 7e-4
    if mode in ("CUDA_CSR_ROWS", "HIP_SELL_ROWS", "CUDA_ELL_ROWS", "HIP_ELL_ROWS_NN_TRANSPOSED"):
        assert np.max(np.abs(y - g["y_csr"]), initial=0) <= 1e-19       # ascending-j kernels: same bits up to simd re-association
    txt = np.loadtxt("/tmp/outVectorDump", ndmin=1)
    assert np.allclose(txt, y, rtol=1e-6, atol=0)


def test_cli_reads_gz(tmp_path):
    import gzip
    g = load_golden("sym6")
    src = tmp_path / "sym6.mtx.gz"
    src.write_bytes(gzip.compress(open(os.path.join(GOLD, "sym6.mtx"), "rb").read()))
    r = subprocess.run([CLI, str(src), os.path.join(GOLD, "x_sym6.bin"), "CUDA_CSR_ROWS"], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stderr
    assert np.max(np.abs(np.fromfile("/tmp/outVectorDumpRaw") - g["y_csr"])) <= 1e-19


def test_cli_sharded_env():
    """SPMV_NGPU=1 drives the spmvHipShardCSR path from the CLI."""
    g = load_golden("rand300")
    r = subprocess.run([CLI, os.path.join(GOLD, "rand300.mtx"), os.path.join(GOLD, "x_rand300.bin"), "CUDA_CSR_ROWS"],
                       capture_output=True, text=True, timeout=120, env=dict(os.environ, SPMV_NGPU="1"))
    assert r.returncode == 0 and "nGPU: 1" in r.stdout, r.stdout + r.stderr
    y = np.fromfile("/tmp/outVectorDumpRaw")
    assert np.max(np.abs(y - g["y_csr"])) <= 1e-19


def test_cli_at_scale_on_a_generated_matrix(tmp_path, oracle):
    """The plain-C path end to end on a matrix large enough for everything to matter: a 1 M-row road-network stand-in
    written as MatrixMarket text (csrc/host/structured.c), read by the CLI's loader (parallel parser), uploaded, run
    through the reference's mode names -- the two CSR names resolve through the library's selections inside the CLI's
    single call -- and the dumped y compared with sgemvSerial on the same file (bit for bit for the serial-order modes)."""
    from spmv_openmp_cuda_amd import api
    H = api.hostlib
    mtx = str(tmp_path / "road1m.mtx")
    Mv, NZv, mx = C.c_ulong(), C.c_ulong(), C.c_ulong()
    assert H.spmvSynthWriteMtx(mtx.encode(), 1, 1_000_000, 0, 0, 4242, C.byref(Mv), C.byref(NZv), C.byref(mx)) == 0
    m = H.MMtoCSR(mtx.encode())
    assert m
    c = m.contents
    M, nnz = int(c.M), int(c.NZ)
    assert nnz >= 1 << 18
    irp = np.ctypeslib.as_array(c.IRP, shape=(M + 1,)).copy()
    ja = np.ctypeslib.as_array(c.JA, shape=(nnz,)).copy()
    as_ = np.ctypeslib.as_array(c.AS, shape=(nnz,)).copy()
    H.freeSpmat(m)
    x = np.sin(np.random.default_rng(3).uniform(0, 7, M)) * 3e-5
    xfile = str(tmp_path / "x.bin")
    x.tofile(xfile)
    y_ref = oracle.csr_serial(irp, ja, as_, x)
    for mode, exact in (("CUDA_CSR_ROWS", True), ("CUDA_CSR_ROWS_WARP", False), ("CUDA_ELL_ROWS", True),
                        ("CUDA_ELL_ROWS_WARP_NN_TRANSPOSED", False), ("HIP_ELL_ROWS_NN_TRANSPOSED", True)):
        r = subprocess.run([CLI, mtx, xfile, mode], capture_output=True, text=True, timeout=300)
        assert r.returncode == 0, mode + r.stderr[-2000:]
        y = np.fromfile("/tmp/outVectorDumpRaw")
        assert y.size == M and not np.isnan(y).any(), mode
        assert np.max(np.abs(y - y_ref)) <= 7e-4, mode
        if exact:
            assert np.array_equal(y, y_ref + 0.0), mode
        assert "GFLOPS:" in r.stdout and r.stdout.strip().splitlines()[-1].startswith("cmode:")


def test_cli_rejects_bad_usage():
    mtx = os.path.join(GOLD, "cage4like.mtx")
    x = os.path.join(GOLD, "x_cage4like.bin")
    for argv in ([CLI], [CLI, mtx, x, "CUDA_CSR"], [CLI, mtx, x, "CSR_ROWS"], [CLI, "/nonexistent.mtx", x],
                 [CLI, mtx, os.path.join(GOLD, "x_sym6.bin")]):
        assert subprocess.run(argv, capture_output=True, timeout=60).returncode != 0


@pytest.mark.parametrize("name", ["cage4like", "rand300", "skew12x40"])
def test_harness_all_implementations_pass(name):
    r = subprocess.run([HARNESS, os.path.join(GOLD, name + ".mtx"), os.path.join(GOLD, f"x_{name}.bin")],
                       capture_output=True, text=True, timeout=300, env=dict(os.environ, OMP_NUM_THREADS="4"))
    assert r.returncode == 0, r.stdout + r.stderr
    out = r.stdout
    assert out.count("cudaBlockSize:") == 9 and out.count("threadNum:") == 8     # 6 CSR-upload (the last: hipSpMVAutoCSR) + 3 ELL launchers; 5 + 3 OpenMP variants like the reference
    assert "AVG_TIMES_ITERATION:25" in out and "MAX_ROW_NZ" in out and "omp sched gather:" in out


def test_harness_reports_the_kernel_behind_the_reference_names(tmp_path):
    """On a matrix large enough for the selection to measure (>= 2^18 entries) the harness shows which kernel the
    reference's names SpmvCUDA_CSRFuncs[0] (serial-order selection) and SpmvCUDA_CSRFuncs[SpmvCUDA_CSRFuncs_WarpPerRowIdx]
    (src/include/SpMV.h:130-134; the same one hipSpMVAutoCSR picks) resolved to, and every implementation still passes the gate."""
    rng = np.random.default_rng(41)
    M = N = 30_000
    IRP, JA, AS = random_csr(rng, M, N, np.full(M, 10))
    mtx = tmp_path / "u30k.mtx"
    with open(mtx, "w") as f:
        f.write("%%MatrixMarket matrix coordinate real general\n")
        f.write(f"{M} {N} {JA.size}\n")
        rows = np.repeat(np.arange(M), 10)
        np.savetxt(f, np.column_stack([rows + 1, JA.astype(np.int64) + 1, AS]), fmt="%d %d %.17g")
    r = subprocess.run([HARNESS, str(mtx), "RNDVECT", "CUDA_ONLY"], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    picks = [l for l in r.stdout.splitlines() if l.startswith("#auto CSR")]
    assert len(picks) == 3 and picks[0].startswith("#auto CSR 0\tpick:hipSpMV") and picks[1].startswith("#auto CSR 1\tpick:hipSpMV")
    assert picks[2].startswith("#auto CSR 5\tpick:hipSpMV") and picks[1].split("\t")[1] == picks[2].split("\t")[1]
    assert all("msStripes:0.0" not in p for p in picks)       # the candidates were really measured


def test_harness_thread_sweep():
    """DECREASE_THREAD_NUM (test/SpMV_test.cu:73-78): every OpenMP implementation measured with 3, 2 and 1 threads"""
    r = subprocess.run([HARNESS, os.path.join(GOLD, "rand300.mtx"), os.path.join(GOLD, "x_rand300.bin")],
                       capture_output=True, text=True, timeout=300, env=dict(os.environ, OMP_NUM_THREADS="3", DECREASE_THREAD_NUM="1"))
    assert r.returncode == 0, r.stdout + r.stderr
    assert r.stdout.count("threadNum:") == 8 * 3
    for t in (3, 2, 1):
        assert r.stdout.count(f"threadNum: {t}\t") == 8


def test_sharded_single_process_path(oracle):
    """spmvHipShardCSR / spmvHipShardCSRGroups / spmvHipSpMVSharded on one device: 1, 2 and 3 row groups (the gather of
    group g on its own stream while group g+1 is computed), with and without RCCL (a 1-rank communicator), mode 0 (the
    serial-order kernel: y bit-identical to the oracle) and mode 1 (the kernel hipSpMVWarpPerRowCSR runs: every block is
    large enough -- >= 2^18 entries -- for the per-block selection among the reduction-order kernels to really measure)."""
    from spmv_openmp_cuda_amd import api
    from conftest import tight_error
    api.spmvHipInit(0)
    rng = np.random.default_rng(5)
    M, N = 90_000, 70_000
    lens = rng.integers(0, 24, size=M)
    lens[77] = 3500
    lens[::11] = 0
    IRP, JA, AS = random_csr(rng, M, N, lens)
    assert JA.size >= 3 * (1 << 18)
    x = np.sin(rng.uniform(0, 7, N)) * 3e-5
    y_ref = oracle.csr_serial(IRP, JA, AS, x)
    host = api.HostCSR(M, N, IRP, JA, AS)
    try:
        for force, groups in (("0", 0), ("1", 1), ("1", 2), ("0", 3), ("1", 3)):
            os.environ["SPMV_SHARD_FORCE_RCCL"] = force
            h = C.c_void_p()
            if groups:
                assert api.lib.spmvHipShardCSRGroups(C.byref(host.struct), 1, groups, C.byref(h)) == 0
            else:
                assert api.lib.spmvHipShardCSR(C.byref(host.struct), 1, C.byref(h)) == 0
            for mode, exact in ((0, True), (1, False), (1, False), (0, True)):
                y = np.full(M, np.nan)
                ks, gs = C.c_double(), C.c_double()
                assert api.lib.spmvHipSpMVSharded(h, x.ctypes.data_as(C.c_void_p), mode, y.ctypes.data_as(C.c_void_p),
                                                  C.byref(ks), C.byref(gs)) == 0, (force, groups, mode)
                assert not np.isnan(y).any() and np.max(np.abs(y - y_ref)) <= 7e-4, (force, groups, mode)
                if exact:
                    assert np.array_equal(y, y_ref)
                else:
                    assert tight_error(IRP, JA, AS, x, y_ref, y) <= 1e-13
                assert ks.value > 0 and gs.value >= 0
            api.lib.spmvHipShardFree(h)
    finally:
        os.environ.pop("SPMV_SHARD_FORCE_RCCL", None)
    # asking for more devices than exist, or an absurd group count, fails loudly
    h = C.c_void_p()
    assert api.lib.spmvHipShardCSR(C.byref(host.struct), 64, C.byref(h)) != 0
    assert api.lib.spmvHipShardCSRGroups(C.byref(host.struct), 1, 1000, C.byref(h)) != 0
    api.spmvHipFinalize()


def test_bench_spawns_its_own_ranks():
    """`python bench.py --gpus 2` with NO external launcher in the environment: bench.py must start its ranks itself
    (fresh processes through torch.distributed.run), relay rank 0's ONE JSON line on stdout and its exit code.  Here as a
    shared-GPU rehearsal (both ranks on device 0, gloo as control plane) on a 5 % workload."""
    import json
    import sys
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT",
                                                              "TORCHELASTIC_RUN_ID", "GROUP_RANK", "LOCAL_WORLD_SIZE")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--rehearse-shared-gpu", "--scale", "0.05",
                        "--steps", "3", "--warmup", "1", "--exchange-budget", "20"], capture_output=True, text=True, timeout=900, env=env, cwd=ROOT)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    lines = [l for l in r.stdout.splitlines() if l.strip()]
    assert len(lines) == 1, r.stdout[-2000:]
    line = json.loads(lines[0])
    assert line["n_gpus"] == 2 and line["metric"] == "spmv_gflops" and line["value"] > 0
    assert line["parity"]["ok"] is True


def test_serial_order_y_is_the_same_bits_on_one_rank_and_on_three():
    """The whole y of `hipSpMVRowsCSR` in its default (serial-order) variant, hashed: the 1-rank bench and a 3-rank
    shared-GPU rehearsal (three processes, each computing a third of the rows with whatever serial-order kernel its own
    selection picked, y assembled through the exchange) must produce byte-identical vectors ("gathered y
    bitwise equal to the 1-GPU y" -- here for the fast kernels, not only for the LDS-stream one)."""
    import json
    import sys
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT",
                                                              "TORCHELASTIC_RUN_ID", "GROUP_RANK", "LOCAL_WORLD_SIZE")}
    common = ["--workload", "c3", "--scale", "0.05", "--steps", "2", "--warmup", "1", "--launcher", "hipSpMVRowsCSR", "--variant", "2",
              "--y-hash", "--no-cpu-baseline", "--no-extra"]
    hashes = []
    for extra in (["--gpus", "1"], ["--gpus", "3", "--rehearse-shared-gpu", "--exchange-budget", "20"]):
        r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + extra + common, capture_output=True, text=True, timeout=900,
                           env=env, cwd=ROOT)
        assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
        line = json.loads([l for l in r.stdout.splitlines() if l.strip()][-1])
        assert line["parity"]["ok"] is True and len(line["y_sha256"]) == 64
        hashes.append(line["y_sha256"])
    assert hashes[0] == hashes[1]

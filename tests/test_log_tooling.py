"""The harness keeps the reference's stdout grammar, and scripts/parse_harness_log.py turns it into the
reference tool's CSV columns (+ GFLOPS / GBps / rooflineFrac).  Runs the harness in OMP_ONLY mode (no GPU)."""
import importlib.util
import os
import subprocess

from conftest import ROOT

HARNESS = os.path.join(ROOT, "tests", "harness", "test_SpMV_HIP.elf")
GOLD = os.path.join(ROOT, "tests", "golden")


def _parser():
    spec = importlib.util.spec_from_file_location("parse_harness_log", os.path.join(ROOT, "scripts", "parse_harness_log.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def test_harness_log_grammar_and_csv():
    if not os.path.exists(HARNESS):
        subprocess.check_call(["make", "-C", ROOT, "harness"], stdout=subprocess.DEVNULL)
    r = subprocess.run([HARNESS, os.path.join(GOLD, "rand300.mtx"), os.path.join(GOLD, "x_rand300.bin"), "OMP_ONLY"],
                       capture_output=True, text=True, timeout=120, env=dict(os.environ, OMP_NUM_THREADS="2"))
    assert r.returncode == 0, r.stderr
    rows = _parser().parse(r.stdout.splitlines(True))
    assert [x["funcID"] for x in rows] == [f"OMP CSR {i}" for i in range(5)] + [f"OMP ELL {i}" for i in range(3)]
    for x in rows:
        assert (x["matRows"], x["matCols"], x["NNZ"], x["maxRowNNZ"], x["sampleSize"]) == (300, 300, 2106, 250, 25)
        assert x["threadNum"] == 2 and x["ompGrid"] == "8x8" and x["ompSchedKind"].startswith("OMP_SCHED_")
        assert x["timeAvg"] > 0 and abs(x["GFLOPS"] - 2 * 2106 / x["timeAvg"] * 1e-9) < 1e-9
        assert x["blockSize_x"] is None


def test_parser_reads_gpu_lines():
    log = ("#A.mtx\\nSpMV_OMP_test.c\\tAVG_TIMES_ITERATION:25\\tsparse matrix: 2597x2597-76367NNZ-62=MAX_ROW_NZ\\n"
           "omp sched gather:\\tkind: OMP_SCHED_DYNAMIC\\tomp chunkSize: 1\\tmonotonic: N\\n"
           "#auto CSR 0\\tpick:hipSpMVRowsCSR\\tmsStream:0 msTiles:0 msStripes:0 msStripesOrdered:0\\n"
           "\\x1b[1m\\x1b[92m@computing SpMV   with func: CUDA CSR 1 at:0x40d858\\n\\x1b[0m"
           "cudaBlockSize: 256 1 1\\tcudaGridSize: 38 1 1\\t\\ttimeAvg:4.828000e-05 timeVar:4.415998e-13\\t"
           "timeInternalAvg:4.828000e-05 timeInternalVar:4.415998e-13 \\n#perf HIP CSR 1\\tseconds:1e-5\\n").encode().decode("unicode_escape")
    rows = _parser().parse(log.splitlines(True))
    assert len(rows) == 1
    x = rows[0]
    assert x["funcID"] == "CUDA CSR 1" and x["source"] == "A.mtx" and x["blockSize_x"] == 256 and x["gridSize_x"] == 38
    assert x["timeAvg"] == 4.828e-05 and x["threadNum"] is None
    assert abs(x["GFLOPS"] - 2 * 76367 / 4.828e-05 * 1e-9) < 1e-9

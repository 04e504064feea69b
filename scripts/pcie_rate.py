"""PCIe-inclusive rate of the host-pointer boundary (spmvHipWarpPerRowCSR: SPMV_INTERF-style, host x in, host y out;
matrix uploaded once and cached): what a caller that keeps its vectors on the host gets.  DESIGN.md section 6."""
import ctypes as C, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from spmv_openmp_cuda_amd import api, synth
from conftest import Oracle
api.spmvHipInit(0)
ora = Oracle()
for key in sys.argv[1:] or ["c2"]:
    w = synth.WORKLOADS[key]
    lens = synth.row_lengths(w); irp = synth.prefix(lens)
    ja32, as_ = ora.synth_fill(w.N, 0, irp, synth.SEED_STRUCT + w.cfg, synth.SEED_VAL + w.cfg, w.band)
    host = api.HostCSR(w.N, w.N, irp, ja32.astype(np.uint64), as_)
    del ja32
    x = synth.make_x(w.N, w.cfg); y = np.empty(w.N)
    cfg = api.CONFIG()
    nnz = int(irp[-1])
    for name in ("spmvHipRowsCSR", "spmvHipWarpPerRowCSR"):      # the two reference-named wrappers, library-default variants
        fn = getattr(api.lib, name)
        t0 = time.perf_counter()
        assert fn(C.byref(host.struct), api._ptr(x), C.byref(cfg), api._ptr(y)) == 0
        t_first = time.perf_counter() - t0
        ts = []
        for _ in range(10):
            t0 = time.perf_counter()
            assert fn(C.byref(host.struct), api._ptr(x), C.byref(cfg), api._ptr(y)) == 0
            ts.append(time.perf_counter() - t0)
        t = sum(ts) / len(ts)
        print(f"{w.name} {name}: first call (upload {nnz * 16 / 1e9:.2f} GB of host CSR + analysis + kernel selection) {t_first * 1e3:.1f} ms; "
              f"steady call (x up {w.N * 8 / 1e6:.0f} MB, kernel {api.lib.spmvHipLastKernelSeconds() * 1e3:.3f} ms, y down {w.N * 8 / 1e6:.0f} MB) "
              f"{t * 1e3:.3f} ms = {2 * nnz / t * 1e-9:.1f} GFLOP/s PCIe-inclusive", flush=True)
        api.lib.spmvHipDropCache()

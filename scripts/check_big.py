"""Spot-check the full-size workloads: GPU y vs oracle on row ranges at the start, middle and end."""
import sys, os, ctypes as C
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from spmv_openmp_cuda_amd import api, synth
from conftest import Oracle
key = sys.argv[1] if len(sys.argv) > 1 else "c5"
oracle = Oracle()
api.spmvHipInit(0)
w = synth.WORKLOADS[key]
lens = synth.row_lengths(w); irp = synth.prefix(lens)
dm = synth.device_csr(w, irp, 0, w.N)
x = synth.make_x(w.N, w.cfg)
dx = api.DeviceVector(w.N).up(x); dy = api.DeviceVector(w.N)
for launcher in ("hipSpMVWarpPerRowCSR", "hipSpMVRowsCSR"):
    dy.poison()
    api.spmv(launcher, dm, dx, dy)
    print(launcher, "kernel ms", api.lib.spmvHipLastKernelSeconds() * 1e3)
    y = dy.down()
    print(" nan count", int(np.isnan(y).sum()))
    S = 200_000
    for r0 in (0, w.N // 2, w.N - S, int(np.argmax(lens)) // 1000 * 1000):
        r1 = min(r0 + S, w.N)
        ja, as_ = oracle.synth_fill(w.N, r0, irp[r0:r1 + 1], synth.SEED_STRUCT + w.cfg, synth.SEED_VAL + w.cfg, w.band)
        yr = oracle.csr_serial_dev((irp[r0:r1 + 1] - irp[r0]).astype(np.uint32), ja, as_, x)
        d = np.abs(yr - y[r0:r1])
        print(f"  rows [{r0},{r1}): max|dy| {d.max():.3e}  exact {np.array_equal(yr, y[r0:r1])}  sum|y| {np.abs(yr).sum():.6e}")
    # column statistics of a few rows
    ja_dev = dm.buffers["JA"].down(np.uint32)
    for r in (5, w.N // 2 + 3):
        print("  row", r, "cols", ja_dev[int(irp[r]):int(irp[r + 1])][:8])
    del ja_dev

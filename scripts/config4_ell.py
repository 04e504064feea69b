"""BASELINE config 4: the power-law matrix clipped to 64 slots, ELL (transposed+pitched thread-per-row,
row-major thread-per-row, row-major lanes-per-row) with and without the row-length early exit, against
the CSR launchers on the SAME clipped matrix.  Writes a markdown table on stdout."""
import ctypes as C
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np

from spmv_openmp_cuda_amd import api, synth
from conftest import Oracle

scale = float(sys.argv[1]) if len(sys.argv) > 1 else 1.0
band = int(sys.argv[2]) if len(sys.argv) > 2 else 0
oracle = Oracle()
api.spmvHipInit(0)
w = synth.WORKLOADS["c4"]
if scale != 1.0:
    w = synth.scaled(w, scale)
    w.clip = 64
if band:
    w = synth.Workload(w.name + f"-band{band}", w.N, w.nnz, w.law, w.max_row, w.cfg, band, w.clip)
lens = synth.row_lengths(w)
irp = synth.prefix(lens)
nnz, M, K = int(irp[-1]), w.N, int(lens.max())
dm = synth.device_csr(w, irp, 0, M)
x = synth.make_x(M, w.cfg)
dx = api.DeviceVector(M).up(x)
dy = api.DeviceVector(M)
# oracle on a 300 k-row window (twin-generated) as the checker
r0, r1 = M // 3, M // 3 + 300_000
ja, as_ = oracle.synth_fill(w.N, r0, irp[r0:r1 + 1], synth.SEED_STRUCT + w.cfg, synth.SEED_VAL + w.cfg, w.band)
y_ref = oracle.csr_serial_dev((irp[r0:r1 + 1] - irp[r0]).astype(np.uint32), ja, as_, x)
b_csr = synth.algorithmic_bytes_csr(nnz, M, M)
b_ell = M * K * 12 + M * 8 + M * 8
print(f"# config 4: {w.name}  M=N={M}  nnz={nnz}  K(max row)={K}  padding ratio M*K/nnz = {M * K / nnz:.2f}")
print(f"# algorithmic bytes: CSR / ELL with row lengths {b_csr / 1e9:.2f} GB, ELL all slots {b_ell / 1e9:.2f} GB")
print("| format / launcher | row-lens early exit | kernel ms | GFLOP/s | GB/s (algorithmic) | % of 8 TB/s | max abs dy (300 k rows) |")
print("|---|---|---|---|---|---|---|")


def run(label, launcher, mat, rl, nbytes):
    api.lib.spmvHipSetEllRowLens(1 if rl else 0)
    api.spmv(launcher, mat, dx, dy)            # warm-up (builds formats)
    ts = []
    for _ in range(10):
        dy.poison()
        api.spmv(launcher, mat, dx, dy)
        ts.append(api.lib.spmvHipLastKernelSeconds())
    y = dy.down()
    assert not np.isnan(y).any()
    err = np.max(np.abs(y[r0:r1] - y_ref))
    assert err <= 7e-4
    t = sum(ts) / len(ts)
    print(f"| {label} | {'yes' if rl else 'no' if rl is not None else '-'} | {t * 1e3:.3f} | {2 * nnz / t * 1e-9:.1f} | "
          f"{nbytes / t * 1e-9:.0f} | {100 * nbytes / t / 8e12:.1f} | {err:.2e} |")


for name in ("hipSpMVRowsCSR", "hipSpMVWarpPerRowCSR", "hipSpMVTilesCSR"):
    run("CSR " + name, name, dm, None, b_csr)
ell_t = api.csr_to_ell_device(dm, True)
for rl in (True, False):
    run("ELL transposed+pitched, thread/row (hipSpMVRowsELL)", "hipSpMVRowsELL", ell_t, rl, b_csr if rl else b_ell)
ell_t.free()
ell = api.csr_to_ell_device(dm, False)
for rl in (True, False):
    run("ELL row-major, lanes/row (hipSpMVWarpsPerRowELLNTrasposed)", "hipSpMVWarpsPerRowELLNTrasposed", ell, rl, b_csr if rl else b_ell)
for rl in (True, False):
    run("ELL row-major, thread/row (hipSpMVRowsELLNNTransposed)", "hipSpMVRowsELLNNTransposed", ell, rl, b_csr if rl else b_ell)
api.lib.spmvHipSetEllRowLens(1)

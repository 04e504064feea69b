#!/usr/bin/env python3
"""Can the stripes kernel (bounded by the CUs' L1 fill path) and the two-phase kernel (bounded by the fabric) share the
chip?  Rows [0, Ms) of the workload go to hipSpMVStripesCSR on a grid of G workgroups (SPMV_SB_GRID=G in the
environment, one per CU), rows [Ms, M) to hipSpMVTilesCSR on a second stream (its workgroups take the CUs the
stripes grid leaves free).  Prints each part alone and both together (wall time between events on a third stream).
    SPMV_SB_GRID=208 python3 scripts/concurrency_probe.py c3 0.75"""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
from spmv_openmp_cuda_amd import api, synth
import bench

key, frac = sys.argv[1], float(sys.argv[2])
w = synth.WORKLOADS[key]
torch.cuda.set_device(0)
api.spmvHipInit(0)
api.lib.spmvHipSetSync(0)
lens = synth.row_lengths(w)
irp = synth.prefix(lens)
nnz = int(irp[-1])
# the split row: `frac` of the entries
Ms = int(np.searchsorted(irp, np.uint64(frac * nnz)))
A = synth.device_csr(w, irp, 0, Ms)
B = synth.device_csr(w, irp, Ms, w.N)
x_host = synth.make_x(w.N, w.cfg)
x = torch.from_numpy(x_host).cuda()
y = torch.full((w.N,), float("nan"), dtype=torch.float64, device="cuda")
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
cfg = api.CONFIG()
fs, ft = api.SPMV_LAUNCHERS["hipSpMVStripesCSR"], api.SPMV_LAUNCHERS["hipSpMVTilesCSR"]
yA, yB = y.data_ptr(), y.data_ptr() + 8 * Ms


def run_a():
    api.lib.spmvHipSetStream(C.c_void_p(s1.cuda_stream))
    assert fs(C.byref(A.handle), x.data_ptr(), cfg, yA) == 0


def run_b():
    api.lib.spmvHipSetStream(C.c_void_p(s2.cuda_stream))
    assert ft(C.byref(B.handle), x.data_ptr(), cfg, yB) == 0


def timed(fn, steps=20):
    # steps are enqueued back to back (a host synchronise between them lets the GPU drop its clocks)
    cur = torch.cuda.current_stream()
    evs = []
    for i in range(steps + 5):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(cur)
        s1.wait_stream(cur); s2.wait_stream(cur)
        fn()
        cur.wait_stream(s1); cur.wait_stream(s2)
        e1.record(cur)
        evs.append((e0, e1))
    torch.cuda.synchronize()
    out = [a.elapsed_time(b) for a, b in evs[5:]]
    return sum(out) / len(out), min(out)


def both():
    run_a()
    run_b()


def both_ba():
    run_b()
    run_a()


ta = timed(run_a)
tb = timed(run_b)
tab = timed(both)
tba = timed(both_ba)
alg = synth.algorithmic_bytes_csr(nnz, w.N, w.N)
print(f"{w.name} split at row {Ms} ({frac:.2f} of the entries) SPMV_SB_GRID={os.environ.get('SPMV_SB_GRID')}: "
      f"stripes part alone {ta[0]:.4f} ms, two-phase part alone {tb[0]:.4f} ms, together {tab[0]:.4f} (min {tab[1]:.4f}) / tiles first {tba[0]:.4f} (min {tba[1]:.4f}) ms "
      f"= {alg / min(tab[0], tba[0]) / 8e9 * 100:.1f}% of 8 TB/s")
win = bench.OracleWindows(synth, w, irp, x_host, lens)
par = win.check(lambda r0, r1: y[r0:r1].cpu().numpy())
print("check", "ok" if par["ok"] and par["max_diff_over_sum_abs_ax"] <= 1e-12 else "FAIL " + str(par))

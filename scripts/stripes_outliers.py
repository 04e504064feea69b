#!/usr/bin/env python3
"""Which workgroups make a slow launch of the stripes kernel slow?  Tuning build with -DSPMV_SB_DEBUG (make dbglib) through
SPMV_LIB; launches the kernel `n` times, and for the slowest and a typical launch prints, per XCD and round of bins, when
the bins ended and the five slowest bins.
    SPMV_LIB=.../libspmvhip_dbg.so python3 scripts/stripes_outliers.py c3 40"""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
from spmv_openmp_cuda_amd import api, synth

key = sys.argv[1] if len(sys.argv) > 1 else "c3"
n = int(sys.argv[2]) if len(sys.argv) > 2 else 40
w = synth.WORKLOADS[key]
torch.cuda.set_device(0)
api.spmvHipInit(0)
lens = synth.row_lengths(w)
irp = synth.prefix(lens)
dm = synth.device_csr(w, irp, 0, w.N)
x = torch.from_numpy(synth.make_x(w.N, w.cfg)).cuda()
y = torch.empty(w.N, dtype=torch.float64, device="cuda")
fn = api.SPMV_LAUNCHERS["hipSpMVStripesCSR"]
cfg = api.CONFIG()
api.lib.spmvHipStripesDebugDump.argtypes = [C.c_void_p, C.c_size_t]
nb = C.c_uint()
for _ in range(4):
    assert fn(C.byref(dm.handle), x.data_ptr(), cfg, y.data_ptr()) == 0
api.lib.spmvHipStripesShape(C.byref(dm.handle), C.byref(nb), None, None, None)
B = nb.value
runs = []
for i in range(n):
    # keep the GPU busy between the synchronous launches (clocks)
    for _ in range(2):
        assert fn(C.byref(dm.handle), x.data_ptr(), cfg, y.data_ptr()) == 0
    ms = api.lib.spmvHipLastKernelSeconds() * 1e3
    buf = np.zeros(8 * 8192, dtype=np.uint64)
    assert api.lib.spmvHipStripesDebugDump(buf.ctypes.data_as(C.c_void_p), buf.size) == 0
    runs.append((ms, buf.reshape(8192, 8)[:min(B, 8192)].copy()))
times = np.array([r[0] for r in runs])
print("kernel ms:", " ".join(f"{t:.3f}" for t in times))
order = np.argsort(times)


def show(tag, idx):
    ms, d = runs[idx]
    xcc = (d[:, 0] >> np.uint64(32)).astype(int)
    t = d[:, 1:6].astype(np.float64)
    t = (t - t[:, 0].min()) / 100.0
    o = np.argsort(t[:, 0])
    rnd = np.empty(len(xcc), dtype=int)
    rnd[o] = np.arange(len(xcc)) // 256
    print(f"-- {tag}: launch {idx}, {ms:.3f} ms")
    for r in range(int(rnd.max()) + 1):
        line = []
        for xc in range(8):
            m = (xcc == xc) & (rnd == r)
            if m.any():
                line.append(f"x{xc} {t[m, 0].min():6.1f}-{t[m, 0].max():6.1f} -> {t[m, 4].min():6.1f}-{t[m, 4].max():6.1f}")
        print(f"   round {r}: " + " | ".join(line))
    dur = t[:, 4] - t[:, 0]
    worst = np.argsort(dur)[-5:][::-1]
    print("   longest bins: " + ", ".join(f"bin {b} xcc {xcc[b]} round {rnd[b]} {dur[b]:.1f} us (quarters {t[b,1]-t[b,0]:.0f}/{t[b,2]-t[b,1]:.0f}/{t[b,3]-t[b,2]:.0f}/{t[b,4]-t[b,3]:.0f})" for b in worst))
    print(f"   bin duration: median {np.median(dur):.1f}  p95 {np.percentile(dur, 95):.1f}  max {dur.max():.1f}")


# who straggles?  (HW_ID: CU_ID bits 11:8, SH_ID 12, SE_ID 15:13)
med = np.median(times)
print("stragglers of the slow launches (kernel > 1.05 x median): launch, ms, bin, xcc, se, sh, cu, duration us")
for i in np.nonzero(times > 1.05 * med)[0]:
    ms, d = runs[i]
    t = d[:, 1:6].astype(np.float64) / 100.0
    dur = t[:, 4] - t[:, 0]
    b = int(np.argmax(dur))
    hw = int(d[b, 0] & np.uint64(0xFFFFFFFF))
    print(f"   {i:3d} {ms:.3f} bin {b:3d} wg {b % 256:3d} xcc {int(d[b, 0] >> np.uint64(32))} se {(hw >> 13) & 7} sh {(hw >> 12) & 1} cu {(hw >> 8) & 15}  {dur[b]:.1f}")
# placement of workgroup -> CU in the last launch
d = runs[-1][1]
hw = (d[:256, 0] & np.uint64(0xFFFFFFFF)).astype(np.int64)
print("placement (last launch) wg: xcc/se/cu for wgs 0..31:", " ".join(f"{int(d[i, 0] >> np.uint64(32))}/{(int(hw[i]) >> 13) & 7}/{(int(hw[i]) >> 8) & 15}" for i in range(32)))
show("typical", order[len(order) // 2])
show("slowest", order[-1])
show("second slowest", order[-2])
dm.free()
api.spmvHipFinalize()

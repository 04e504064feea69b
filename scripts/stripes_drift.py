#!/usr/bin/env python3
"""How far apart do the workgroups of one XCD run in the stripes kernel?  Needs a tuning build with
-DSPMV_SB_DEBUG (make dbglib) loaded through SPMV_LIB; prints, per XCD and per round of bins, the spread of the
start / quarter / half / three-quarter / end stamps (microseconds).
    SPMV_LIB=spmv_openmp_cuda_amd/lib/libspmvhip_dbg.so python3 scripts/stripes_drift.py c3 [scale]"""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
from spmv_openmp_cuda_amd import api, synth

key = sys.argv[1] if len(sys.argv) > 1 else "c3"
w = synth.WORKLOADS[key]
if len(sys.argv) > 2:
    w = synth.scaled(w, float(sys.argv[2]))
torch.cuda.set_device(0)
api.spmvHipInit(0)
lens = synth.row_lengths(w)
irp = synth.prefix(lens)
dm = synth.device_csr(w, irp, 0, w.N)
x = torch.from_numpy(synth.make_x(w.N, w.cfg)).cuda()
y = torch.empty(w.N, dtype=torch.float64, device="cuda")
fn = api.SPMV_LAUNCHERS["hipSpMVStripesCSR"]
cfg = api.CONFIG()
for _ in range(4):
    assert fn(C.byref(dm.handle), x.data_ptr(), cfg, y.data_ptr()) == 0
print("kernel ms", api.lib.spmvHipLastKernelSeconds() * 1e3)
nb = C.c_uint()
api.lib.spmvHipStripesShape(C.byref(dm.handle), C.byref(nb), None, None, None)
B = nb.value
buf = np.zeros(8 * 8192, dtype=np.uint64)
api.lib.spmvHipStripesDebugDump.argtypes = [C.c_void_p, C.c_size_t]
assert api.lib.spmvHipStripesDebugDump(buf.ctypes.data_as(C.c_void_p), buf.size) == 0
d = buf.reshape(8192, 8)[:min(B, 8192)]
xcc = (d[:, 0] >> np.uint64(32)).astype(int)
hwid = (d[:, 0] & np.uint64(0xFFFFFFFF)).astype(int)
t = d[:, 1:6].astype(np.float64)
t = (t - t[:, 0].min()) / 100.0          # us
print("bins", B, " blocks per xcc:", np.bincount(xcc, minlength=8), " block%8 == xcc for", int((np.arange(len(xcc)) % 8 == xcc).sum()))
order = np.argsort(t[:, 0])
rnd = np.empty(len(xcc), dtype=int)
rnd[order] = np.arange(len(xcc)) // 256
names = ["start", "1/4", "1/2", "3/4", "end"]
for r in range(int(rnd.max()) + 1):
    print(f"-- round {r}")
    for xc in range(8):
        m = (xcc == xc) & (rnd == r)
        if not m.any():
            continue
        tt = t[m]
        print(f"xcc {xc}: n={int(m.sum()):3d} " + "  ".join(f"{n} {tt[:, i].min():7.1f}..{tt[:, i].max():7.1f} (sd {tt[:, i].std():5.1f})" for i, n in enumerate(names)))
dur = t[:, 4] - t[:, 0]
print("bin duration us: min %.1f  median %.1f  max %.1f" % (dur.min(), np.median(dur), dur.max()))
dm.free()
api.spmvHipFinalize()

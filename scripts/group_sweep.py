"""Two-phase SpMV on a workload cut into G consecutive row groups run back to back (each its own slice-major format,
ONE shared product workspace): does keeping a group's products inside the 256 MiB Infinity Cache pay for the extra
x-slice fills?   usage: group_sweep.py <workload> G [G ...]"""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import numpy as np
from spmv_openmp_cuda_amd import api, synth, sharding
api.spmvHipInit(0)
api.lib.spmvHipSetSync(0)
w = synth.WORKLOADS[sys.argv[1]]
lens = synth.row_lengths(w); irp = synth.prefix(lens)
x = synth.make_x(w.N, w.cfg); dx = api.DeviceVector(w.N).up(x)
dy = api.DeviceVector(w.N)
for G in [int(a) for a in sys.argv[2:]]:
    plan = sharding.make_plan(irp, 1, G)
    dms = [synth.device_csr(w, irp, *plan.block(0, g)) for g in range(G)]
    ys = [C.c_void_p(dy.ptr.value + 8 * plan.block(0, g)[0]) for g in range(G)]
    cfg = api.CONFIG()
    def step():
        for dm, yp in zip(dms, ys):
            assert api.lib.hipSpMVTilesCSR(C.byref(dm.handle), dx.ptr, cfg, yp) == 0
    step(); api.lib.spmvHipDeviceSynchronize()
    e0, e1 = C.c_void_p(), C.c_void_p()
    api.lib.spmvHipEventCreate(C.byref(e0)); api.lib.spmvHipEventCreate(C.byref(e1))
    api.lib.spmvHipEventRecord(e0)
    for _ in range(10): step()
    api.lib.spmvHipEventRecord(e1)
    ms = C.c_float(); api.lib.spmvHipEventElapsedMs(e0, e1, C.byref(ms))
    nnz = int(irp[-1])
    alg = synth.algorithmic_bytes_csr(nnz, w.N, w.N)
    print(f"{w.name} G={G}: {ms.value / 10:.4f} ms  {alg / (ms.value / 10 * 1e-3) / 8e12 * 100:.1f} % of 8 TB/s", flush=True)
    for dm in dms: dm.free()

"""Does the push kernel really run BESIDE phase 2?  One GPU, one extra destination on the same GPU (a scratch vector):
phase 2 alone, phase 2 with fused stores to the scratch, phase 2 + push kernel.  If the push kernel only started after
phase 2, its time would simply add (copying M doubles at ~2-3 TB/s)."""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import numpy as np
from spmv_openmp_cuda_amd import api, synth
api.spmvHipInit(0); api.lib.spmvHipSetSync(0)
w = synth.WORKLOADS[sys.argv[1] if len(sys.argv) > 1 else "c3"]
lens = synth.row_lengths(w); irp = synth.prefix(lens)
x = synth.make_x(w.N, w.cfg); dx = api.DeviceVector(w.N).up(x)
dy = api.DeviceVector(w.N); scratch = [api.DeviceVector(w.N) for _ in range(7)]
dm = synth.device_csr(w, irp, 0, w.N)
h = C.byref(dm.handle)
nb, rpb = C.c_uint(), C.c_uint(); api.lib.spmvHipTilesShape(h, C.byref(nb), C.byref(rpb))
def timed(fn, reps=10):
    fn(); api.lib.spmvHipDeviceSynchronize()
    e0, e1 = C.c_void_p(), C.c_void_p(); api.lib.spmvHipEventCreate(C.byref(e0)); api.lib.spmvHipEventCreate(C.byref(e1))
    api.lib.spmvHipEventRecord(e0)
    for _ in range(reps): fn()
    api.lib.spmvHipEventRecord(e1)
    ms = C.c_float(); api.lib.spmvHipEventElapsedMs(e0, e1, C.byref(ms)); return ms.value / reps
for n in (1, 7):
    extra = (C.c_void_p * n)(*[s.ptr.value for s in scratch[:n]])
    t_exp = timed(lambda: api.lib.hipSpMVTilesExpand(h, dx.ptr))
    t_plain = timed(lambda: (api.lib.hipSpMVTilesExpand(h, dx.ptr), api.lib.hipSpMVTilesReduce(h, 0, nb.value, dy.ptr, 0, None)))
    t_fused = timed(lambda: (api.lib.hipSpMVTilesExpand(h, dx.ptr), api.lib.hipSpMVTilesReduce(h, 0, nb.value, dy.ptr, n, extra)))
    t_pushk = timed(lambda: (api.lib.hipSpMVTilesExpand(h, dx.ptr), api.lib.hipSpMVTilesReducePush(h, dy.ptr, n, extra)))
    ok = api.lib.spmvHipTilesPushFailed(h) == 0 and np.array_equal(scratch[n - 1].down(), dy.down())
    print(f"{w.name} {n} extra destination(s) of {w.N * 8 / 1e6:.0f} MB: expand {t_exp:.3f} ms; +reduce {t_plain - t_exp:.3f}; +reduce fused {t_fused - t_exp:.3f}; "
          f"+reduce with push kernel {t_pushk - t_exp:.3f}  (copy complete and identical: {ok})", flush=True)

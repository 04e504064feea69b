"""A slow link on a one-GPU box: the extra destination of phase 2 is PINNED HOST memory, so the stores leave over PCIe
(~50 GB/s) the way the peers' rows leave over xGMI.  The matrix is shaped so that the link time of y is close to the
time of phase 2 (few rows, many nnz per row).  Question: does the exchange hide under phase 2 (max) or add to it (sum),
for the fused store and for the push kernel?"""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import numpy as np, torch
from spmv_openmp_cuda_amd import api, synth
api.spmvHipInit(0); api.lib.spmvHipSetSync(0)
rows = int(sys.argv[1]) if len(sys.argv) > 1 else 2_500_000
w = synth.Workload(f"dense-{rows}", rows, 200_000_000, "powerlaw", cfg=7)
lens = synth.row_lengths(w); irp = synth.prefix(lens)
x = synth.make_x(w.N, w.cfg); dx = api.DeviceVector(w.N).up(x)
dy = api.DeviceVector(w.N)
host = torch.full((w.N,), float("nan"), dtype=torch.float64).pin_memory()
dm = synth.device_csr(w, irp, 0, w.N)
h = C.byref(dm.handle)
nb, rpb = C.c_uint(), C.c_uint(); api.lib.spmvHipTilesShape(h, C.byref(nb), C.byref(rpb))
extra = (C.c_void_p * 1)(host.data_ptr())
def timed(fn, reps=8):
    fn(); api.lib.spmvHipDeviceSynchronize()
    e0, e1 = C.c_void_p(), C.c_void_p(); api.lib.spmvHipEventCreate(C.byref(e0)); api.lib.spmvHipEventCreate(C.byref(e1))
    api.lib.spmvHipEventRecord(e0)
    for _ in range(reps): fn()
    api.lib.spmvHipEventRecord(e1)
    ms = C.c_float(); api.lib.spmvHipEventElapsedMs(e0, e1, C.byref(ms)); return ms.value / reps
copy_ms = timed(lambda: api.lib.spmvHipMemcpyDown(host.data_ptr(), dy.ptr, w.N * 8), 4)
t_exp = timed(lambda: api.lib.hipSpMVTilesExpand(h, dx.ptr))
t_plain = timed(lambda: (api.lib.hipSpMVTilesExpand(h, dx.ptr), api.lib.hipSpMVTilesReduce(h, 0, nb.value, dy.ptr, 0, None))) - t_exp
t_fused = timed(lambda: (api.lib.hipSpMVTilesExpand(h, dx.ptr), api.lib.hipSpMVTilesReduce(h, 0, nb.value, dy.ptr, 1, extra))) - t_exp
host.fill_(float("nan"))
t_pushk = timed(lambda: (api.lib.hipSpMVTilesExpand(h, dx.ptr), api.lib.hipSpMVTilesReducePush(h, dy.ptr, 1, extra), api.lib.spmvHipTilesPushJoin())) - t_exp
api.lib.spmvHipDeviceSynchronize()
ok = api.lib.spmvHipTilesPushFailed(h) == 0 and np.array_equal(host.numpy(), dy.down())
print(f"{w.name}: {nb.value} bins of {rpb.value} rows, y = {w.N * 8 / 1e6:.0f} MB; blocking D2H copy of y {copy_ms:.3f} ms ({w.N * 8 / copy_ms / 1e6:.1f} GB/s); "
      f"phase 2 alone {t_plain:.3f} ms; fused store to host {t_fused:.3f} ms; push kernel to host {t_pushk:.3f} ms (complete and identical: {ok})")

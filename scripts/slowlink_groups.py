"""PCIe stand-in for the links, with ROW GROUPS: G groups run back to back (expand, reduce) and every group's rows go to
pinned host memory -- by the fused store (inside the stream-ordered reduce kernel) or by the push kernel (side stream,
joined at the end) or by the copy engines behind each group.  Measured: only the copy engines let a group's rows
travel under the NEXT group's kernels; CU-issued stores to the slow destination hold the other kernels up."""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import numpy as np, torch
from spmv_openmp_cuda_amd import api, synth, sharding
api.spmvHipInit(0); api.lib.spmvHipSetSync(0)
rows = int(sys.argv[1]) if len(sys.argv) > 1 else 10_000_000
w = synth.Workload(f"dense-{rows}", rows, 200_000_000, "powerlaw", cfg=7)
lens = synth.row_lengths(w); irp = synth.prefix(lens)
x = synth.make_x(w.N, w.cfg); dx = api.DeviceVector(w.N).up(x)
dy = api.DeviceVector(w.N)
host = torch.full((w.N,), float("nan"), dtype=torch.float64).pin_memory()
def timed(fn, reps=6):
    fn(); api.lib.spmvHipDeviceSynchronize()
    e0, e1 = C.c_void_p(), C.c_void_p(); api.lib.spmvHipEventCreate(C.byref(e0)); api.lib.spmvHipEventCreate(C.byref(e1))
    api.lib.spmvHipEventRecord(e0)
    for _ in range(reps): fn()
    api.lib.spmvHipEventRecord(e1)
    ms = C.c_float(); api.lib.spmvHipEventElapsedMs(e0, e1, C.byref(ms)); return ms.value / reps
for G in [int(a) for a in sys.argv[2:]] or [1, 2, 4]:
    plan = sharding.make_plan(irp, 1, G)
    dms = [synth.device_csr(w, irp, *plan.block(0, g)) for g in range(G)]
    hs = [C.byref(d.handle) for d in dms]
    r0s = [plan.block(0, g)[0] for g in range(G)]
    nbs = []
    for h in hs:
        nb, rpb = C.c_uint(), C.c_uint(); api.lib.spmvHipTilesShape(h, C.byref(nb), C.byref(rpb)); nbs.append(nb.value)
    ys = [C.c_void_p(dy.ptr.value + 8 * r) for r in r0s]
    ex = [(C.c_void_p * 1)(host.data_ptr() + 8 * r) for r in r0s]
    def plain():
        for g in range(G):
            api.lib.hipSpMVTilesExpand(hs[g], dx.ptr); api.lib.hipSpMVTilesReduce(hs[g], 0, nbs[g], ys[g], 0, None)
    def fused():
        for g in range(G):
            api.lib.hipSpMVTilesExpand(hs[g], dx.ptr); api.lib.hipSpMVTilesReduce(hs[g], 0, nbs[g], ys[g], 1, ex[g])
    def pushk():
        for g in range(G):
            api.lib.hipSpMVTilesExpand(hs[g], dx.ptr); api.lib.hipSpMVTilesReducePush(hs[g], ys[g], 1, ex[g])
        api.lib.spmvHipTilesPushJoin()
    peer = (C.c_void_p * 1)(host.data_ptr())
    ends = [plan.block(0, g)[1] for g in range(G)]
    def sdma():
        for g in range(G):
            api.lib.hipSpMVTilesExpand(hs[g], dx.ptr); api.lib.hipSpMVTilesReduce(hs[g], 0, nbs[g], ys[g], 0, None)
            api.lib.spmvHipPeerPush(dy.ptr, 8 * r0s[g], 8 * (ends[g] - r0s[g]), 1, peer)
        api.lib.spmvHipPeerPushJoin()
    t3 = timed(sdma)
    t0, t1 = timed(plain), timed(fused)
    host.fill_(float("nan"))
    t2 = timed(pushk)
    api.lib.spmvHipDeviceSynchronize()
    ok = all(api.lib.spmvHipTilesPushFailed(h) == 0 for h in hs) and np.array_equal(host.numpy(), dy.down())
    print(f"{w.name} y = {w.N * 8 / 1e6:.0f} MB to pinned host, G = {G} row groups ({sum(nbs)} bins): kernels alone {t0:.3f} ms; fused store {t1:.3f} ms; "
          f"copy-engine push behind each group {t3:.3f} ms; push kernel {t2:.3f} ms (complete and identical: {ok})", flush=True)
    for d in dms: d.free()

import ctypes as C, os, sys, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo")); sys.path.insert(0, os.path.join(os.environ.get("GRAFT_REPO_ROOT", "/root/repo"), "tests"))
import numpy as np, torch
from spmv_openmp_cuda_amd import api
from conftest import random_csr, Oracle
oracle = Oracle()
torch.cuda.set_device(0)
api.spmvHipInit(0)
rng = np.random.default_rng(5)
M = N = 20000
IRP, JA, AS = random_csr(rng, M, N, rng.integers(0, 30, size=M))
x_host = np.sin(rng.uniform(0, 6.28, N)) * 3e-5
y_ref = oracle.csr_serial(IRP, JA, AS, x_host)
dm = api.spMatCpyCSR(api.HostCSR(M, N, IRP, JA, AS))
s = torch.cuda.Stream()
names = ["hipSpMVRowsCSR", "hipSpMVWarpPerRowCSR", "hipSpMVTilesCSR", "hipSpMVStripesCSR", "hipSpMVRowsSELL"]
cfg = api.CONFIG()
with torch.cuda.stream(s):
    x = torch.from_numpy(x_host).cuda()
    ys = [torch.full((M,), float("nan"), dtype=torch.float64, device="cuda") for _ in names]
    api.lib.spmvHipSetStream(C.c_void_p(s.cuda_stream)); api.lib.spmvHipSetSync(0)
    for n, y in zip(names, ys):
        assert api.SPMV_LAUNCHERS[n](C.byref(dm.handle), x.data_ptr(), cfg, y.data_ptr()) == 0   # warm-up: builds the formats
    torch.cuda.synchronize()
    for y in ys: y.fill_(float("nan"))
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, stream=s):
        for n, y in zip(names, ys):
            rc = api.SPMV_LAUNCHERS[n](C.byref(dm.handle), x.data_ptr(), cfg, y.data_ptr())
            assert rc == 0, n
    torch.cuda.synchronize()
    print("after capture (nothing ran yet): all NaN =", all(bool(torch.isnan(y).all()) for y in ys))
    g.replay(); torch.cuda.synchronize()
    for n, y in zip(names, ys):
        print(n, "max|dy| =", float(np.max(np.abs(y.cpu().numpy() - y_ref))))
    # launch-bound case: 200 SpMVs of the small matrix, graph vs individual launches
    g2 = torch.cuda.CUDAGraph()
    fn = api.SPMV_LAUNCHERS["hipSpMVWarpPerRowCSR"]
    with torch.cuda.graph(g2, stream=s):
        for _ in range(200): fn(C.byref(dm.handle), x.data_ptr(), cfg, ys[1].data_ptr())
    torch.cuda.synchronize()
    for rep in range(3):
        t0 = time.perf_counter(); g2.replay(); torch.cuda.synchronize(); tg = time.perf_counter() - t0
        t0 = time.perf_counter()
        for _ in range(200): fn(C.byref(dm.handle), x.data_ptr(), cfg, ys[1].data_ptr())
        torch.cuda.synchronize(); tl = time.perf_counter() - t0
        print(f"200 SpMVs ({int(IRP[-1])} nnz): graph replay {tg*1e3:.3f} ms, individual launches {tl*1e3:.3f} ms")

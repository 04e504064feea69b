"""Time the kernels on ONE rank's shard of c5 as it would be at N ranks (rows [0, M/N) of the 80 M-row matrix,
all 80 M columns): the shape the multi-GPU run gives every GPU, measurable on one GPU."""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import numpy as np
from spmv_openmp_cuda_amd import api, synth, sharding
N = int(sys.argv[1]) if len(sys.argv) > 1 else 8
G = int(sys.argv[2]) if len(sys.argv) > 2 else 1
api.spmvHipInit(0)
w = synth.WORKLOADS["c5"]
lens = synth.row_lengths(w); irp = synth.prefix(lens)
plan = sharding.make_plan(irp, N, G)
x = synth.make_x(w.N, w.cfg); dx = api.DeviceVector(w.N).up(x)
tot = {}
for g in range(G):
    b0, b1 = plan.block(0, g)
    dm = synth.device_csr(w, irp, b0, b1)
    dy = api.DeviceVector(b1 - b0)
    for launcher in ("hipSpMVWarpPerRowCSR", "hipSpMVTilesCSR"):
        api.spmv(launcher, dm, dx, dy)
        ts = []
        for _ in range(5):
            api.spmv(launcher, dm, dx, dy); ts.append(api.lib.spmvHipLastKernelSeconds())
        tot[launcher] = tot.get(launcher, 0) + sum(ts) / len(ts)
    dm.free(); dy.free()
print(f"N={N} G={G} rows/rank={plan.rows(0)[1]}: " + "  ".join(f"{k} {v*1e3:.3f} ms" for k, v in tot.items()))

#!/bin/bash
# round 3, second GPU call: deterministic two-phase form + the slimmer format build
set -o pipefail
O=gpurun_out/r03_second
mkdir -p $O
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "deterministic or tiles or full_size or auto" > $O/pytest.log 2>&1
echo "pytest rc=$?" | tee -a $O/pytest.log
tail -15 $O/pytest.log
for w in c3 c5; do
  timeout -k 10 400 python scripts/time_launchers.py $w hipSpMVTilesCSR hipSpMVTilesCSR:det --check >> $O/time.log 2>&1 || echo "time_launchers $w failed" >> $O/time.log
done
cat $O/time.log
timeout -k 10 300 python - >> $O/build.log 2>&1 <<'PY'
import sys, time
sys.path.insert(0, ".")
import torch
from spmv_openmp_cuda_amd import api, synth
api.spmvHipInit(0)
for key in ("c3", "c5"):
    w = synth.WORKLOADS[key]
    irp = synth.prefix(synth.row_lengths(w))
    dm = synth.device_csr(w, irp, 0, w.N)
    for det in (False, True):
        t0 = time.perf_counter()
        api.build_tiles(dm, deterministic=det)
        wall = time.perf_counter() - t0
        i = api.tiles_info(dm)
        print(f"{key} tiles det={det}: wall {wall*1e3:.0f} ms  buildMs {i.buildMs:.0f}  allocMs {i.allocMs:.0f}  format {i.bytes/1e9:.2f} GB  temporaries {i.tempBytes/1e9:.2f} GB = {i.tempBytes/dm.nnz:.1f} B/nnz  bins {i.nBins} x {i.rowsPerBin}", flush=True)
    if key == "c3":
        for det in (False, True):
            t0 = time.perf_counter()
            api.build_stripes(dm, deterministic=det)
            wall = time.perf_counter() - t0
            s = api.stripes_info(dm)
            print(f"{key} stripes det={det}: wall {wall*1e3:.0f} ms  buildMs {s.buildMs:.0f}  format {s.bytes/1e9:.2f} GB", flush=True)
    dm.free()
PY
cat $O/build.log

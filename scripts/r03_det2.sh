#!/bin/bash
# experiment: deterministic stripes by ordered tickets (one shared column-ordered stream, adds handed over batch by batch)
set -o pipefail
O=gpurun_out/r03_det2
mkdir -p $O
for w in c2 c3 c3b; do
  timeout -k 10 300 python scripts/time_launchers.py $w hipSpMVStripesCSR hipSpMVStripesCSR:det hipSpMVStripesCSR:det2 hipSpMVTilesCSR:det --check >> $O/time.log 2>&1 || echo "time_launchers $w failed" >> $O/time.log
done
cat $O/time.log
timeout -k 10 300 python - > $O/bits.log 2>&1 <<'PY'
import sys
sys.path.insert(0, "."); sys.path.insert(0, "tests")
import numpy as np
from spmv_openmp_cuda_amd import api, synth
from conftest import Oracle
oracle = Oracle()
api.spmvHipInit(0)
for key in ("tiny", "c2", "c3"):
    w = synth.WORKLOADS[key]
    lens = synth.row_lengths(w); irp = synth.prefix(lens)
    dm = synth.device_csr(w, irp, 0, w.N)
    x = synth.make_x(w.N, w.cfg)
    dx, dy = api.DeviceVector(w.N).up(x), api.DeviceVector(w.N)
    r1 = min(w.N, 300_000)
    ja, as_ = oracle.synth_fill(w.N, 0, irp[:r1 + 1], synth.SEED_STRUCT + w.cfg, synth.SEED_VAL + w.cfg, w.band)
    yr = oracle.csr_serial_dev(irp[:r1 + 1].astype(np.uint32), ja, as_, x)
    for det in (1, 2):
        api.build_stripes(dm, deterministic=det)
        ys = []
        for _ in range(4):
            dy.poison(); api.spmv("hipSpMVStripesCSR", dm, dx, dy); ys.append(dy.down())
        print(key, "det", det, "serial bits:", np.array_equal(ys[0][:r1], yr), "repeatable:", all(np.array_equal(ys[0], y) for y in ys), flush=True)
    dm.free()
PY
cat $O/bits.log

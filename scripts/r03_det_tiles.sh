#!/bin/bash
set -o pipefail
O=gpurun_out/r03_det_tiles
mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "deterministic or tiles" > $O/pytest.log 2>&1
echo "pytest rc=$?"; tail -5 $O/pytest.log
for w in c2 c3 c5; do
  timeout -k 10 400 python scripts/time_launchers.py $w hipSpMVTilesCSR hipSpMVTilesCSR:det --check >> $O/time.log 2>&1 || echo "time_launchers $w failed" >> $O/time.log
done
cat $O/time.log

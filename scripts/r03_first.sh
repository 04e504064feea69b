#!/bin/bash
# round 3, first GPU call: the whole -m gpu suite on the cleaned library, then the reference-named modes and the
# deterministic stripes form on c2 / c3
set -o pipefail
O=gpurun_out/r03_first
mkdir -p $O
timeout -k 10 1000 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1
echo "pytest rc=$?" | tee -a $O/pytest.log
tail -15 $O/pytest.log
for w in c2 c3; do
  timeout -k 10 300 python scripts/time_launchers.py $w CUDA_CSR_ROWS_WARP hipSpMVWarpPerRowCSR:1 hipSpMVStripesCSR hipSpMVStripesCSR:det hipSpMVTilesCSR --check >> $O/time.log 2>&1 || echo "time_launchers $w failed" >> $O/time.log
done
cat $O/time.log

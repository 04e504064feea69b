#!/bin/bash
# what the reference's two CSR mode names resolve to and cost on the synthetic workloads (library defaults)
set -o pipefail
O=gpurun_out/r03_names
mkdir -p $O
for w in c2 c3 c3b c5; do
  timeout -k 10 400 python scripts/time_launchers.py $w CUDA_CSR_ROWS CUDA_CSR_ROWS_WARP --check >> $O/names.log 2>&1 || echo "failed $w" >> $O/names.log
done
grep -v amdgpu.ids $O/names.log
timeout -k 10 300 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "deterministic or auto or full_size" > $O/pytest.log 2>&1
echo "pytest rc=$?"; tail -4 $O/pytest.log

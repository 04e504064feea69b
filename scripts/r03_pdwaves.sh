#!/bin/bash
set -o pipefail
O=gpurun_out/r03_pdwaves
mkdir -p $O
for v in 4 16; do
  for w in c3 c5; do
    echo "PD_WAVES=$v" >> $O/time.log
    SPMV_LIB=$PWD/spmv_openmp_cuda_amd/lib/libspmvhip_pd$v.so timeout -k 10 400 python scripts/time_launchers.py $w hipSpMVTilesCSR:det --check >> $O/time.log 2>&1 || echo "failed $v $w" >> $O/time.log
  done
done
grep -v amdgpu.ids $O/time.log
# the whole suite on the current default library (4-candidate serial-order selection, stripes modes)
timeout -k 10 900 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1
echo "pytest rc=$?"; tail -6 $O/pytest.log
for w in c2 c3 c3b; do
  timeout -k 10 300 python scripts/time_launchers.py $w CUDA_CSR_ROWS CUDA_CSR_ROWS_WARP --check >> $O/names.log 2>&1
done
grep -v amdgpu.ids $O/names.log

#!/bin/bash
set -o pipefail
mkdir -p gpurun_out/r03_fuzz
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "fuzz" > gpurun_out/r03_fuzz/pytest.log 2>&1
echo "pytest rc=$?"; tail -12 gpurun_out/r03_fuzz/pytest.log

#!/bin/bash
set -o pipefail
mkdir -p gpurun_out/r03_cli
timeout -k 10 600 python -m pytest tests/test_gpu_programs.py -m gpu -x -q -k "cli" > gpurun_out/r03_cli/pytest.log 2>&1
echo "pytest rc=$?"; tail -8 gpurun_out/r03_cli/pytest.log

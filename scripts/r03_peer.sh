#!/bin/bash
set -o pipefail
mkdir -p gpurun_out/r03_peer
timeout -k 10 600 python -m pytest tests/test_gpu_peer.py -m gpu -x -q > gpurun_out/r03_peer/pytest.log 2>&1
echo "pytest rc=$?"; tail -12 gpurun_out/r03_peer/pytest.log

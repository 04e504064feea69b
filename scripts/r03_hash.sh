#!/bin/bash
set -o pipefail
O=gpurun_out/r03_hash
mkdir -p $O
timeout -k 10 900 python -m pytest tests/test_gpu_programs.py tests/test_gpu_peer.py -m gpu -x -q > $O/pytest.log 2>&1
echo "pytest rc=$?"; tail -15 $O/pytest.log

#!/bin/bash
# round 3: full GPU suite, then the round's profile run (scripts/final_profile.sh r03)
set -o pipefail
mkdir -p gpurun_out/r03_suite
timeout -k 10 900 python -m pytest tests -m gpu -x -q > gpurun_out/r03_suite/pytest.log 2>&1
echo "pytest rc=$?"; tail -6 gpurun_out/r03_suite/pytest.log
bash scripts/final_profile.sh r03

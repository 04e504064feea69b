#!/bin/bash
# round 3: full GPU suite, PCIe-inclusive rate of the host-pointer wrappers, then the round's profile run (scripts/final_profile.sh r03)
set -o pipefail
mkdir -p gpurun_out/r03_suite
timeout -k 10 900 python -m pytest tests -m gpu -x -q > gpurun_out/r03_suite/pytest.log 2>&1
echo "pytest rc=$?"; tail -6 gpurun_out/r03_suite/pytest.log
timeout -k 10 600 python scripts/pcie_rate.py c2 c3 > gpurun_out/r03_suite/pcie.log 2>&1; grep -v amdgpu.ids gpurun_out/r03_suite/pcie.log
timeout -k 10 300 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/r03_suite/smoke.log 2>&1; tail -12 gpurun_out/r03_suite/smoke.log
bash scripts/final_profile.sh r03

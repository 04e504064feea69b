#!/bin/bash
set -o pipefail
mkdir -p gpurun_out/r03_suite
timeout -k 10 1000 python -m pytest tests -m gpu -x -q > gpurun_out/r03_suite/pytest.log 2>&1
echo "pytest rc=$?"; tail -8 gpurun_out/r03_suite/pytest.log
timeout -k 10 300 python bench.py > gpurun_out/r03_suite/bench.json 2> gpurun_out/r03_suite/bench.err
echo "bench rc=$?"
python - <<'PY'
import json
l = json.load(open("gpurun_out/r03_suite/bench.json"))
print(l["value"], l["ms_per_step"], l["roofline"]["frac"], "traffic", l["roofline"]["traffic"], l["roofline"]["traffic_source"])
PY

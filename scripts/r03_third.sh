#!/bin/bash
# round 3, third GPU call: full suite with the two selections, default bench line
set -o pipefail
O=gpurun_out/r03_third
mkdir -p $O
timeout -k 10 1000 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1
echo "pytest rc=$?" | tee -a $O/pytest.log
tail -15 $O/pytest.log
timeout -k 10 900 python bench.py > $O/bench.json 2> $O/bench.err
echo "bench rc=$?"
tail -30 $O/bench.err
python - <<'PY'
import json
l = json.load(open("gpurun_out/r03_third/bench.json"))
print({k: l[k] for k in ("value", "ms_per_step")}, l["roofline"]["frac"], l["config"].get("library_auto_choice"))
for k in ("headline_c3", "extra_c3b", "extra_c2"):
    b = l.get(k)
    if b: print(k, b["launcher"], b["kernel_ms_avg"], b["hbm_frac"], b.get("library_auto_choice"), b.get("library_serial_order_choice"))
PY

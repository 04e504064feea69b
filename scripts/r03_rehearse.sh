#!/bin/bash
# shared-GPU rehearsals of the N > 1 bench path with more ranks and the extra exchange candidates (timings mean nothing;
# at most 5 ranks: the box allows 6 processes on its GPU)
set -o pipefail
O=gpurun_out/r03_rehearse
mkdir -p $O
run() {
  tag=$1; shift
  timeout -k 10 500 python bench.py --rehearse-shared-gpu --steps 3 --warmup 1 --exchange-budget 120 "$@" > $O/$tag.json 2> $O/$tag.err
  echo "$tag rc=$?"
  grep "exchange candidates\|PARITY\|Error\|error" $O/$tag.err | cut -c1-400
  python - <<PY
import json
l = json.load(open("$O/$tag.json"))
c = l["config"]
print("$tag", l["n_gpus"], c["kernel"], c["exchange"], "parity", l["parity"]["ok"], "ms", {k: round(v, 2) for k, v in c["exchange_step_ms"].items()}, "rejected:", c["exchange_rejected"], "skipped:", list(c["exchange_skipped"]))
PY
  sleep 3
}
run n4_tiles_extra --gpus 4 --scale 0.1 --launcher hipSpMVTilesCSR --exchange-extra
run n5_auto --gpus 5 --scale 0.05
run n3_serial --gpus 3 --scale 0.05 --launcher hipSpMVRowsCSR --variant 2 --y-hash

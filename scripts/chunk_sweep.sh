#!/bin/bash
# two-phase kernel vs phase-1 work item size (SPMV_PB_CHUNK, tuning knob of tiles.hip)
cd $GRAFT_REPO_ROOT
for w in ${WL:-c3 c5}; do
  for r in ${CHUNKS:-65536 98304 131072 170000 262144 524288}; do
    export SPMV_PB_CHUNK=$r
    timeout -k 10 200 python3 bench.py --workload $w --launcher hipSpMVTilesCSR --steps 10 --warmup 2 --no-cpu-baseline --no-extra 2>/dev/null | python3 -c "import sys,json; j=json.loads(sys.stdin.read()); print('$w chunk=$r', round(j['roofline']['kernel_ms_avg'],4), 'ms', round(j['roofline']['frac'],4))" || exit 1
  done
done
for n in ${SHARDS:-2}; do
  for r in ${CHUNKS:-65536 98304 131072 170000 262144 524288}; do
    export SPMV_PB_CHUNK=$r
    timeout -k 10 200 python3 scripts/shard_shape.py $n 1 2>/dev/null | tail -1 || exit 1
  done
done

#!/bin/bash
# two-phase kernel vs rows per bin (SPMV_PB_ROWS, tuning knob of tiles.hip): whole workloads and shard shapes
out=$GRAFT_REPO_ROOT/gpurun_out/${1:-rows}; mkdir -p $out; cd $GRAFT_REPO_ROOT
for w in c3 c5; do
  for r in ${ROWS:-auto 8192 16384 19584 20000}; do
    if [ $r = auto ]; then unset SPMV_PB_ROWS; else export SPMV_PB_ROWS=$r; fi
    timeout -k 10 200 python3 bench.py --workload $w --launcher hipSpMVTilesCSR --steps 10 --warmup 2 --no-cpu-baseline --no-extra 2>/dev/null | python3 -c "import sys,json; j=json.loads(sys.stdin.read()); print('$w rows=$r', round(j['roofline']['kernel_ms_avg'],4), 'ms', round(j['roofline']['frac'],4))" || exit 1
  done
done
for n in ${SHARDS:-8 4 2}; do
  for r in ${ROWS:-auto 8192 16384 19584 20000}; do
    if [ $r = auto ]; then unset SPMV_PB_ROWS; else export SPMV_PB_ROWS=$r; fi
    timeout -k 10 200 python3 scripts/shard_shape.py $n 1 2>/dev/null | tail -1 || exit 1
  done
done

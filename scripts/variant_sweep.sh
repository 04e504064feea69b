#!/bin/bash
# one-pass kernel build variants (SPMV_WG_THREADS x SPMV_STREAM_NNZ) on the cache-friendly workloads
# usage: variant_sweep.sh "<workloads>" <lib>...   ("default" = in-tree library)
cd $GRAFT_REPO_ROOT
wls=$1; shift
for w in $wls; do
  for lib in "$@"; do
    if [ "$lib" = default ]; then unset SPMV_LIB; else export SPMV_LIB=$GRAFT_REPO_ROOT/$lib; fi
    timeout -k 10 200 python3 bench.py --workload $w --launcher hipSpMVWarpPerRowCSR --steps 20 --warmup 3 --no-cpu-baseline --no-extra 2>/dev/null | python3 -c "import sys,json; j=json.loads(sys.stdin.read()); print('$w $(basename $lib)', round(j['roofline']['kernel_ms_avg'],4), 'ms', round(j['roofline']['frac'],4), 'parity n/a')" || exit 1
  done
done

#!/bin/bash
# A/B of launcher variants of the one-pass kernels; usage: variant_ab.sh "<workloads>" <launcher> <variant>...
cd $GRAFT_REPO_ROOT
wls=$1; l=$2; shift 2
for w in $wls; do
  for v in "$@"; do
    timeout -k 10 200 python3 bench.py --workload $w --launcher $l --variant $v --steps 20 --warmup 3 --no-cpu-baseline --no-extra 2>/dev/null | python3 -c "import sys,json; j=json.loads(sys.stdin.read()); print('$w $l variant $v', round(j['roofline']['kernel_ms_avg'],4), 'ms', round(j['roofline']['frac'],4))" || exit 1
  done
done

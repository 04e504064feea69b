#!/bin/bash
# per-phase kernel times of the two-phase path on one rank's shard of c5 at N = 1,2,4,8 ranks (measured on one GPU)
# usage: shard_sweep.sh <outdir-name> [N ...]
out=$GRAFT_REPO_ROOT/gpurun_out/${1:-shards}; shift
Ns=${@:-"2 4 8"}
rm -rf "$out"; mkdir -p "$out"
export TMPDIR=/tmp
cd /tmp
for n in $Ns; do
  timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$out/n$n" -- python3 "$GRAFT_REPO_ROOT/scripts/shard_shape.py" $n 1 > "$out/n$n.log" 2>&1 || exit 1
  tail -1 "$out/n$n.log"
  python3 - "$out/n$n" <<'PY'
import sys, glob, csv
for f in glob.glob(sys.argv[1] + "/**/*kernel_stats.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "pb_expand" in r["Name"] or "pb_reduce" in r["Name"]:
            print("   ", r["Name"].split("(")[0][-20:], r["Calls"], "avg us", round(float(r["AverageNs"]) / 1e3, 1), "min us", round(float(r["MinNs"]) / 1e3, 1))
PY
done

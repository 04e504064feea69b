#!/bin/bash
# A/B of libspmvhip builds on the two-phase kernel: per-phase kernel times from rocprofv3 on one box
# usage: ab_tiles.sh <outdir-name> <workload> <lib>...   ("default" = the in-tree library)
set -e
out=$GRAFT_REPO_ROOT/gpurun_out/$1; w=$2; shift 2
rm -rf "$out"; mkdir -p "$out"
export TMPDIR=/tmp
cd /tmp
for lib in "$@"; do
  tag=$(basename "$lib" .so)
  if [ "$lib" = default ]; then unset SPMV_LIB; else export SPMV_LIB=$GRAFT_REPO_ROOT/$lib; fi
  timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$out/$tag" -- python "$GRAFT_REPO_ROOT/bench.py" --workload $w --launcher hipSpMVTilesCSR --steps 8 --warmup 2 --no-cpu-baseline --no-extra > "$out/$tag.log" 2>&1
done
python "$GRAFT_REPO_ROOT/scripts/ab_report.py" "$out"

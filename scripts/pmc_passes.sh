#!/bin/bash
# rocprofv3 --pmc passes (one counter group per run, never together with a trace) over one command; prints the per-kernel
# averages of every counter for kernels whose name contains <substr>.
#   usage: pmc_passes.sh <outdir> <substr> "<group1 counters>" ["<group2 counters>" ...] -- <program> <args...>
out=$1; sub=$2; shift 2
groups=()
while [ "$1" != "--" ]; do groups+=("$1"); shift; done
shift
rm -rf $out; mkdir -p $out; export TMPDIR=/tmp
i=0
for g in "${groups[@]}"; do
  i=$((i+1))
  ( cd /tmp && timeout -k 10 300 rocprofv3 --pmc $g --output-format csv -d $out/p$i -- "$@" > $out/p$i.log 2>&1 ) || echo "pass $i failed: $g"
done
python3 - $out "$sub" <<'PY'
import sys, glob, csv, collections
acc = collections.defaultdict(list)
for f in glob.glob(sys.argv[1] + "/p*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if sys.argv[2] in r["Kernel_Name"]:
            acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
for n in sorted(acc):
    v = acc[n]
    print(f"{n:40s} {sum(v) / len(v):16.1f}   (n={len(v)})")
PY

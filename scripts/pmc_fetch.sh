#!/bin/bash
# FETCH_SIZE (one counter per pass, as scripts/final_profile.sh does: FETCH_SIZE together with WRITE_SIZE hung the run) of the two-phase kernels for one workload; usage: pmc_fetch.sh <tag> <workload>  (env passes through)
out=$GRAFT_REPO_ROOT/gpurun_out/$1; rm -rf $out; mkdir -p $out; export TMPDIR=/tmp; cd /tmp
timeout -k 10 240 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $out -- python3 $GRAFT_REPO_ROOT/bench.py --workload $2 --launcher hipSpMVTilesCSR --steps 4 --warmup 1 --no-cpu-baseline --no-extra > $out/log 2>&1
python3 - $out <<'PY'
import sys, glob, csv, collections
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "pb_expand" in r["Kernel_Name"] or "pb_reduce" in r["Kernel_Name"]:
            acc["pb_expand" if "pb_expand" in r["Kernel_Name"] else "pb_reduce"][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, d in acc.items():
    print(k, {n: round(sum(v) / len(v) * (2048 if n == "FETCH_SIZE" else 1024) / 1e9, 2) for n, v in d.items()}, "GB (fetch x2 corrected)")
PY

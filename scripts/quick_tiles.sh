#!/bin/bash
# quick A/B of the two-phase kernel: c3 and c5, tiles launcher only, kernel trace for the per-phase split
set -e
out=gpurun_out/${1:-quick}
rm -rf "$out"; mkdir -p "$out"
export TMPDIR=/tmp
for w in c3 c5; do
  timeout -k 10 400 python bench.py --workload $w --launcher hipSpMVTilesCSR --steps 10 --warmup 2 --no-cpu-baseline --no-extra > "$out/bench_$w.json" 2> "$out/bench_$w.err"
done
(cd /tmp && timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d "$GRAFT_REPO_ROOT/$out/prof" -o c5 -- python "$GRAFT_REPO_ROOT/bench.py" --workload c5 --launcher hipSpMVTilesCSR --steps 8 --warmup 2 --no-cpu-baseline --no-extra > "$GRAFT_REPO_ROOT/$out/prof.log" 2>&1)
python - "$out" <<'PY'
import sys, json, glob, csv
out = sys.argv[1]
for w in ("c3", "c5"):
    j = json.loads(open(f"{out}/bench_{w}.json").read().strip().splitlines()[-1])
    print(w, "ms", round(j["ms_per_step"], 4), "frac", round(j["roofline"]["frac"], 4), "parity", j.get("parity", {}).get("ok"))
for f in glob.glob(f"{out}/prof/**/*kernel_stats.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if r["Name"].startswith("pb_"):
            print(r["Name"][:40], r["Calls"], r["AverageNs"], r["MinNs"])
PY

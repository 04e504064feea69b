cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/prof4; rm -rf $O; mkdir -p $O; cd $R
for wl in c3n c2; do
rocprofv3 --kernel-trace --stats --output-format csv -d $O/t_$wl -- python3 bench.py --workload $wl --launcher hipSpMVRowsSELL --no-extra --no-cpu-baseline --steps 5 --warmup 1 > $O/t_$wl.log 2>&1
echo "== $wl"; python3 - <<PY
import csv,glob
for r in csv.DictReader(open(glob.glob("$O/t_$wl/*/*_kernel_stats.csv")[0])):
    if "sell" in r["Name"] or "radix" in r["Name"]: print(r["Name"].split("(")[0][-60:], r["Calls"], float(r["AverageNs"])/1e3)
PY
done

cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r02g; rm -rf $O; mkdir -p $O; cd $R
B="python3 bench.py --no-cpu-baseline --no-extra --steps 10 --warmup 2 --workload c5 --launcher hipSpMVTilesCSR"
timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/pmc_fetch -- $B > $O/pmc1.log 2>&1
timeout -k 10 300 rocprofv3 --pmc WRITE_SIZE TCC_HIT_sum TCC_MISS_sum --output-format csv -d $O/pmc_write -- $B > $O/pmc2.log 2>&1
python3 - $O <<'PY'
import sys, glob, csv, collections
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(sys.argv[1] + "/pmc*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        if "pb_" in k: acc[k.split("(")[0][-40:]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, d in acc.items():
    for n, v in d.items():
        print(k, n, round(sum(v) / len(v), 1), "(x2 KiB -> GB: %.2f)" % (sum(v) / len(v) * 2048 / 1e9) if n == "FETCH_SIZE" else "")
PY

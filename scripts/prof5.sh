cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/prof5; rm -rf $O; mkdir -p $O; cd $R
for sg in 256 16384; do
export SPMV_SELL_SIGMA=$sg
B="python3 bench.py --workload c3n --launcher hipSpMVRowsSELL --no-cpu-baseline --no-extra --steps 5 --warmup 1"
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/a$sg -- $B > $O/a.log 2>&1
rocprofv3 --pmc WRITE_SIZE TCC_HIT_sum TCC_MISS_sum --output-format csv -d $O/b$sg -- $B > $O/b.log 2>&1
rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_VMEM_RD --output-format csv -d $O/c$sg -- $B > $O/c.log 2>&1
rocprofv3 --pmc TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_TCC_WRITE_REQ_sum --output-format csv -d $O/d$sg -- $B > $O/d.log 2>&1
echo "== sigma $sg"
python3 - <<PY
import csv,glob,collections
acc=collections.defaultdict(list)
for p in glob.glob("$O/?$sg/*/*_counter_collection.csv"):
    for r in csv.DictReader(open(p)):
        if "sell_spmv" in r["Kernel_Name"]: acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
for k in sorted(acc): print(f"{k:34s} {sum(acc[k])/len(acc[k]):.4g}")
PY
done

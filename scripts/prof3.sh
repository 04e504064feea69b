cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/prof3; rm -rf $O; mkdir -p $O; cd $R
B="python3 bench.py --workload c3n --launcher hipSpMVWarpPerRowCSR --no-cpu-baseline --no-extra --steps 10 --warmup 2"
rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU --output-format csv -d $O/p1 -- $B > $O/p1.log 2>&1
rocprofv3 --pmc SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_SMEM SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS --output-format csv -d $O/p2 -- $B > $O/p2.log 2>&1
rocprofv3 --pmc TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_PENDING_STALL_CYCLES_sum --output-format csv -d $O/p3 -- $B > $O/p3.log 2>&1
rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum --output-format csv -d $O/p4 -- $B > $O/p4.log 2>&1
rocprofv3 --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_FLAT SQ_INSTS_FLAT SQ_WAVE32_INSTS SQ_THREAD_CYCLES_VALU --output-format csv -d $O/p5 -- $B > $O/p5.log 2>&1
python3 - <<PY
import csv,glob,collections
acc=collections.defaultdict(list)
for p in glob.glob("$O/p*/*/*_counter_collection.csv"):
    for r in csv.DictReader(open(p)):
        if "csr_stream2" in r["Kernel_Name"]: acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
for k in sorted(acc): print(f"{k:34s} {sum(acc[k])/len(acc[k]):.4g}")
PY

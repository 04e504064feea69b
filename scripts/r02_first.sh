# round 2, first GPU run of the stripes kernel: parity, times vs the two-phase kernel, kernel trace + L2 counters
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r02a; rm -rf $O; mkdir -p $O; cd $R
timeout -k 10 500 python3 -m pytest tests/test_gpu_parity.py -x -q -k "Stripes or synth_device or degenerate or 64bit or (full_size and c3)" > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -3 $O/pytest.log
timeout -k 10 200 python3 scripts/time_launchers.py c3 hipSpMVTilesCSR hipSpMVStripesCSR --check 2>&1 | grep -v "^\[" | tee $O/c3.log
timeout -k 10 100 python3 scripts/time_launchers.py c2 hipSpMVRowsSELL hipSpMVTilesCSR hipSpMVStripesCSR --check 2>&1 | tee $O/c2.log
timeout -k 10 200 python3 scripts/time_launchers.py c3b hipSpMVWarpPerRowCSR hipSpMVTilesCSR hipSpMVStripesCSR --check 2>&1 | tee $O/c3b.log
B="python3 scripts/time_launchers.py c3 hipSpMVStripesCSR --steps 5"
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -- $B > $O/trace.log 2>&1
timeout -k 10 200 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/pmc_fetch -- $B > $O/pmc1.log 2>&1
timeout -k 10 200 rocprofv3 --pmc WRITE_SIZE TCC_HIT_sum TCC_MISS_sum --output-format csv -d $O/pmc_write -- $B > $O/pmc2.log 2>&1
timeout -k 10 200 rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_LDS_BANK_CONFLICT --output-format csv -d $O/pmc_sq -- $B > $O/pmc3.log 2>&1
python3 - $O <<'PY'
import sys, glob, csv, collections
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(sys.argv[1] + "/pmc*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        if "sb_spmv" in k:
            acc["sb_spmv"][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, d in acc.items():
    for n, v in d.items():
        print(k, n, sum(v) / len(v))
for f in glob.glob(sys.argv[1] + "/trace/**/*kernel_stats.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        print(r["Name"][:60], r["Calls"], r["AverageNs"])
PY

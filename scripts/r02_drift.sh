cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r02b; rm -rf $O; mkdir -p $O; cd $R
timeout -k 10 300 python3 -m pytest tests/test_gpu_parity.py -x -q -k "Stripes or synth_device or degenerate or 64bit" > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -3 $O/pytest.log
SPMV_LIB=$R/spmv_openmp_cuda_amd/lib/libspmvhip_dbg.so timeout -k 10 200 python3 scripts/stripes_drift.py c3 2>&1 | grep -v amdgpu.ids | tee $O/drift_c3.log
timeout -k 10 200 python3 scripts/time_launchers.py c3 hipSpMVTilesCSR hipSpMVStripesCSR --check 2>&1 | grep -v amdgpu.ids | tee $O/c3.log
timeout -k 10 200 python3 scripts/time_launchers.py c3 hipSpMVStripesCSR --scale 0.5 2>&1 | grep -v amdgpu.ids | tee $O/c3half.log
timeout -k 10 200 python3 scripts/time_launchers.py c2 hipSpMVStripesCSR 2>&1 | grep -v amdgpu.ids | tee $O/c2.log
timeout -k 10 200 python3 scripts/time_launchers.py c3b hipSpMVStripesCSR 2>&1 | grep -v amdgpu.ids | tee $O/c3b.log
B="python3 scripts/time_launchers.py c3 hipSpMVStripesCSR --steps 5"
timeout -k 10 200 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/pmc_fetch -- $B > $O/pmc1.log 2>&1
timeout -k 10 200 rocprofv3 --pmc WRITE_SIZE TCC_HIT_sum TCC_MISS_sum --output-format csv -d $O/pmc_write -- $B > $O/pmc2.log 2>&1
python3 - $O <<'PY'
import sys, glob, csv, collections
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(sys.argv[1] + "/pmc*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "sb_spmv" in r["Kernel_Name"]:
            acc["sb_spmv"][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, d in acc.items():
    for n, v in d.items():
        print(k, n, sum(v) / len(v))
PY

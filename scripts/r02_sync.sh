cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r02e; rm -rf $O; mkdir -p $O; cd $R
L=$R/spmv_openmp_cuda_amd/lib
timeout -k 10 300 python3 -m pytest tests/test_gpu_parity.py -x -q -k "Stripes or degenerate" > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -2 $O/pytest.log
run() { timeout -k 10 200 python3 scripts/time_launchers.py $* hipSpMVStripesCSR --check 2>&1 | grep -v amdgpu.ids | cut -c1-150; }
for sw in "0 8192" "1 8192" "1 12288" "1 16384" "1 32768" "1 131072"; do
    set -- $sw; echo "== sync=$1 window=$2"
    SPMV_SB_SYNC=$1 SPMV_SB_WINDOW=$2 run c3
done 2>&1 | tee $O/sync.log
SPMV_SB_SYNC=0 run c3 --scale 0.5; SPMV_SB_SYNC=1 run c3 --scale 0.5
SPMV_SB_SYNC=1 run c3b; SPMV_SB_SYNC=1 run c2
SPMV_SB_SYNC=1 SPMV_SB_WINDOW=16384 SPMV_LIB=$L/libspmvhip_dbg.so timeout -k 10 200 python3 scripts/stripes_drift.py c3 2>&1 | grep -v amdgpu.ids > $O/drift_c3.log; tail -22 $O/drift_c3.log | cut -c1-200

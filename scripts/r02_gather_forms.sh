# x gather of the stripes kernel as global_load with a 64-bit address per lane (default), buffer_load with a 32-bit
# offset (g4) and global_load with a scalar base + 32-bit offset (g5).  Tuning builds: make tunelib TUNE_NAME=g4 TUNE_DEFS=-DSPMV_SB_GATHER=4
cd $GRAFT_REPO_ROOT
for w in c3 c2; do
  for lib in default g4 g5 default g4 g5; do
    if [ "$lib" = default ]; then unset SPMV_LIB; else export SPMV_LIB=$GRAFT_REPO_ROOT/spmv_openmp_cuda_amd/lib/libspmvhip_$lib.so; fi
    echo "== $w $lib"
    timeout -k 10 200 python3 scripts/time_launchers.py $w hipSpMVStripesCSR --check 2>&1 | grep -v amdgpu.ids
  done
done

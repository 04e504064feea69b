cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r02d; mkdir -p $O; cd $R
L=$R/spmv_openmp_cuda_amd/lib
for v in $VARIANTS; do
  echo "== lib_$v"
  for wl in $WLS; do
    SPMV_LIB=$L/libspmvhip_$v.so timeout -k 10 200 python3 scripts/time_launchers.py $wl hipSpMVStripesCSR --check 2>&1 | grep -v amdgpu.ids
  done
done | tee $O/variants_$TAG.log

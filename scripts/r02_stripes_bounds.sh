# What each stage of the stripes kernel costs alone (tuning builds, not shipped):
#   noadd    = entry stream + x gathers, no LDS adds      (make tunelib TUNE_NAME=noadd TUNE_DEFS=-DSPMV_SB_NOADD)
#   nogather = entry stream + LDS adds, no x gathers      (make tunelib TUNE_NAME=nogather TUNE_DEFS=-DSPMV_SB_GATHER=3)
cd $GRAFT_REPO_ROOT
for w in c3 c2 c3b; do
  for lib in default noadd nogather; do
    if [ "$lib" = default ]; then unset SPMV_LIB; else export SPMV_LIB=$GRAFT_REPO_ROOT/spmv_openmp_cuda_amd/lib/libspmvhip_$lib.so; fi
    echo "== $w $lib"
    timeout -k 10 200 python3 scripts/time_launchers.py $w hipSpMVStripesCSR 2>&1 | grep -v amdgpu.ids
  done
done

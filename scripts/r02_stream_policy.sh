# AS/JA streams of the one-pass and ELL kernels: non-temporal (default) vs default cache policy (tuning build splain)
cd $GRAFT_REPO_ROOT
for rep in 1 2; do
for lib in default splain; do
  if [ "$lib" = default ]; then unset SPMV_LIB; else export SPMV_LIB=$GRAFT_REPO_ROOT/spmv_openmp_cuda_amd/lib/libspmvhip_$lib.so; fi
  for w in c3n c3b c2 c3; do
    echo "== $lib $w"
    timeout -k 10 200 python3 scripts/time_launchers.py $w hipSpMVWarpPerRowCSR hipSpMVRowsCSR hipSpMVRowsSELL --steps 20 2>&1 | grep -v amdgpu.ids
  done
  echo "== $lib c4 ELL"; timeout -k 10 300 python3 scripts/config4_ell.py 2>&1 | grep -v amdgpu.ids | grep "^|"
  echo "== $lib c4 ELL band512"; timeout -k 10 300 python3 scripts/config4_ell.py 1.0 512 2>&1 | grep -v amdgpu.ids | grep "^|"
done
done

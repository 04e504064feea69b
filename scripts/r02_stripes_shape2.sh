# stripes kernel: wavefronts per workgroup x steps per batch once more, with the staggered sweeps (SB_SPREAD = 6)
cd $GRAFT_REPO_ROOT
for rep in 1 2; do
for lib in default t192d8 t256d5 t256d7 t320d5; do
  if [ "$lib" = default ]; then unset SPMV_LIB; else export SPMV_LIB=$GRAFT_REPO_ROOT/spmv_openmp_cuda_amd/lib/libspmvhip_$lib.so; fi
  for w in c3 c2; do echo -n "$lib "; timeout -k 10 200 python3 scripts/time_launchers.py $w hipSpMVStripesCSR --steps 60 --check 2>&1 | grep -v amdgpu.ids | cut -c1-110; done
done
done

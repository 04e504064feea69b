# stripes kernel: the workgroups of an XCD start their column sweeps spread over SPMV_SB_SPREAD/1024 of the bin (wrapping around)
cd $GRAFT_REPO_ROOT
for w in c3 c2; do
for sp in 0 8 16 32 64 128 256 1024 0; do
  echo "== $w spread=$sp"
  SPMV_SB_SPREAD=$sp timeout -k 10 200 python3 scripts/time_launchers.py $w hipSpMVStripesCSR --check 2>&1 | grep -v amdgpu.ids
done
done

cd $GRAFT_REPO_ROOT
for rep in 1 2; do
for sp in 0 3 4 6 8; do
  echo "== c3 spread=$sp"
  SPMV_SB_SPREAD=$sp timeout -k 10 200 python3 scripts/time_launchers.py c3 hipSpMVStripesCSR --steps 300 2>&1 | grep -v amdgpu.ids
done
done
for sp in 0 4 6 8; do
  echo "== c4 spread=$sp"
  SPMV_SB_SPREAD=$sp timeout -k 10 200 python3 scripts/time_launchers.py c4 hipSpMVStripesCSR --steps 300 2>&1 | grep -v amdgpu.ids
done

cd $GRAFT_REPO_ROOT
for sp in 0 2 4 6 8 12 16 24 0 8; do
  echo "== c3 spread=$sp"
  SPMV_SB_SPREAD=$sp timeout -k 10 200 python3 scripts/time_launchers.py c3 hipSpMVStripesCSR --steps 60 2>&1 | grep -v amdgpu.ids
done
for w in c3b c4 c2; do for sp in 0 8 16; do
  echo "== $w spread=$sp"
  SPMV_SB_SPREAD=$sp timeout -k 10 200 python3 scripts/time_launchers.py $w hipSpMVStripesCSR --steps 60 2>&1 | grep -v amdgpu.ids
done; done

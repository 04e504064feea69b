cd $GRAFT_REPO_ROOT
for gf in "208 0.75" "208 0.80" "224 0.80" "224 0.85" "192 0.70" "240 0.88"; do
  set -- $gf
  SPMV_SB_GRID=$1 timeout -k 10 200 python3 scripts/concurrency_probe.py c3 $2 2>&1 | grep -v amdgpu.ids
done

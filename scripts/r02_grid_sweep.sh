# stripes kernel with fewer persistent workgroups than CUs: is the gather bound per CU (time ~ rounds) or chip-wide (time ~ constant)?
cd $GRAFT_REPO_ROOT
for g in 256 171 128 103 86 64; do
  echo "== grid=$g"
  SPMV_SB_GRID=$g timeout -k 10 200 python3 scripts/time_launchers.py c3 hipSpMVStripesCSR 2>&1 | grep -v amdgpu.ids
done

cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/exp3; rm -rf $O; mkdir -p $O; cd $R
for dbg in 0 4; do
  SPMV_PB_DBG=$dbg rocprofv3 --kernel-trace --stats --output-format csv -d $O/t$dbg -- python3 bench.py --workload c5 --launcher hipSpMVTilesCSR --no-extra --no-cpu-baseline --steps 5 --warmup 1 > $O/t$dbg.log 2>&1
  echo "dbg=$dbg $(grep -h pb_expand $O/t$dbg/*/*_kernel_stats.csv | sed -E 's/.*\",([0-9]+),([0-9]+),([0-9.]+),.*/avg_ns=\3/')"
done

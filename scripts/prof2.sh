cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/prof2; rm -rf $O; mkdir -p $O; cd $R
for wl in c2 c3 c3b c5; do
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace_$wl -- python3 bench.py --workload $wl --launcher hipSpMVTilesCSR --no-extra --no-cpu-baseline --steps 10 --warmup 2 > $O/trace_$wl.log 2>&1
  echo "== $wl"; grep -E "pb_expand|pb_reduce|radix|pb_gather|pb_rowof|pb_keys|pb_desc" $O/trace_$wl/*/*_kernel_stats.csv | sed -E 's/"void (spmvhip::)?(\(anonymous namespace\)::)?//; s/\(.*\)"//' | cut -d, -f1-4,6,7 | cut -c1-150
done

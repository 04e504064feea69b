mkdir -p gpurun_out
for wl in c3 c3b c2 c2b; do
 for lv in "hipSpMVWarpPerRowCSR 1" "hipSpMVWarpPerRowCSR 0" "hipSpMVRowsCSR 1" "hipSpMVRowsCSR 0"; do
  set -- $lv
  echo "== $wl $1 v$2" >> gpurun_out/exp1.log
  timeout -k 10 200 python bench.py --workload $wl --launcher $1 --variant $2 --no-extra --no-cpu-baseline --steps 10 2>&1 | grep "^\[bench\]" >> gpurun_out/exp1.log
 done
done
cat gpurun_out/exp1.log

mkdir -p gpurun_out; rm -f gpurun_out/exp2.log
for wl in c2 c3 c3b c5; do
 for l in hipSpMVWarpPerRowCSR hipSpMVTilesCSR; do
  echo "== $wl $l" >> gpurun_out/exp2.log
  timeout -k 10 300 python bench.py --workload $wl --launcher $l --no-extra --no-cpu-baseline --steps 10 2>&1 | grep "^\[bench\]\|rror\|libspmvhip" >> gpurun_out/exp2.log
 done
done
cat gpurun_out/exp2.log

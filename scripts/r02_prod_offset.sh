# does the placement of the product workspace relative to the value array matter? (store microbenchmark of round 1: output at +4096 B 9 % slower)
cd $GRAFT_REPO_ROOT
for rep in 1 2 3; do
for off in 0 256 4096 65536 1048576; do
  SPMV_PB_DEBUG_ADDR=1 SPMV_PB_PROD_OFFSET=$off timeout -k 10 300 python3 bench.py --workload c5 --launcher hipSpMVTilesCSR --steps 12 --warmup 3 --no-cpu-baseline --no-extra 2> gpurun_out/_err.txt | python3 -c "import sys,json; j=json.loads(sys.stdin.read()); r=j['roofline']; p=r['kernel_ms_phases']; print('off=$off', round(r['kernel_ms_avg'],3), round(p['pb_expand_kernel'],3), round(p['pb_reduce_kernel'],3))"
  grep "tiles: val" gpurun_out/_err.txt | head -1
done
done

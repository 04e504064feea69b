cd $GRAFT_REPO_ROOT
for rep in 1 2 3 4; do
for nt in 1 0; do
  SPMV_PB_NTSTORE=$nt timeout -k 10 300 python3 bench.py --workload c5 --launcher hipSpMVTilesCSR --steps 12 --warmup 3 --no-cpu-baseline --no-extra 2>/dev/null | python3 -c "import sys,json; j=json.loads(sys.stdin.read()); r=j['roofline']; p=r['kernel_ms_phases']; print('ntstore=$nt', round(r['kernel_ms_avg'],3), round(p['pb_expand_kernel'],3), round(p['pb_reduce_kernel'],3))"
done
done

cd $GRAFT_REPO_ROOT
for rep in 1 2 3 4 5; do
for lib in default xfirst; do
  if [ "$lib" = default ]; then unset SPMV_LIB; else export SPMV_LIB=$GRAFT_REPO_ROOT/spmv_openmp_cuda_amd/lib/libspmvhip_$lib.so; fi
  timeout -k 10 300 python3 bench.py --workload c5 --launcher hipSpMVTilesCSR --steps 12 --warmup 3 --no-cpu-baseline --no-extra 2>/dev/null | python3 -c "import sys,json; j=json.loads(sys.stdin.read()); r=j['roofline']; p=r['kernel_ms_phases']; print('$lib', round(r['kernel_ms_avg'],3), round(p['pb_expand_kernel'],3), round(p['pb_reduce_kernel'],3), j['parity']['ok'])"
done
done

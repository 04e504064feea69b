# cache policy of the two-phase kernel's streams: non-temporal (default so far) vs default policy, per stream
cd $GRAFT_REPO_ROOT
for w in c5 c3; do
for lib in default p2plain p2row p2pr p2p1 all default; do
  if [ "$lib" = default ]; then unset SPMV_LIB; else export SPMV_LIB=$GRAFT_REPO_ROOT/spmv_openmp_cuda_amd/lib/libspmvhip_$lib.so; fi
  echo "== $w $lib"
  timeout -k 10 300 python3 bench.py --workload $w --launcher hipSpMVTilesCSR --steps 15 --warmup 3 --no-cpu-baseline --no-extra 2>/dev/null | python3 -c "import sys,json; j=json.loads(sys.stdin.read()); r=j['roofline']; print(round(r['kernel_ms_avg'],3), r['kernel_ms_phases'], j['parity']['ok'])"
done
done

# phase 2 of the two-phase kernel: do the product lines that straddle two neighbouring bins meet in an XCD's L2 when the
# two bins run on the same XCD (SPMV_P2_XCD) and the products are read with the default cache policy (SPMV_P2_PLAIN_PROD)?
cd $GRAFT_REPO_ROOT
for lib in default p2xcd p2plain p2xcdplain default; do
  if [ "$lib" = default ]; then unset SPMV_LIB; else export SPMV_LIB=$GRAFT_REPO_ROOT/spmv_openmp_cuda_amd/lib/libspmvhip_$lib.so; fi
  echo "== $lib"
  timeout -k 10 300 python3 bench.py --workload c5 --launcher hipSpMVTilesCSR --steps 15 --warmup 3 --no-cpu-baseline --no-extra 2>/dev/null | python3 -c "import sys,json; j=json.loads(sys.stdin.read()); r=j['roofline']; print(round(r['kernel_ms_avg'],3), r['kernel_ms_phases'])"
done

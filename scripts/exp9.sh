for dbg in 0 3; do for sg in 256 16384; do for wl in c3n c3 c2; do
 echo "== dbg $dbg sigma $sg $wl"; SPMV_SELL_DBG=$dbg SPMV_SELL_SIGMA=$sg timeout -k 10 300 python bench.py --workload $wl --launcher hipSpMVRowsSELL --no-extra --no-cpu-baseline --steps 5 2>&1 | grep "^\[bench\]"
done; done; done

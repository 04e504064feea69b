for sk in 0 1; do for sg in 256 16384; do for wl in c3n c3; do
 echo "== skew $sk sigma $sg $wl"; SPMV_SELL_SKEW=$sk SPMV_SELL_SIGMA=$sg timeout -k 10 300 python bench.py --workload $wl --launcher hipSpMVRowsSELL --no-extra --no-cpu-baseline --steps 5 2>&1 | grep "^\[bench\]"
done; done; done

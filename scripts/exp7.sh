for sg in 64 256 1024 4096 16384 65536; do for wl in c3n c3; do
 echo "== sigma $sg $wl"; SPMV_SELL_SIGMA=$sg timeout -k 10 300 python bench.py --workload $wl --launcher hipSpMVRowsSELL --no-extra --no-cpu-baseline --steps 5 2>&1 | grep "^\[bench\]"
done; done

for wl in c3n c3b c3 c2; do for l in hipSpMVWarpPerRowCSR hipSpMVRowsSELL; do
 echo "== $wl $l"; timeout -k 10 300 python bench.py --workload $wl --launcher $l --no-extra --no-cpu-baseline --steps 10 2>&1 | grep "^\[bench\]"
done; done

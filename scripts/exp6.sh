for wl in c3n c3b c3 c2; do for lv in "hipSpMVWarpPerRowCSR 2" "hipSpMVWarpPerRowCSR 3" "hipSpMVRowsCSR 3"; do
 set -- $lv; echo "== $wl $1 v$2"; timeout -k 10 300 python bench.py --workload $wl --launcher $1 --variant $2 --no-extra --no-cpu-baseline --steps 10 2>&1 | grep "^\[bench\]"
done; done

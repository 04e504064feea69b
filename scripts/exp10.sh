for lib in libspmvhip tune_128_1024 tune_128_2048 tune_256_1024 tune_256_4096 tune_512_4096; do for wl in c3n c3b c2b; do
 echo "== $lib $wl"; SPMV_LIB=$GRAFT_REPO_ROOT/spmv_openmp_cuda_amd/lib/$lib.so timeout -k 10 300 python bench.py --workload $wl --launcher hipSpMVWarpPerRowCSR --no-extra --no-cpu-baseline --steps 10 2>&1 | grep "^\[bench\]"
done; done

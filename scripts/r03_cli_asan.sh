#!/bin/bash
# The CLI built whole (host C + the library's host side) under AddressSanitizer + UBSan with ROCm's clang (`make asan`),
# run on the golden matrices in every mode, through the sharded path, on a .gz, and with bad usage.
set -o pipefail
out=gpurun_out/r03_asan; mkdir -p $out
export ASAN_OPTIONS=allocator_may_return_null=1:detect_leaks=0:alloc_dealloc_mismatch=0:new_delete_type_mismatch=0:protect_shadow_gap=0:halt_on_error=1:log_path=$PWD/$out/cli_asan
export UBSAN_OPTIONS=print_stacktrace=1:halt_on_error=1
CLI=spmv_openmp_cuda_amd/bin/SpMV_HIP_asan.elf
G=tests/golden
n=0; bad=0
for m in cage4like int5x7 pattern8x5 rand300 skew12x40 sym6; do
  for mode in CUDA_CSR_ROWS CUDA_CSR_ROWS_WARP HIP_CSR_TILES HIP_CSR_STRIPES CUDA_CSR_AUTO HIP_SELL_ROWS CUDA_ELL_ROWS HIP_ELL_ROWS_NN_TRANSPOSED CUDA_ELL_ROWS_WARP_NN_TRANSPOSED; do
    timeout -k 5 120 $CLI $G/$m.mtx $G/x_$m.bin $mode > $out/cli_$m.$mode.log 2>&1; rc=$?
    n=$((n+1)); if [ $rc -ne 0 ]; then bad=$((bad+1)); echo "FAILED rc=$rc: $m $mode"; tail -5 $out/cli_$m.$mode.log; fi
  done
done
echo "cli under ASan/UBSan: $n runs, $bad failed"
SPMV_NGPU=1 timeout -k 5 120 $CLI $G/rand300.mtx $G/x_rand300.bin CUDA_CSR_ROWS > $out/cli_sharded.log 2>&1; echo "sharded rc=$?"
SPMV_NGPU=1 timeout -k 5 120 $CLI $G/rand300.mtx RNDVECT CUDA_CSR_ROWS_WARP > $out/cli_sharded2.log 2>&1; echo "sharded warp rc=$?"
gzip -c $G/sym6.mtx > /tmp/sym6.mtx.gz; timeout -k 5 120 $CLI /tmp/sym6.mtx.gz $G/x_sym6.bin CUDA_CSR_ROWS > $out/cli_gz.log 2>&1; echo "gz rc=$?"
# usage errors must be refused with a message, not a report
timeout -k 5 60 $CLI > $out/cli_bad1.log 2>&1; echo "no args rc=$?"
timeout -k 5 60 $CLI $G/sym6.mtx RNDVECT NOT_A_MODE > $out/cli_bad2.log 2>&1; echo "bad mode rc=$?"
timeout -k 5 60 $CLI /nonexistent.mtx RNDVECT CUDA_CSR_ROWS > $out/cli_bad3.log 2>&1; echo "no file rc=$?"
timeout -k 5 60 $CLI $G/sym6.mtx $G/x_rand300.bin CUDA_CSR_ROWS > $out/cli_bad4.log 2>&1; echo "wrong vector rc=$?"
ls $out | grep cli_asan; for f in $out/cli_asan.*; do [ -f "$f" ] && head -30 "$f"; done
[ $bad -eq 0 ] && ! ls $out/cli_asan.* > /dev/null 2>&1

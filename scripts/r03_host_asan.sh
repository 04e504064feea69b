#!/bin/bash
# The product library with its HOST code under AddressSanitizer + UBSan (device code untouched: every -fsanitize= sits
# behind -Xarch_host in the build line of `make asan`), loaded through SPMV_LIB by the GPU tests that run in ONE process
# (children started by the multi-process tests die inside libamdhip64's own allocations under the sanitizer runtime;
# the programs of tests/test_gpu_programs.py link the ordinary library).
set -o pipefail
out=gpurun_out/r03_asan; mkdir -p $out
rt=$(ls /opt/rocm/lib/llvm/lib/clang/*/lib/linux/libclang_rt.asan-x86_64.so | head -1)
# the sanitizer's dlopen interceptor does not see the RUNPATH of the library that calls it: torch's own lazy dlopen()s need its directory spelled out
export LD_LIBRARY_PATH=/usr/local/lib/python3.10/dist-packages/torch/lib:$LD_LIBRARY_PATH
export SPMV_LIB=$PWD/spmv_openmp_cuda_amd/lib/libspmvhip_asan.so
export ASAN_OPTIONS=allocator_may_return_null=1:detect_leaks=0:alloc_dealloc_mismatch=0:new_delete_type_mismatch=0:protect_shadow_gap=0:halt_on_error=1:log_path=$PWD/$out/asan
export UBSAN_OPTIONS=print_stacktrace=1:halt_on_error=1
LD_PRELOAD=$rt timeout -k 10 1000 python -m pytest tests/test_gpu_parity.py tests/test_gpu_peer.py tests/test_gpu_programs.py -m gpu -q \
    -k "not shared_gpu and not cli and not harness and not spawns and not one_rank" > $out/pytest.log 2>&1
rc=$?
echo "pytest under host ASan/UBSan rc=$rc"; tail -6 $out/pytest.log
ls $out; for f in $out/asan.*; do [ -f "$f" ] && head -40 "$f"; done
exit $rc

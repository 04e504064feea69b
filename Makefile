# Top-level build.  `make` = GPU library + host programs + checker.
#   lib      spmv_openmp_cuda_amd/lib/libspmvhip.so   (hipcc, gfx950)      -- the product
#   host     spmv_openmp_cuda_amd/lib/libspmvhost.so, bin/SpMV_HIP.elf     (gcc, plain C host side)
#   oracle   oracle/liboracle.so (+ oracle/_ref when /root/reference exists) -- checker only
#   harness  tests/harness/test_SpMV_HIP.elf (links product + oracle; test program)
#   fuzz     tests/harness/fuzz_loader.elf (host loader + decompressors under ASan/UBSan; test program, not in `all`)
HIPCC    ?= hipcc
CC        = gcc
ARCH     ?= gfx950
PKG       = spmv_openmp_cuda_amd
HIPSRC    = $(PKG)/csrc/hip/abi.hip $(PKG)/csrc/hip/synth.hip $(PKG)/csrc/hip/shard.hip $(PKG)/csrc/hip/tiles.hip $(PKG)/csrc/hip/stripes.hip $(PKG)/csrc/hip/sell.hip $(PKG)/csrc/hip/peer.hip
HIPHDR    = $(PKG)/csrc/hip/kernels.hpp $(PKG)/csrc/hip/device_mat.hpp include/spmvHip.h include/spmv_types.h
HIPFLAGS  = --offload-arch=$(ARCH) -O3 -fPIC -shared -std=c++17 -ffp-contract=off -Wall -Wno-unused-function -Iinclude -ldl
HOSTSRC   = $(wildcard $(PKG)/csrc/host/*.c)
HOSTLIBSRC= $(filter-out %/main.c,$(HOSTSRC))
CFLAGS    = -O2 -fopenmp -fPIC -Wall -Wextra -Wno-unused-parameter -Iinclude

all: lib host oracle harness

lib: $(PKG)/lib/libspmvhip.so
$(PKG)/lib/libspmvhip.so: $(HIPSRC) $(HIPHDR)
	mkdir -p $(PKG)/lib
	$(HIPCC) $(HIPFLAGS) -o $@ $(HIPSRC)

host: $(PKG)/lib/libspmvhost.so $(PKG)/bin/SpMV_HIP.elf
$(PKG)/lib/libspmvhost.so: $(HOSTLIBSRC) $(wildcard include/*.h)
	mkdir -p $(PKG)/lib
	$(CC) $(CFLAGS) -shared -o $@ $(HOSTLIBSRC) -lm -lz -ldl
$(PKG)/bin/SpMV_HIP.elf: $(PKG)/csrc/host/main.c $(PKG)/lib/libspmvhost.so $(PKG)/lib/libspmvhip.so
	mkdir -p $(PKG)/bin
	$(CC) $(CFLAGS) -o $@ $(PKG)/csrc/host/main.c -L$(PKG)/lib -lspmvhost -lspmvhip -Wl,-rpath,'$$ORIGIN/../lib' -lm

oracle:
	$(MAKE) -C oracle all

harness: tests/harness/test_SpMV_HIP.elf
tests/harness/test_SpMV_HIP.elf: tests/harness/spmv_test.c oracle $(PKG)/lib/libspmvhost.so $(PKG)/lib/libspmvhip.so
	$(CC) $(CFLAGS) -DAVG_TIMES_ITERATION=25 -o $@ tests/harness/spmv_test.c -L$(PKG)/lib -lspmvhost -lspmvhip -Loracle -loracle \
	    -Wl,-rpath,'$$ORIGIN/../../$(PKG)/lib' -Wl,-rpath,'$$ORIGIN/../../oracle' -lm

# the product library with its HOST side under AddressSanitizer + UBSan (device code untouched); scripts/r03_host_asan.sh
# loads it into the GPU parity tests through SPMV_LIB
asan: $(PKG)/lib/libspmvhip_asan.so $(PKG)/bin/SpMV_HIP_asan.elf
# the CLI built whole under the sanitizers with ROCm's clang (its ASan runtime is the one the library above links; gcc's is another)
CLANG ?= /opt/rocm/lib/llvm/bin/clang
$(PKG)/bin/SpMV_HIP_asan.elf: $(HOSTSRC) $(wildcard include/*.h) $(PKG)/lib/libspmvhip_asan.so
	mkdir -p $(PKG)/bin
	$(CLANG) -O1 -g -fopenmp -fsanitize=address,undefined -fno-omit-frame-pointer -shared-libsan -Wall -Wextra -Wno-unused-parameter -Iinclude \
	    -o $@ $(HOSTSRC) -L$(PKG)/lib -l:libspmvhip_asan.so -Wl,-rpath,'$$ORIGIN/../lib' -Wl,-rpath,/opt/rocm/lib/llvm/lib \
	    -Wl,-rpath,$(dir $(firstword $(wildcard /opt/rocm/lib/llvm/lib/clang/*/lib/linux/libclang_rt.asan-x86_64.so))) -lm -lz -ldl
$(PKG)/lib/libspmvhip_asan.so: $(HIPSRC) $(HIPHDR)
	$(HIPCC) $(HIPFLAGS) -g -Xarch_host -fsanitize=address -Xarch_host -fsanitize=undefined -Xarch_host -fno-omit-frame-pointer \
	    -shared-libsan -o $@ $(HIPSRC)

# the loader and the decompressors under AddressSanitizer + UBSan (host code only; tests/test_host_side.py runs it)
fuzz: tests/harness/fuzz_loader.elf
tests/harness/fuzz_loader.elf: tests/harness/fuzz_loader.c $(HOSTLIBSRC) $(wildcard include/*.h)
	$(CC) -O1 -g -fopenmp -fsanitize=address,undefined -fno-sanitize-recover=undefined -fno-omit-frame-pointer -Wall -Wextra \
	    -Wno-unused-parameter -Iinclude -o $@ tests/harness/fuzz_loader.c $(HOSTLIBSRC) -lm -lz -ldl

clean:
	rm -f $(PKG)/lib/*.so $(PKG)/bin/*.elf tests/harness/*.elf
	$(MAKE) -C oracle clean
.PHONY: all lib host oracle harness fuzz asan clean

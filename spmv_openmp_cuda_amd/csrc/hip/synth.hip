// synth.hip -- on-device generator for the synthetic benchmark matrices
// (DESIGN.md "Synthetic inputs").  The reference has no generator -- its
// matrices come from MatrixMarket files only -- so this is measurement
// infrastructure of this repo; oracle/synth_ref.c is its CPU twin and
// tests/test_synth.py checks the two bit-for-bit.
//
// Counter-based: entry k of global row r depends only on (seed, r, k, len), so a
// row-block shard generated on rank p is identical to the same rows of the
// single-GPU matrix.
#include <hip/hip_runtime.h>
#include "spmvHip.h"
#include "device_mat.hpp"

namespace {

__device__ __forceinline__ uint64_t splitmix64(uint64_t z) {
    z += 0x9E3779B97F4A7C15ull;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}
__device__ __forceinline__ uint64_t mix3(uint64_t seed, uint64_t a, uint64_t b) {
    return splitmix64(splitmix64(seed ^ a) + b);
}

// one wavefront per row, lanes stride the entries
template <typename I>
__global__ __launch_bounds__(256) void synth_fill_kernel(
    uint64_t M, uint64_t N, uint64_t rowOffset, const I* __restrict__ IRP, uint32_t* __restrict__ JA,
    double* __restrict__ AS, uint64_t seedStruct, uint64_t seedVal, uint64_t band) {
    const uint64_t i = ((uint64_t)blockIdx.y * gridDim.x + blockIdx.x) * (blockDim.x / 64) + threadIdx.x / 64;
    if (i >= M) return;
    const uint32_t lane = threadIdx.x % 64;
    const uint64_t base0 = IRP[0];
    const uint64_t b = (uint64_t)IRP[i] - base0, len = (uint64_t)IRP[i + 1] - (uint64_t)IRP[i];
    if (len > N) return;
    const uint64_t r = rowOffset + i;
    uint64_t w0 = 0, S = N;
    if (band) {
        S = 2 * band + 1;
        if (S < 2 * len) S = 2 * len;
        if (S > N) S = N;
        const uint64_t half = S / 2;
        w0 = r > half ? r - half : 0;
        if (w0 + S > N) w0 = N - S;
    }
    for (uint64_t k = lane; k < len; k += 64) {
        const uint64_t lo = (k * S) / len, hi = ((k + 1) * S) / len;
        const uint64_t col = w0 + lo + mix3(seedStruct, r, k) % (hi - lo);
        JA[b + k] = (uint32_t)col;
        AS[b + k] = (double)(mix3(seedVal, r, k) >> 11) * 0x1.0p-52 - 1.0;
    }
}

}  // namespace

extern "C" int spmvHipSynthFillCSR(ulong M, ulong N, ulong rowOffset, const void* dIRP, int irpBytes,
                                   uint32_t* dJA, double* dAS, uint64_t seedStruct, uint64_t seedVal,
                                   ulong band) {
    if (M == 0) return EXIT_SUCCESS;
    if (!dIRP || !dJA || !dAS || (irpBytes != 4 && irpBytes != 8) || N == 0 || N > (1ull << 32)) {
        fprintf(stderr, "libspmvhip: spmvHipSynthFillCSR: bad arguments\n");
        return EXIT_FAILURE;
    }
    const uint64_t blocks = (M + 3) / 4;
    const dim3 grid = spmvhip::grid2d(blocks, 256);     // 2-D when blocks*256 would overflow the 32-bit work-item count
    if (irpBytes == 4)
        hipLaunchKernelGGL((synth_fill_kernel<uint32_t>), grid, dim3(256), 0, nullptr, M, N, rowOffset,
                           static_cast<const uint32_t*>(dIRP), dJA, dAS, seedStruct, seedVal, band);
    else
        hipLaunchKernelGGL((synth_fill_kernel<uint64_t>), grid, dim3(256), 0, nullptr, M, N, rowOffset,
                           static_cast<const uint64_t*>(dIRP), dJA, dAS, seedStruct, seedVal, band);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipDeviceSynchronize());
    return EXIT_SUCCESS;
}

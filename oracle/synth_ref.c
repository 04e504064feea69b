/*
 * oracle/synth_ref.c -- TEST INFRASTRUCTURE: CPU twin of the on-device synthetic
 * matrix generator (spmv_openmp_cuda_amd/csrc/hip/synth.hip).  The reference has
 * no generator (its matrices only come from MatrixMarket files), so there is no
 * reference code to follow here; the recipe is this repo's (DESIGN.md
 * "Synthetic inputs") and the twin exists so that tests can check the device
 * generator bit-for-bit and feed identical matrices to the oracle.
 *
 * Entry k (0-based) of global row r with row length len:
 *   uniform : stratum k of [0,N):  lo=floor(k*N/len), hi=floor((k+1)*N/len)
 *   banded  : the same strata over a window of S=min(N,max(2*band+1,2*len))
 *             columns centred on r and shifted into [0,N)
 *   column  = lo + mix(seedStruct, r, k) mod (hi-lo)      -> ascending, distinct
 *   value   = (mix(seedVal, r, k) >> 11) * 2^-52 - 1      -> U[-1,1), exact in fp64
 */
#include <stdint.h>
#include <stdlib.h>

static inline uint64_t splitmix64(uint64_t z) {
    z += 0x9E3779B97F4A7C15ull;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}
static inline uint64_t mix3(uint64_t seed, uint64_t a, uint64_t b) {
    return splitmix64(splitmix64(seed ^ a) + b);
}

void synthRowRef(uint64_t N, uint64_t r, uint64_t len, uint64_t seedStruct, uint64_t seedVal,
                 uint64_t band, uint32_t* ja, double* as) {
    uint64_t w0 = 0, S = N;
    if (band) {
        S = 2 * band + 1;
        if (S < 2 * len) S = 2 * len;
        if (S > N) S = N;
        uint64_t half = S / 2;
        w0 = r > half ? r - half : 0;
        if (w0 + S > N) w0 = N - S;
    }
    for (uint64_t k = 0; k < len; k++) {
        uint64_t lo = (k * S) / len, hi = ((k + 1) * S) / len;
        uint64_t col = w0 + lo + mix3(seedStruct, r, k) % (hi - lo);
        ja[k] = (uint32_t)col;
        as[k] = (double)(mix3(seedVal, r, k) >> 11) * 0x1.0p-52 - 1.0;
    }
}

/* rows [rowOffset, rowOffset+M) ; IRP local (IRP[0] may be any base; entries are
 * written at IRP[i]-IRP[0]) */
int synthFillCsrRef(uint64_t M, uint64_t N, uint64_t rowOffset, const uint64_t* IRP,
                    uint32_t* JA, double* AS, uint64_t seedStruct, uint64_t seedVal,
                    uint64_t band) {
    const uint64_t base = IRP[0];
    #pragma omp parallel for schedule(dynamic, 1024)
    for (uint64_t i = 0; i < M; i++) {
        uint64_t len = IRP[i + 1] - IRP[i];
        if (len > N) continue;      /* caller error; leaves the row untouched */
        synthRowRef(N, rowOffset + i, len, seedStruct, seedVal, band,
                    JA + (IRP[i] - base), AS + (IRP[i] - base));
    }
    return EXIT_SUCCESS;
}

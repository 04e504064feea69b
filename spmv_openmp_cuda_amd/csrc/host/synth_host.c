/*
 * synth_host.c -- host half of the synthetic benchmark inputs (DESIGN.md
 * "Synthetic inputs"): row-length laws, prefix sums and the dense vector x.
 * The reference has no generator (matrices come from files), so this is this
 * repo's measurement infrastructure; the per-entry half runs on the device
 * (csrc/hip/synth.hip, CPU twin oracle/synth_ref.c).
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#include "spmv_types.h"

static inline uint64_t splitmix64(uint64_t z) {
    z += 0x9E3779B97F4A7C15ull;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}

/* Bijection of [0,N): 4-round balanced Feistel network over the smallest even
 * bit width covering N, cycle-walked back into range.  Scatters the heavy
 * rows of the power-law without a sequential shuffle. */
typedef struct { uint64_t N, seed; unsigned half; uint64_t mask; } perm_t;
static perm_t permInit(uint64_t N, uint64_t seed) {
    unsigned bits = 2;
    while (bits < 64 && (1ull << bits) < N) bits += 2;
    perm_t p = {N, seed, bits / 2, (1ull << (bits / 2)) - 1};
    return p;
}
static inline uint64_t permApply(const perm_t* p, uint64_t v) {
    do {
        uint64_t L = v >> p->half, R = v & p->mask;
        for (unsigned round = 0; round < 4; ++round) {
            const uint64_t F = splitmix64(p->seed + 0x1000003ull * round + R) & p->mask;
            const uint64_t nL = R;
            R = L ^ F;
            L = nL;
        }
        v = (L << p->half) | R;
    } while (v >= p->N);
    return v;
}

static inline uint32_t plLen(uint64_t rank, double s, uint32_t maxRow) {
    const double f = floor((double)maxRow * pow((double)(rank + 1), -s));
    return f < 1.0 ? 1u : (uint32_t)f;
}

/* sum_k max(1, floor(maxRow*(k+1)^-s)) by counting, O(maxRow): monotone in s,
 * used only to bracket s */
static double plSumFast(uint64_t N, double s, uint32_t maxRow) {
    /* #{k in [1,N] : maxRow * k^-s >= v} = min(N, floor((maxRow/v)^(1/s))) */
    double total = (double)N;               /* v = 1 via the max(1,.) floor */
    for (uint32_t v = 2; v <= maxRow; ++v) {
        double cnt = floor(pow((double)maxRow / v, 1.0 / s));
        if (cnt > (double)N) cnt = (double)N;
        if (cnt < 1) break;
        total += cnt;
    }
    return total;
}

/*
 * Power-law row lengths: len(rank) = max(1, floor(maxRow*(rank+1)^-s)),
 * s chosen so that the total is the largest value <= nnz, the remainder added
 * one per row from the lightest rank upward (so the maximum stays maxRow), and
 * row r gets the length of rank perm(r).  Returns EXIT_FAILURE if nnz < N or
 * the law cannot reach nnz.  *sOut receives the exponent.
 */
int spmvSynthPowerLawLengths(uint64_t N, uint64_t nnz, uint32_t maxRow, uint64_t seed,
                             uint32_t* len, double* sOut) {
    if (N == 0 || nnz < N || maxRow == 0 || (double)nnz > (double)N * maxRow) return EXIT_FAILURE;
    double lo = 1e-3, hi = 8.0;             /* plSumFast decreases with s */
    if (plSumFast(N, lo, maxRow) < (double)nnz) lo = 1e-9;
    for (int it = 0; it < 80; ++it) {
        const double mid = 0.5 * (lo + hi);
        if (plSumFast(N, mid, maxRow) > (double)nnz) lo = mid; else hi = mid;
    }
    double s = hi;
    uint64_t total = 0;
    uint32_t* byRank = malloc(N * sizeof *byRank);
    if (!byRank) return EXIT_FAILURE;
    for (int attempt = 0; attempt < 64; ++attempt) {
        total = 0;
        #pragma omp parallel for reduction(+ : total) schedule(static)
        for (uint64_t k = 0; k < N; ++k) { byRank[k] = plLen(k, s, maxRow); total += byRank[k]; }
        if (total <= nnz) break;
        s *= 1.0 + 1e-6;                    /* exact sum can exceed the counted estimate by rounding */
    }
    if (total > nnz) { free(byRank); return EXIT_FAILURE; }
    uint64_t rem = nnz - total;
    while (rem) {                           /* lightest ranks first, never above maxRow */
        uint64_t added = 0;
        for (uint64_t k = N; k-- > 0 && rem;) {
            if (byRank[k] < maxRow) { byRank[k]++; rem--; added++; }
        }
        if (!added) { free(byRank); return EXIT_FAILURE; }
    }
    const perm_t p = permInit(N, seed);
    #pragma omp parallel for schedule(static)
    for (uint64_t r = 0; r < N; ++r) len[r] = byRank[permApply(&p, r)];
    free(byRank);
    if (sOut) *sOut = s;
    return EXIT_SUCCESS;
}

/* IRP[0]=0, IRP[i+1]=IRP[i]+len[i]; returns nnz */
uint64_t spmvSynthPrefix(const uint32_t* len, uint64_t N, uint64_t* IRP) {
    uint64_t acc = 0;
    IRP[0] = 0;
    for (uint64_t i = 0; i < N; ++i) { acc += len[i]; IRP[i + 1] = acc; }
    return acc;
}

/* x_i = sin(2*pi*u_i) * MAXRND, u_i in [0,1) from the counter hash: same
 * envelope as the reference's fillRndVector (utils.c:322-329,351-359) but
 * finite and reproducible */
void spmvSynthMakeX(uint64_t n, uint64_t seed, double* x) {
    #pragma omp parallel for schedule(static)
    for (uint64_t i = 0; i < n; ++i) {
        const double u = (double)(splitmix64(seed ^ splitmix64(i)) >> 11) * 0x1.0p-53;
        x[i] = sin(6.283185307179586476925 * u) * MAXRND;
    }
}

/* permutation probe for the tests */
uint64_t spmvSynthPerm(uint64_t N, uint64_t seed, uint64_t v) {
    const perm_t p = permInit(N, seed);
    return permApply(&p, v);
}

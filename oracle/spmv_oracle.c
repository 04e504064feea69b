/*
 * oracle/spmv_oracle.c -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 *
 * CPU restatement of the reference's serial oracle and of its two OpenMP
 * row-parallel SpMV baselines, written from the algorithm (nothing copied):
 *   sgemvSerial       follows  src/SpMV_CSR_OMP.c:229-250
 *   spmvRowsBasicCSR  follows  src/SpMV_CSR_OMP.c:34-63
 *   spmvRowsBasicELL  follows  src/SpMV_ELL_OMP.c:33-67
 *   chunksNOOP / chunksFair / chunksFairFolded   follow src/include/ompChunksDivide.h:33-91
 *   ompGetRuntimeSchedule                        follows src/commons/ompGetICV.c:23-50
 *   oracleVectorsDiffRef                         follows src/commons/utils.c:362-393
 *                                                (incl. its "NaN passes" behaviour)
 *
 * (the reference's other five OpenMP variants are restated in spmv_oracle_tiles.c)
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
 * build, link, load or call this file.  Nothing under spmv_openmp_cuda_amd/
 * does; the GPU library has no CPU fallback.
 *
 * Parity pin: tests/test_oracle.py checks these functions bit-for-bit against
 * (a) the committed golden vectors in tests/golden/ that were produced by the
 * reference's own CLI built from /root/reference (oracle/Makefile target
 * `ref`, script tests/golden/make_golden.py) and (b) oracle/_ref/libspmvref.so
 * (the reference's own C sources compiled where they lie) whenever that file
 * is present.
 *
 * Build: gcc -O2 -fopenmp -ffp-contract=off (the reference builds with
 * gcc -O2 -fopenmp on x86-64, test/Makefile:19-24: no FMA contraction there).
 */
#include <assert.h>
#include <math.h>
#include <omp.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "spmv_types.h"      /* spmat, CONFIG, tolerances: the only product header the oracle sees */

/* audit globals are owned by the driver program in the reference (config.h:112,
 * main.cu:56).  The oracle library is loaded stand-alone by the tests, so it
 * carries weak definitions a driver may override. */
__attribute__((weak)) double Start, End, Elapsed, ElapsedInternal;

typedef void (*CHUNKS_DISTR_INTERF)(ulong, spmat*, CONFIG*);

/* ---- schedule(runtime) chunk adaptation ---------------------------------- */
void chunksNOOP(ulong r, spmat* mat, CONFIG* cfg) { (void)r; (void)mat; (void)cfg; }

static void adaptDynamicChunk(ulong iters, CONFIG* cfg, unsigned folding) {
    assert(cfg->threadNum > 0);
    omp_sched_t kind;
    int chunk;
    omp_get_schedule(&kind, &chunk);
    omp_sched_t base = kind;
#if _OPENMP >= 201811
    if (kind & omp_sched_monotonic) base = (omp_sched_t)(kind - omp_sched_monotonic);
#endif
    if (base != omp_sched_dynamic) return;      /* static is already fair; guided/auto untouched... */
    ulong want = iters / ((ulong)cfg->threadNum * folding);
    int chunkNew = (int)(want > 0 ? want : 1);
    if (chunkNew != chunk) omp_set_schedule(kind, chunkNew);
}
/* ...except that the reference falls through its switch for guided/auto with
 * chunk_size_new == 0 and then sets chunk 0 (= implementation default) unless
 * it already was 0.  Reproduced for completeness. */
static void adaptOtherKinds(CONFIG* cfg) {
    (void)cfg;
    omp_sched_t kind;
    int chunk;
    omp_get_schedule(&kind, &chunk);
    omp_sched_t base = kind;
#if _OPENMP >= 201811
    if (kind & omp_sched_monotonic) base = (omp_sched_t)(kind - omp_sched_monotonic);
#endif
    if ((base == omp_sched_guided || base == omp_sched_auto) && chunk != 0)
        omp_set_schedule(kind, 0);
}
void chunksFair(ulong r, spmat* mat, CONFIG* cfg) {
    (void)mat;
    adaptDynamicChunk(r, cfg, 1);
    adaptOtherKinds(cfg);
}
void chunksFairFolded(ulong r, spmat* mat, CONFIG* cfg) {
    (void)mat;
    adaptDynamicChunk(r, cfg, FAIR_CHUNKS_FOLDING);
    adaptOtherKinds(cfg);
}

static const char* const SCHED_NAMES[] = {"OMP_SCHED_STATIC", "OMP_SCHED_DYNAMIC",
                                          "OMP_SCHED_GUIDED", "OMP_SCHED_AUTO"};
void ompGetRuntimeSchedule(int* kindChunkMonotonic) {
    omp_sched_t kind;
    int chunk, monotonic = 0;
    omp_get_schedule(&kind, &chunk);
    omp_sched_t base = kind;
#if _OPENMP >= 201811
    monotonic = (kind & omp_sched_monotonic) != 0;
    if (monotonic) base = (omp_sched_t)(kind - omp_sched_monotonic);
#endif
    printf("omp sched gather:\tkind: %s\tomp chunkSize: %d\tmonotonic: %s\n",
           SCHED_NAMES[base - 1], chunk, monotonic ? "Y" : "N");
    if (kindChunkMonotonic) {
        kindChunkMonotonic[0] = (int)base;
        kindChunkMonotonic[1] = chunk;
        kindChunkMonotonic[2] = monotonic;
    }
}

/* ---- the oracle ---------------------------------------------------------- */
int sgemvSerial(spmat* mat, double* vect, CONFIG* cfg, double* outVect) {
    (void)cfg;
    for (ulong i = 0; i < mat->M; i++) {
        double acc = 0;
        for (ulong j = mat->IRP[i]; j < mat->IRP[i + 1]; j++)
            acc += mat->AS[j] * vect[mat->JA[j]];
        outVect[i] = acc;
    }
    return EXIT_SUCCESS;
}

/* ---- OpenMP baselines ---------------------------------------------------- */
int spmvRowsBasicCSR(spmat* mat, double* vect, CONFIG* cfg, double* outVect) {
    if (cfg->chunkDistrbFunc) ((CHUNKS_DISTR_INTERF)cfg->chunkDistrbFunc)(mat->M, mat, cfg);
    AUDIT_INTERNAL_TIMES Start = omp_get_wtime();
    const ulong  M = mat->M;
    const ulong* IRP = mat->IRP;
    const ulong* JA = mat->JA;
    const double* AS = mat->AS;
    #pragma omp parallel for schedule(runtime)
    for (ulong r = 0; r < M; r++) {
        double acc = 0;
#if SIMD_ROWS_REDUCTION == TRUE
        #pragma omp simd reduction(+ : acc)
#endif
        for (ulong j = IRP[r]; j < IRP[r + 1]; j++)
            acc += AS[j] * vect[JA[j]];
        outVect[r] = acc;
    }
    AUDIT_INTERNAL_TIMES {
        End = omp_get_wtime();
        ElapsedInternal = End - Start;
    }
    return EXIT_SUCCESS;
}

int spmvRowsBasicELL(spmat* mat, double* vect, CONFIG* cfg, double* outVect) {
    if (cfg->chunkDistrbFunc) ((CHUNKS_DISTR_INTERF)cfg->chunkDistrbFunc)(mat->M, mat, cfg);
    AUDIT_INTERNAL_TIMES Start = omp_get_wtime();
    const ulong  M = mat->M, K = mat->MAX_ROW_NZ;
    const ulong* JA = mat->JA;
    const ulong* RL = mat->RL;      /* NULL = walk the padding too (no -DROWLENS) */
    const double* AS = mat->AS;
    #pragma omp parallel for schedule(runtime)
    for (ulong r = 0; r < M; r++) {
        double acc = 0;
        const ulong first = IDX2D(r, 0, K);
        const ulong last  = first + (RL ? RL[r] : K);
#if SIMD_ROWS_REDUCTION == TRUE
        #pragma omp simd reduction(+ : acc)
#endif
        for (ulong j = first; j < last; j++)
            acc += AS[j] * vect[JA[j]];
        outVect[r] = acc;
    }
    AUDIT_INTERNAL_TIMES {
        End = omp_get_wtime();
        ElapsedInternal = End - Start;
    }
    return EXIT_SUCCESS;
}

/* ---- the reference's own gate, restated with its NaN blind spot ------------
 * (the product's doubleVectorsDiff in csrc/host/utils.c fails on NaN instead) */
int oracleVectorsDiffRef(const double* a, const double* b, ulong n, double* diffMax) {
    int out = EXIT_SUCCESS;
    double worst = 0;
    for (ulong i = 0; i < n; i++) {
        double d = a[i] - b[i];
        double ad = ABS(d);
        if (ad > DOUBLE_DIFF_THREASH) out = EXIT_FAILURE;
        if (ABS(worst) < ad) worst = d;
    }
    if (diffMax) *diffMax = worst;
    return out;
}

/* tight, scale-aware check reported next to the 7e-4 gate:
 * returns max_i |a_i-b_i| / (eps-free) sum_j |A_ij x_j| ; rows with zero scale
 * must match exactly (returns INFINITY otherwise). CSR only. */
double oracleScaledError(const spmat* mat, const double* x, const double* yRef, const double* y) {
    double worst = 0;
    for (ulong i = 0; i < mat->M; i++) {
        double scale = 0;
        for (ulong j = mat->IRP[i]; j < mat->IRP[i + 1]; j++)
            scale += fabs(mat->AS[j] * x[mat->JA[j]]);
        double d = fabs(yRef[i] - y[i]);
        if (isnan(y[i])) return INFINITY;
        if (scale == 0) { if (d != 0) return INFINITY; continue; }
        if (d / scale > worst) worst = d / scale;
    }
    return worst;
}

/* flat-array conveniences for ctypes callers (no struct marshalling) */
int oracleCsrSerial(ulong M, const ulong* IRP, const ulong* JA, const double* AS,
                    const double* x, double* y) {
    spmat m;
    memset(&m, 0, sizeof m);
    m.M = M; m.IRP = (ulong*)IRP; m.JA = (ulong*)JA; m.AS = (double*)AS;
    return sgemvSerial(&m, (double*)x, NULL, y);
}
/* same walk with 32-bit device-format indices (for synthetic matrices that are
 * generated directly in device format and never widened) */
int oracleCsrSerial32(ulong M, const uint32_t* IRP, const uint32_t* JA, const double* AS,
                      const double* x, double* y) {
    for (ulong i = 0; i < M; i++) {
        double acc = 0;
        for (uint32_t j = IRP[i]; j < IRP[i + 1]; j++) acc += AS[j] * x[JA[j]];
        y[i] = acc;
    }
    return EXIT_SUCCESS;
}
int oracleCsrSerial64_32(ulong M, const uint64_t* IRP, const uint32_t* JA, const double* AS,
                         const double* x, double* y) {
    for (ulong i = 0; i < M; i++) {
        double acc = 0;
        for (uint64_t j = IRP[i]; j < IRP[i + 1]; j++) acc += AS[j] * x[JA[j]];
        y[i] = acc;
    }
    return EXIT_SUCCESS;
}
/* OpenMP row-parallel walk on device-format arrays: the loop nest of
 * spmvRowsBasicCSR on 32-bit indices (used as the "port" CPU baseline when the
 * full-size matrix only exists in device format) */
int oracleCsrOmp32(ulong M, const uint32_t* IRP, const uint32_t* JA, const double* AS,
                   const double* x, double* y) {
    #pragma omp parallel for schedule(runtime)
    for (ulong r = 0; r < M; r++) {
        double acc = 0;
#if SIMD_ROWS_REDUCTION == TRUE
        #pragma omp simd reduction(+ : acc)
#endif
        for (uint32_t j = IRP[r]; j < IRP[r + 1]; j++) acc += AS[j] * x[JA[j]];
        y[r] = acc;
    }
    return EXIT_SUCCESS;
}
int oracleMaxThreads(void) { return omp_get_max_threads(); }
void oracleSetSchedule(int kind, int chunk) { omp_set_schedule((omp_sched_t)kind, chunk); }
size_t oracleSizeofSpmat(void) { return sizeof(spmat); }
size_t oracleSizeofConfig(void) { return sizeof(CONFIG); }

/* ---- CPU-baseline helpers (bench.py's cpu_baseline leg only) -------------------------------------------
 * First touch.  The reference allocates and fills its matrix from ONE thread (malloc + serial fill in
 * src/lib/parser.c:318-326), so on a NUMA host every page of AS/JA sits on the loading thread's node.
 * SURVEY 8d asks for both figures: as the reference does it, and with the pages first touched by the threads
 * that will read them.  oracleFirstTouchCsr copies a CSR into freshly allocated (still untouched) arrays inside
 * the same static row partition spmvRowsBasicCSR runs with under OMP_SCHEDULE=static. */
void oracleFirstTouchCsr(ulong M, const ulong* IRP, const ulong* jaSrc, const double* asSrc,
                         ulong* irpDst, ulong* jaDst, double* asDst, double* y) {
    #pragma omp parallel for schedule(static)
    for (ulong r = 0; r < M; r++) {
        irpDst[r] = IRP[r];
        for (ulong j = IRP[r]; j < IRP[r + 1]; j++) { jaDst[j] = jaSrc[j]; asDst[j] = asSrc[j]; }
        y[r] = 0.0;
    }
    irpDst[M] = IRP[M];
}
/* what libgomp actually bound: number of places and the binding policy (0 false, 1 true, 2 master, 3 close, 4 spread) */
int oracleOmpPlaces(void) { return omp_get_num_places(); }
int oracleOmpProcBind(void) { return (int)omp_get_proc_bind(); }

/* ---- sanity figures beside the CPU baseline (bench.py) --------------------------------------
 * How many threads a parallel region REALLY gets (affinity masks, cgroup quotas and OMP_NUM_THREADS all have a say),
 * how many distinct CPUs they sit on, and what the same static row partition streams when no x is gathered: the row
 * sums of AS (8 B per entry + 16 B per row).  A CSR SpMV that is far below this figure is bound by its gather, one that
 * is near it by memory bandwidth; a stream figure far below the machine's says the threads share too few cores. */
int oracleThreadsInRegion(void) {
    int n = 0;
    #pragma omp parallel
    {
        #pragma omp single
        n = omp_get_num_threads();
    }
    return n;
}
#ifndef _GNU_SOURCE
#define _GNU_SOURCE
#endif
#include <sched.h>
extern int sched_getcpu(void);
int oracleDistinctCpusInRegion(void) {
    enum { MAXCPU = 4096 };
    static unsigned char seen[MAXCPU];
    for (int i = 0; i < MAXCPU; ++i) seen[i] = 0;
    #pragma omp parallel
    {
        volatile double spin = 0;                      /* long enough for the scheduler to spread the threads */
        for (int i = 0; i < 2000000; ++i) spin += i * 1e-9;
        const int c = sched_getcpu();
        if (c >= 0 && c < MAXCPU) seen[c] = 1;
    }
    int n = 0;
    for (int i = 0; i < MAXCPU; ++i) n += seen[i];
    return n;
}
double oracleRowSumSeconds(ulong M, const ulong* IRP, const double* AS, double* y) {
    const double t0 = omp_get_wtime();
    #pragma omp parallel for schedule(static)
    for (ulong r = 0; r < M; r++) {
        double acc = 0;
        for (ulong j = IRP[r]; j < IRP[r + 1]; j++) acc += AS[j];
        y[r] = acc;
    }
    return omp_get_wtime() - t0;
}

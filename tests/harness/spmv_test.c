/*
 * tests/harness/spmv_test.c -- TEST PROGRAM (links the product AND the oracle).
 * Plain-C counterpart of the reference's check + timing harness
 * (test/SpMV_test.cu:152-389): every implementation's y is compared with the
 * serial oracle under the 7e-4 gate, timed AVG_TIMES_ITERATION times, and
 * reported in the reference's stdout grammar so scripts/parseLog.py-style
 * tooling keeps parsing it:
 *   "#<path>" / "SpMV_OMP_test.c\tAVG_TIMES_ITERATION:%d\tsparse matrix: MxN-NNZ-maxRow=MAX_ROW_NZ"
 *   "@computing SpMV   with func: CUDA CSR %u at:%p"      + "cudaBlockSize: x y z\tcudaGridSize: x y z\t\ttimeAvg:.. timeVar:..\ttimeInternalAvg:.. timeInternalVar:.."
 *   "@computing SpMV   with func: OMP CSR %u at:%p"       + "threadNum: %d\tompGridSize: %ux%u\ttimeAvg:.. ..."
 * plus one extra "#perf ..." line per implementation with GFLOP/s, GB/s and the
 * HBM-roofline fraction (lines starting with '#' are comments to that grammar).
 *
 * What is different from the reference harness on purpose:
 *   - y on the device is poisoned with a NaN pattern before EVERY launch
 *     (the reference never clears it, test/SpMV_test.cu:125, which hides its
 *     rows-0..31-only warp kernels);
 *   - NaN fails the gate;
 *   - each kernel gets the launch shape that belongs to it.
 *
 *   usage: test_SpMV_HIP.elf <matrix.mtx> <vectorFile|RNDVECT> [CUDA_ONLY|OMP_ONLY]
 */
#include <math.h>
#include <omp.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#define SPMV_WITH_OMP_TABLES
#include "SpMV.h"
#include "parser.h"
#include "sparseMatrix.h"
#include "utils.h"

double Start, End, Elapsed, ElapsedInternal;
CONFIG Conf = {.gridRows = 8, .gridCols = 8};

/* oracle-side helpers (oracle/spmv_oracle.c) */
void chunksNOOP(ulong, spmat*, CONFIG*);
void chunksFair(ulong, spmat*, CONFIG*);
void ompGetRuntimeSchedule(int*);
double oracleScaledError(const spmat*, const double*, const double*, const double*);

#define RNDVECT "RNDVECT"
#define POISON 0x7FF8DEADDEADDEADull

static void perfLine(const char* tag, double bytes, ulong nnz, double seconds) {
    printf("#perf %s\tseconds:%le\tGFLOPS:%lf\tGBps:%lf\trooflineFrac:%lf\n", tag, seconds,
           2.0 * nnz / seconds * 1e-9, bytes / seconds * 1e-9, bytes / seconds / 8e12);
}

/* DECREASE_THREAD_NUM (a compile-time macro in the reference, test/SpMV_test.cu:73-78,97-99; an environment variable
 * here): repeat the measurement with omp_get_max_threads(), ..., 2, 1 threads, one "threadNum:" line each. */
static int testSpMVImplOMP(SPMV_INTERF f, spmat* mat, double* vector, double* outV, double* oracleOut) {
    double times[AVG_TIMES_ITERATION], timesInternal[AVG_TIMES_ITERATION], st[2], sti[2];
    const char* sweep = getenv("DECREASE_THREAD_NUM");
    const int tMax = (int)Conf.threadNum, tMin = (sweep && *sweep && *sweep != '0') ? 1 : tMax;
    int rc = EXIT_SUCCESS;
    for (int t = tMax; t >= tMin && rc == EXIT_SUCCESS; t--) {
        omp_set_num_threads(t);
        Conf.threadNum = (uint)t;
        for (uint i = 0; i < AVG_TIMES_ITERATION; i++) {
            for (ulong r = 0; r < mat->M; ++r) outV[r] = NAN;
            double start = omp_get_wtime();
            if (f(mat, vector, &Conf, outV)) { ERRPRINTS("compute func at:%p failed...\n", (void*)f); rc = EXIT_FAILURE; break; }
            double end = omp_get_wtime();
            if (doubleVectorsDiff(oracleOut, outV, mat->M, NULL)) { rc = EXIT_FAILURE; break; }
            times[i] = end - start;
            timesInternal[i] = ElapsedInternal;
            ElapsedInternal = Elapsed = 0;
        }
        if (rc) break;
        statsAvgVar(times, AVG_TIMES_ITERATION, st);
        statsAvgVar(timesInternal, AVG_TIMES_ITERATION, sti);
        printf("threadNum: %d\tompGridSize: %ux%u\ttimeAvg:%le timeVar:%le\ttimeInternalAvg:%le timeInternalVar:%le \n",
               t, Conf.gridRows, Conf.gridCols, st[0], st[1], sti[0], sti[1]);
    }
    omp_set_num_threads(tMax);
    Conf.threadNum = (uint)tMax;
    return rc;
}

static int testSpMVImplHip(SPMV_HIP_INTERF f, spmat* dMat, ulong rows, double* dVect, double* dOutV,
                           double* hOutV, double* oracleOut, double* avgSeconds) {
    double times[AVG_TIMES_ITERATION], st[2];
    for (uint i = 0; i < AVG_TIMES_ITERATION; i++) {
        if (spmvHipVecFill(dOutV, rows, POISON)) return EXIT_FAILURE;
        if (f(dMat, dVect, Conf, dOutV)) return EXIT_FAILURE;
        Elapsed = ElapsedInternal = spmvHipLastKernelSeconds();
        if (spmvHipVecDown(hOutV, dOutV, rows)) return EXIT_FAILURE;
        if (doubleVectorsDiff(oracleOut, hOutV, rows, NULL)) return EXIT_FAILURE;
        times[i] = Elapsed;
        ElapsedInternal = Elapsed = 0;
    }
    statsAvgVar(times, AVG_TIMES_ITERATION, st);
    spmvDim3 g, b;
    spmvHipLastLaunch(&g, &b);
    printf("cudaBlockSize: %u %u %u\tcudaGridSize: %u %u %u\t\ttimeAvg:%le timeVar:%le\ttimeInternalAvg:%le timeInternalVar:%le \n",
           b.x, b.y, b.z, g.x, g.y, g.z, st[0], st[1], st[0], st[1]);
    *avgSeconds = st[0];
    return EXIT_SUCCESS;
}

int main(int argc, char** argv) {
    int out = EXIT_FAILURE;
    if (argc < 3) { ERRPRINT("usage: MatrixMarket_sparse_matrix_COO, vectorFile || " RNDVECT " [CUDA_ONLY|OMP_ONLY]\n"); return out; }
    const int cudaOnly = argc > 3 && !strcmp(argv[3], "CUDA_ONLY");
    const int ompOnly  = argc > 3 && !strcmp(argv[3], "OMP_ONLY");

    double *vector = NULL, *outV = NULL, *oracleOut = NULL, *dVect = NULL, *dOutV = NULL;
    spmat *matCSR = NULL, *matELL = NULL, *matELL_t = NULL;
    spmat dMat;
    memset(&dMat, 0, sizeof dMat);

    char* trgtMatrix = TMP_EXTRACTED_MARTIX;            /* compressed input -> scratch copy (SpMV_test.cu:171-172) */
    {
        const int ex = extractInTmpFS(argv[1], TMP_EXTRACTED_MARTIX);
        if (ex < 0) trgtMatrix = argv[1];
        else if (ex > 0) return out;
    }
    if (!(matCSR = MMtoCSR(trgtMatrix))) return out;
    if (!(matELL = MMtoELL(trgtMatrix))) ERRPRINTS("ELL not feasible for %s:(", argv[1]);
    spmat* mat = matCSR;
    ulong vectSize = mat->N;
    if (!strcmp(argv[2], RNDVECT)) {
        if (!(vector = malloc((vectSize ? vectSize : 1) * sizeof *vector))) goto _free;
        if (init_urndfd() || fillRndVector(vectSize, vector)) { ERRPRINT("fillRndVector errd\n"); goto _free; }
    } else {
        if (!(vector = readDoubleVector(argv[2], &vectSize))) goto _free;
        if (vectSize != mat->N) { ERRPRINT("vector not compatible with sparse matrix\n"); goto _free; }
    }
    if (!(outV = malloc((mat->M ? mat->M : 1) * sizeof *outV))) goto _free;
    if (!(oracleOut = malloc((mat->M ? mat->M : 1) * sizeof *oracleOut))) goto _free;
    sgemvSerial(mat, vector, &Conf, oracleOut);
    for (ulong i = 0; i < mat->M; ++i)
        if (isnan(oracleOut[i])) { ERRPRINT("oracle produced NaN: input vector or matrix holds NaN\n"); goto _free; }

    getConfig(&Conf);
    printf("#%s\n", argv[1]);
    printf("SpMV_OMP_test.c\tAVG_TIMES_ITERATION:%d\tsparse matrix: %lux%lu-%luNNZ-%ld=MAX_ROW_NZ\n",
           AVG_TIMES_ITERATION, mat->M, mat->N, mat->NZ, matELL ? (long)matELL->MAX_ROW_NZ : 0);
    Conf.threadNum = (uint)omp_get_max_threads();
    int sched[3];
    ompGetRuntimeSchedule(sched);
    Conf.chunkDistrbFunc = (void*)chunksNOOP;
    if (sched[0] != omp_sched_static) Conf.chunkDistrbFunc = (void*)chunksFair;

    const double bytesCsr = (double)mat->NZ * 12 + (double)mat->M * 12 + (double)mat->N * 8;

    if (!ompOnly) {
        if (spmvHipInit(getenv("SPMV_DEVICE") ? atoi(getenv("SPMV_DEVICE")) : 0, sizeof(spmat), sizeof(CONFIG))) goto _free;
        if (spmvHipVecAlloc(&dVect, mat->N) || spmvHipVecAlloc(&dOutV, mat->M)) goto _free;
        if (spmvHipVecUp(dVect, vector, mat->N)) goto _free;
        double avg;
        /* ---- CSR */
        if (spMatCpyCSR(matCSR, &dMat)) goto _free;
        for (uint f = 0; f < STATIC_ARR_ELEMENTS_N(SpmvCUDA_CSRFuncs); f++) {
            hprintsf("@computing SpMV   with func: CUDA CSR %u at:%p\n", f, (void*)SpmvCUDA_CSRFuncs[f]);
            if (testSpMVImplHip(SpmvCUDA_CSRFuncs[f], &dMat, mat->M, dVect, dOutV, outV, oracleOut, &avg)) goto _free;
            printf("#tight CSR %u\tmax|dy|/sum|a x|:%le\n", f, oracleScaledError(matCSR, vector, oracleOut, outV));
            if (f == 0 || f == SpmvCUDA_CSRFuncs_WarpPerRowIdx || f == SpmvCUDA_CSRFuncs_AutoIdx) {   /* which kernel the name resolved to */
                double ms3[4] = {0, 0, 0, 0};
                const char* pick = f == 0 ? spmvHipAutoChoiceRows(&dMat, ms3) : spmvHipAutoChoice(&dMat, ms3);
                printf("#auto CSR %u\tpick:%s\tmsStream:%le msTiles:%le msStripes:%le msStripesOrdered:%le\n", f, pick ? pick : "(none)", ms3[0], ms3[1], ms3[2], ms3[3]);
            }
            char tagc[32];
            snprintf(tagc, sizeof tagc, "HIP CSR %u", f);
            perfLine(tagc, bytesCsr, mat->NZ, avg);
        }
        hipFreeSpmat(&dMat);
        /* ---- ELL */
        if (matELL) {
            const double bytesEll = (double)matELL->M * matELL->MAX_ROW_NZ * 12 + (double)mat->M * 8 + (double)mat->N * 8;
            if (!(matELL_t = ellTranspose(matELL))) goto _free;
            if (spMatCpyELL(matELL_t, &dMat)) goto _free;
            for (uint f = 0; f < STATIC_ARR_ELEMENTS_N(SpmvCUDA_ELLFuncs); f++) {
                if (f == SpmvCUDA_ELLFuncs_NN_TraposedImpl) {     /* the remaining kernels read the row-major matrix */
                    hipFreeSpmat(&dMat);
                    if (spMatCpyELL(matELL, &dMat)) goto _free;
                }
                hprintsf("@computing SpMV   with func: CUDA ELL %u at:%p\n", f, (void*)SpmvCUDA_ELLFuncs[f]);
                if (testSpMVImplHip(SpmvCUDA_ELLFuncs[f], &dMat, mat->M, dVect, dOutV, outV, oracleOut, &avg)) goto _free;
                char tag[32];
                snprintf(tag, sizeof tag, "HIP ELL %u", f);
                perfLine(tag, matELL->RL ? bytesCsr : bytesEll, mat->NZ, avg);
            }
            hipFreeSpmat(&dMat);
        }
    }
    if (!cudaOnly) {
        for (uint f = 0; f < STATIC_ARR_ELEMENTS_N(SpmvCSRFuncs); f++) {
            hprintsf("@computing SpMV   with func: OMP CSR %u at:%p\n", f, (void*)SpmvCSRFuncs[f]);
            if (testSpMVImplOMP(SpmvCSRFuncs[f], matCSR, vector, outV, oracleOut)) goto _free;
        }
        if (matELL)
            for (uint f = 0; f < STATIC_ARR_ELEMENTS_N(SpmvELLFuncs); f++) {
                hprintsf("@computing SpMV   with func: OMP ELL %u at:%p\n", f, (void*)SpmvELLFuncs[f]);
                if (testSpMVImplOMP(SpmvELLFuncs[f], matELL, vector, outV, oracleOut)) goto _free;
            }
    }
    out = EXIT_SUCCESS;

_free:
    hipFreeSpmat(&dMat);
    if (dVect) spmvHipVecFree(dVect);
    if (dOutV) spmvHipVecFree(dOutV);
    if (!ompOnly) spmvHipFinalize();
    if (matCSR) freeSpmat(matCSR);
    if (matELL) freeSpmat(matELL);
    if (matELL_t) freeSpmat(matELL_t);
    free(vector);
    free(outV);
    free(oracleOut);
    return out;
}

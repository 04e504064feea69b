/*
 * main.c -- plain-C command line driver for the MI355X path.
 * Same argv / scratch files / last stdout line as the reference's CLI
 * (src/main.cu:62-283):
 *     SpMV_HIP.elf <matrix.mtx> <vectorFile|RNDVECT> [COMPUTE_MODE]
 * COMPUTE_MODE is one of the CUDA_* names of the reference (or the HIP_*
 * synonyms); default CUDA_CSR_ROWS.  The OpenMP modes are not built into this
 * binary: the GPU library has no CPU path (the CPU variants live in the
 * reference itself and, restated, under oracle/ for checking).
 * Environment: SPMV_DEVICE (default 0), SPMV_BLOCK_X (threads per workgroup,
 * multiple of 64; default = library choice), SPMV_VARIANT (kernel variant of
 * the selected launcher, see spmvHip.h), SPMV_ELL_ROWLENS=0 to walk the ELL
 * padding like the reference kernels, SPMV_NGPU=n (CSR row modes only): shard
 * the rows over n devices and all-gather y with RCCL (spmvHipShardCSR).
 */
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "SpMV.h"
#include "parser.h"
#include "sparseMatrix.h"
#include "utils.h"

double Start, End, Elapsed, ElapsedInternal;
CONFIG Conf = {.gridRows = 8, .gridCols = 8};

#define RNDVECT "RNDVECT"
#define HELP "usage: MatrixMarket_sparse_matrix_COO, vectorFile || " RNDVECT ", [COMPUTE MODE:\n\tHIP:\t" \
    CUDA_CSR_ROWS "," CUDA_CSR_ROWS_WARP "," CUDA_CSR_TILES "," CUDA_CSR_STRIPES "," CUDA_CSR_AUTO "," CUDA_SELL_ROWS "," CUDA_ELL_ROWS "," CUDA_ELL_ROWS_NT "," CUDA_ELL_ROWS_WARP_NT \
    " (HIP_* synonyms accepted)]\n"

static long envLong(const char* name, long dflt) {
    const char* s = getenv(name);
    return s && *s ? strtol(s, NULL, 10) : dflt;
}

int main(int argc, char** argv) {
    int out = EXIT_FAILURE;
    if (argc < 3) { ERRPRINT(HELP); return out; }

    COMPUTE_MODE cmode = _CUDA_CSR_ROWS;
    if (argc > 3) {
        cmode = spmvModeFromString(argv[3]);
        if (cmode == _COMPUTE_MODE_INVALID) { ERRPRINT("INVALID COMPUTE_MODE ARGV[3] GIVEN\n" HELP); return out; }
        if (!spmvModeIsGpu(cmode)) {
            ERRPRINTS("%s is an OpenMP mode: this binary only drives the MI355X path\n" HELP, argv[3]);
            return out;
        }
        if (cmode == _CUDA_ELL_ROWS_WARP) {     /* declared but never implemented in the reference either (main.cu:137) */
            ERRPRINT("CUDA_ELL_ROWS_WARP has no implementation; use CUDA_ELL_ROWS_WARP_NN_TRANSPOSED\n");
            return out;
        }
    }
    const int toCSR = spmvModeIsCsr(cmode);

    spmat *mat = NULL, *ellT = NULL;
    spmat dMat;
    memset(&dMat, 0, sizeof dMat);
    double *vector = NULL, *outV = NULL, *dVect = NULL, *dOutV = NULL;

    if (spmvHipInit((int)envLong("SPMV_DEVICE", 0), sizeof(spmat), sizeof(CONFIG))) return out;

    char* trgtMatPath = TMP_EXTRACTED_MARTIX;           /* compressed input -> scratch copy, like main.cu:141-142 */
    {
        const int ex = extractInTmpFS(argv[1], TMP_EXTRACTED_MARTIX);
        if (ex < 0) trgtMatPath = argv[1];
        else if (ex > 0) goto _free;
    }
    if (!(mat = toCSR ? MMtoCSR(trgtMatPath) : MMtoELL(trgtMatPath))) {
        ERRPRINTS("err during parsing MatrixMarket -> %s\n", toCSR ? "CSR" : "ELL");
        goto _free;
    }
    ulong vectSize = mat->N;
    if (!strcmp(argv[2], RNDVECT)) {
        if (!(vector = malloc((vectSize ? vectSize : 1) * sizeof *vector))) { ERRPRINT("rnd vector malloc failed\n"); goto _free; }
        if (init_urndfd() || fillRndVector(vectSize, vector)) { ERRPRINT("fillRndVector errd\n"); goto _free; }
        if (writeDoubleVector(RNDVECTORDUMP, vector, vectSize)) ERRPRINT("RNDVECT dump err\n");
    } else {
        if (!(vector = readDoubleVector(argv[2], &vectSize))) { fprintf(stderr, "err during readDoubleVector at:%s\n", argv[2]); goto _free; }
        if (vectSize != mat->N) { ERRPRINT("vector not compatible with sparse matrix\n"); goto _free; }
    }
    if (!(outV = malloc((mat->M ? mat->M : 1) * sizeof *outV))) { ERRPRINT("outV malloc errd\n"); goto _free; }

    /* ---- multi-device path (new; the reference is single-GPU) */
    const long nGpu = envLong("SPMV_NGPU", 1);
    if (nGpu > 1 || getenv("SPMV_NGPU")) {
        if (cmode != _CUDA_CSR_ROWS && cmode != _CUDA_CSR_ROWS_WARP) { ERRPRINT("SPMV_NGPU needs CUDA_CSR_ROWS or CUDA_CSR_ROWS_WARP\n"); goto _free; }
        void* shards = NULL;
        double kSec = 0, gSec = 0;
        if (spmvHipShardCSR(mat, (int)nGpu, &shards)) goto _free;
        int rc = spmvHipSpMVSharded(shards, vector, cmode == _CUDA_CSR_ROWS_WARP, outV, &kSec, &gSec);
        spmvHipShardFree(shards);
        if (rc) { ERRPRINT("sharded SpMV failed\n"); goto _free; }
        Elapsed = ElapsedInternal = (kSec + gSec) * 1e3;
        if (writeDoubleVector(OUTVECTORDUMPRAW, outV, mat->M) || writeDoubleVectorAsStr(OUTVECTORDUMP, outV, mat->M))
            ERRPRINT("outV dump err\n");
        printf("nGPU: %ld\tkernelSeconds:%le gatherSeconds:%le GFLOPS:%lf\n", nGpu, kSec, gSec, 2.0 * mat->NZ / (kSec + gSec) * 1e-9);
        printf("cmode:%d\telapsed:\t %le elapsedInternal %le\n", cmode, Elapsed, ElapsedInternal);
        out = EXIT_SUCCESS;
        goto _free;
    }

    /* ---- host -> device */
    if (spmvHipVecAlloc(&dVect, mat->N) || spmvHipVecAlloc(&dOutV, mat->M)) goto _free;
    if (spmvHipVecUp(dVect, vector, mat->N)) goto _free;
    SPMV_HIP_INTERF func = NULL;
    const char* launcher = "";
    switch (cmode) {
        case _CUDA_CSR_ROWS:         func = &hipSpMVRowsCSR; launcher = "hipSpMVRowsCSR"; break;
        case _CUDA_CSR_ROWS_WARP:    func = &hipSpMVWarpPerRowCSR; launcher = "hipSpMVWarpPerRowCSR"; break;
        case _CUDA_CSR_TILES:        func = &hipSpMVTilesCSR; break;
        case _CUDA_SELL_ROWS:        func = &hipSpMVRowsSELL; break;
        case _CUDA_CSR_STRIPES:      func = &hipSpMVStripesCSR; break;
        case _CUDA_CSR_AUTO:         func = &hipSpMVAutoCSR; break;
        case _CUDA_ELL_ROWS:         func = &hipSpMVRowsELL; break;
        case _CUDA_ELL_ROWS_NT:      func = &hipSpMVRowsELLNNTransposed; break;
        case _CUDA_ELL_ROWS_WARP_NT: func = &hipSpMVWarpsPerRowELLNTrasposed; break;
        default: goto _free;
    }
    if (toCSR) {
        if (spMatCpyCSR(mat, &dMat)) goto _free;
    } else if (cmode == _CUDA_ELL_ROWS) {       /* coalesced thread-per-row wants the transposed matrix (main.cu:206-213) */
        if (!(ellT = ellTranspose(mat))) goto _free;
        if (spMatCpyELL(ellT, &dMat)) goto _free;
        freeSpmat(ellT);
        ellT = NULL;
    } else if (spMatCpyELL(mat, &dMat)) goto _free;

    const long variant = envLong("SPMV_VARIANT", -1);
    if (variant >= 0 && *launcher && spmvHipSetVariant(launcher, (int)variant)) goto _free;
    spmvHipSetEllRowLens((int)envLong("SPMV_ELL_ROWLENS", 1));
    Conf.blockSize.x = (unsigned)envLong("SPMV_BLOCK_X", 0);

    /* ---- launch (y poisoned first so a kernel that skips rows cannot pass) */
    if (spmvHipVecFill(dOutV, mat->M, 0x7FF8DEADDEADDEADull)) goto _free;
    if (func(&dMat, dVect, Conf, dOutV)) { ERRPRINT("compute function selected failed...\n"); goto _free; }
    const double seconds = spmvHipLastKernelSeconds();
    Elapsed = ElapsedInternal = seconds * 1e3;          /* the reference's GPU CLI reports milliseconds */
    if (spmvHipVecDown(outV, dOutV, mat->M)) goto _free;

    if (writeDoubleVector(OUTVECTORDUMPRAW, outV, mat->M) || writeDoubleVectorAsStr(OUTVECTORDUMP, outV, mat->M))
        ERRPRINT("outV dump err\n");
    spmvDim3 g, b;
    spmvHipLastLaunch(&g, &b);
    const double bytes = (double)mat->NZ * 12 + (double)mat->M * 12 + (double)mat->N * 8;
    printf("hipBlockSize: %u %u %u\thipGridSize: %u %u %u\tseconds:%le GFLOPS:%lf GBps:%lf rooflineFrac:%lf\n",
           b.x, b.y, b.z, g.x, g.y, g.z, seconds, 2.0 * mat->NZ / seconds * 1e-9, bytes / seconds * 1e-9,
           bytes / seconds / 8e12);
    printf("cmode:%d\telapsed:\t %le elapsedInternal %le\n", cmode, Elapsed, ElapsedInternal);
    out = EXIT_SUCCESS;

_free:
    if (ellT) freeSpmat(ellT);
    if (mat) freeSpmat(mat);
    free(vector);
    free(outV);
    hipFreeSpmat(&dMat);
    if (dVect) spmvHipVecFree(dVect);
    if (dOutV) spmvHipVecFree(dOutV);
    spmvHipFinalize();
    return out;
}

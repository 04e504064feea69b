/*
 * oracle/spmv_oracle_tiles.c -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 *
 * CPU restatement of the reference's remaining OpenMP SpMV variants, kept as
 * CPU comparators so that the harness tables have the reference's shape
 * (SpmvCSRFuncs[5], SpmvELLFuncs[3], src/include/SpMV.h:146-159):
 *   spmvRowsBlocksCSR    follows src/SpMV_CSR_OMP.c:65-99    (gridRows row blocks, fair split)
 *   spmvTilesCSR         follows src/SpMV_CSR_OMP.c:101-162  (2-D tiles over per-row column offsets)
 *   spmvTilesAllocdCSR   follows src/SpMV_CSR_OMP.c:165-226  (2-D tiles over allocated column partitions)
 *   spmvRowsBlocksELL    follows src/SpMV_ELL_OMP.c:69-108
 *   spmvTilesELL         follows src/SpMV_ELL_OMP.c:110-174
 *   colsOffsetsPartitioningUnifRanges / colsPartitioningUnifRanges
 *                        follow src/commons/sparseUtils.c:37-142
 * Written from the algorithm; pinned bit-for-bit against oracle/_ref/libspmvref.so
 * (the reference's own sources compiled where they lie) by tests/test_oracle.py.
 * Same rules as spmv_oracle.c: only tests/, smoke() and bench.py's cpu_baseline
 * leg may use this file.
 */
#include <omp.h>
#include <stdlib.h>
#include <string.h>

#include "spmv_types.h"

typedef void (*CHUNKS_DISTR_INTERF)(ulong, spmat*, CONFIG*);
extern double Start, End, Elapsed, ElapsedInternal;

static inline void adaptChunks(CONFIG* cfg, ulong iters, spmat* mat) {
    if (cfg->chunkDistrbFunc) ((CHUNKS_DISTR_INTERF)cfg->chunkDistrbFunc)(iters, mat, cfg);
}
#define TIC() AUDIT_INTERNAL_TIMES Start = omp_get_wtime()
#define TOC() AUDIT_INTERNAL_TIMES { End = omp_get_wtime(); ElapsedInternal = End - Start; }

/* first nnz index of every (row, column group): out[r*gridCols + g]; out[M*gridCols] = NZ */
ulong* colsOffsetsPartitioningUnifRanges(spmat* A, ulong gridCols) {
    const ulong cells = A->M * gridCols;
    const ulong cb = A->N / gridCols, cbRem = A->N % gridCols;
    ulong* off = malloc((cells + 1) * sizeof *off);
    if (!off) { ERRPRINT("colsOffsetsPartitioningUnifRanges:\toffsets malloc errd\n"); return NULL; }
    for (ulong r = 0; r < A->M; r++) {
        ulong j = A->IRP[r];
        const ulong rowEnd = A->IRP[r + 1];
        off[IDX2D(r, 0, gridCols)] = j;
        for (ulong g = 1; g < gridCols; g++) {
            const ulong firstCol = UNIF_REMINDER_DISTRI_STARTIDX(g, cb, cbRem);
            while (j < rowEnd && A->JA[j] < firstCol) j++;
            off[IDX2D(r, g, gridCols)] = j;
        }
    }
    off[cells] = A->NZ;
    return off;
}

/* gridCols CSR matrices, one per column range, global column ids kept */
spmat* colsPartitioningUnifRanges(spmat* A, ulong gridCols) {
    const ulong cb = A->N / gridCols, cbRem = A->N % gridCols;
    spmat* parts = calloc(gridCols, sizeof *parts);
    ulong* filled = calloc(gridCols, sizeof *filled);
    if (!parts || !filled) goto fail;
    for (ulong g = 0; g < gridCols; g++) {
        spmat* p = parts + g;
        p->M = A->M;
        p->N = UNIF_REMINDER_DISTRI(g, cb, cbRem);
        p->IRP = calloc(A->M + 1, sizeof *p->IRP);
        p->RL = malloc((A->M ? A->M : 1) * sizeof *p->RL);
        p->AS = malloc((A->NZ ? A->NZ : 1) * sizeof *p->AS);     /* over-allocated, trimmed below */
        p->JA = malloc((A->NZ ? A->NZ : 1) * sizeof *p->JA);
        if (!p->IRP || !p->RL || !p->AS || !p->JA) goto fail;
    }
    for (ulong r = 0; r < A->M; r++) {
        ulong j = A->IRP[r];
        const ulong rowEnd = A->IRP[r + 1];
        ulong lastCol = 0;
        for (ulong g = 0; g < gridCols; g++) {
            spmat* p = parts + g;
            lastCol += UNIF_REMINDER_DISTRI(g, cb, cbRem);
            ulong n = 0;
            while (j + n < rowEnd && A->JA[j + n] < lastCol) n++;
            p->IRP[r] = filled[g];
            memcpy(p->AS + filled[g], A->AS + j, n * sizeof *A->AS);
            memcpy(p->JA + filled[g], A->JA + j, n * sizeof *A->JA);
            p->RL[r] = n;
            filled[g] += n;
            j += n;
        }
    }
    for (ulong g = 0; g < gridCols; g++) {
        spmat* p = parts + g;
        p->NZ = filled[g];
        p->IRP[A->M] = filled[g];
        double* as = realloc(p->AS, (filled[g] ? filled[g] : 1) * sizeof *p->AS);
        ulong* ja = realloc(p->JA, (filled[g] ? filled[g] : 1) * sizeof *p->JA);
        if (as) p->AS = as;
        if (ja) p->JA = ja;
    }
    free(filled);
    return parts;
fail:
    ERRPRINT("colsPartitioningUnifRanges: allocation failed\n");
    if (parts)
        for (ulong g = 0; g < gridCols; g++) { free(parts[g].IRP); free(parts[g].RL); free(parts[g].AS); free(parts[g].JA); }
    free(parts);
    free(filled);
    return NULL;
}

static void freeColParts(spmat* parts, ulong gridCols) {
    if (!parts) return;
    for (ulong g = 0; g < gridCols; g++) { free(parts[g].IRP); free(parts[g].RL); free(parts[g].AS); free(parts[g].JA); }
    free(parts);
}

/* y[r] = sum over column groups of the per-tile partial sums, in group order */
static void sumTilePartials(const double* partial, ulong M, ulong gridCols, double* y) {
    for (ulong r = 0; r < M; r++)
        for (ulong g = 0; g < gridCols; g++) y[r] += partial[IDX2D(r, g, gridCols)];
}

int spmvRowsBlocksCSR(spmat* mat, double* vect, CONFIG* cfg, double* outVect) {
    const ulong nb = cfg->gridRows, rb = mat->M / nb, rbRem = mat->M % nb;
    adaptChunks(cfg, nb, mat);
    TIC();
    #pragma omp parallel for schedule(runtime)
    for (ulong b = 0; b < nb; b++) {
        const ulong rows = UNIF_REMINDER_DISTRI(b, rb, rbRem);
        const ulong first = UNIF_REMINDER_DISTRI_STARTIDX(b, rb, rbRem);
        for (ulong r = first; r < first + rows; r++) {
            double acc = 0;
#if SIMD_ROWS_REDUCTION == TRUE
            #pragma omp simd reduction(+ : acc)
#endif
            for (ulong j = mat->IRP[r]; j < mat->IRP[r + 1]; j++) acc += mat->AS[j] * vect[mat->JA[j]];
            outVect[r] = acc;
        }
    }
    TOC();
    return EXIT_SUCCESS;
}

int spmvTilesCSR(spmat* mat, double* vect, CONFIG* cfg, double* outVect) {
    const ulong gr = cfg->gridRows, gc = cfg->gridCols, tiles = gr * gc;
    const ulong rb = mat->M / gr, rbRem = mat->M % gr;
    ulong* off = colsOffsetsPartitioningUnifRanges(mat, gc);
    double* partial = malloc((mat->M && gc ? mat->M * gc : 1) * sizeof *partial);
    if (!off || !partial) { ERRPRINT("spmvTiles:  aux alloc errd\n"); free(off); free(partial); return EXIT_FAILURE; }
    memset(outVect, 0, mat->M * sizeof *outVect);
    adaptChunks(cfg, tiles, mat);
    TIC();
    #pragma omp parallel for schedule(runtime)
    for (ulong t = 0; t < tiles; t++) {
        const ulong ti = t / gc, tj = t % gc;
        const ulong rows = UNIF_REMINDER_DISTRI(ti, rb, rbRem);
        const ulong first = UNIF_REMINDER_DISTRI_STARTIDX(ti, rb, rbRem);
        for (ulong r = first; r < first + rows; r++) {
            const ulong cell = IDX2D(r, tj, gc);
            double acc = 0;
#if SIMD_ROWS_REDUCTION == TRUE
            #pragma omp simd reduction(+ : acc)
#endif
            for (ulong j = off[cell]; j < off[cell + 1]; j++) acc += mat->AS[j] * vect[mat->JA[j]];
            partial[cell] = acc;
        }
    }
    sumTilePartials(partial, mat->M, gc, outVect);
    TOC();
    free(off);
    free(partial);
    return EXIT_SUCCESS;
}

int spmvTilesAllocdCSR(spmat* mat, double* vect, CONFIG* cfg, double* outVect) {
    const ulong gr = cfg->gridRows, gc = cfg->gridCols, tiles = gr * gc;
    const ulong rb = mat->M / gr, rbRem = mat->M % gr;
    spmat* parts = colsPartitioningUnifRanges(mat, gc);
    if (!parts) return EXIT_FAILURE;
    double* partial = malloc((mat->M && gc ? mat->M * gc : 1) * sizeof *partial);
    if (!partial) { ERRPRINT("spmvTiles:  tilesOutTmp malloc errd\n"); freeColParts(parts, gc); return EXIT_FAILURE; }
    memset(outVect, 0, mat->M * sizeof *outVect);
    adaptChunks(cfg, tiles, mat);
    TIC();
    #pragma omp parallel for schedule(runtime)
    for (ulong t = 0; t < tiles; t++) {
        const ulong ti = t / gc, tj = t % gc;
        const spmat* p = parts + tj;
        const ulong rows = UNIF_REMINDER_DISTRI(ti, rb, rbRem);
        const ulong first = UNIF_REMINDER_DISTRI_STARTIDX(ti, rb, rbRem);
        for (ulong r = first; r < first + rows; r++) {
            double acc = 0;
#if SIMD_ROWS_REDUCTION == TRUE
            #pragma omp simd reduction(+ : acc)
#endif
            for (ulong j = p->IRP[r]; j < p->IRP[r + 1]; j++) acc += p->AS[j] * vect[p->JA[j]];
            partial[IDX2D(r, tj, gc)] = acc;
        }
    }
    sumTilePartials(partial, mat->M, gc, outVect);
    TOC();
    freeColParts(parts, gc);
    free(partial);
    return EXIT_SUCCESS;
}

int spmvRowsBlocksELL(spmat* mat, double* vect, CONFIG* cfg, double* outVect) {
    const ulong nb = cfg->gridRows, rb = mat->M / nb, rbRem = mat->M % nb, K = mat->MAX_ROW_NZ;
    adaptChunks(cfg, nb, mat);
    TIC();
    #pragma omp parallel for schedule(runtime)
    for (ulong b = 0; b < nb; b++) {
        const ulong rows = UNIF_REMINDER_DISTRI(b, rb, rbRem);
        const ulong first = UNIF_REMINDER_DISTRI_STARTIDX(b, rb, rbRem);
        for (ulong r = first; r < first + rows; r++) {
            const ulong s = IDX2D(r, 0, K), e = s + (mat->RL ? mat->RL[r] : K);
            double acc = 0;
#if SIMD_ROWS_REDUCTION == TRUE
            #pragma omp simd reduction(+ : acc)
#endif
            for (ulong j = s; j < e; j++) acc += mat->AS[j] * vect[mat->JA[j]];
            outVect[r] = acc;
        }
    }
    TOC();
    return EXIT_SUCCESS;
}

int spmvTilesELL(spmat* mat, double* vect, CONFIG* cfg, double* outVect) {
    const ulong gr = cfg->gridRows, gc = cfg->gridCols, tiles = gr * gc, K = mat->MAX_ROW_NZ;
    const ulong rb = mat->M / gr, rbRem = mat->M % gr;
    const ulong sb = K / gc, sbRem = K % gc;           /* the SLOTS, not the columns, are split here */
    double* partial = malloc((mat->M && gc ? mat->M * gc : 1) * sizeof *partial);
    if (!partial) { ERRPRINT("spmvTiles:  tilesOutTmp malloc errd\n"); return EXIT_FAILURE; }
    memset(outVect, 0, mat->M * sizeof *outVect);
    adaptChunks(cfg, tiles, mat);
    TIC();
    #pragma omp parallel for schedule(runtime)
    for (ulong t = 0; t < tiles; t++) {
        const ulong ti = t / gc, tj = t % gc;
        const ulong rows = UNIF_REMINDER_DISTRI(ti, rb, rbRem);
        const ulong first = UNIF_REMINDER_DISTRI_STARTIDX(ti, rb, rbRem);
        const ulong slots = UNIF_REMINDER_DISTRI(tj, sb, sbRem);
        const ulong firstSlot = UNIF_REMINDER_DISTRI_STARTIDX(tj, sb, sbRem);
        for (ulong r = first; r < first + rows; r++) {
            const ulong s = IDX2D(r, firstSlot, K);
            ulong e = s + slots;
            if (mat->RL) e = MIN(e, IDX2D(r, 0, K) + mat->RL[r]);
            double acc = 0;
#if SIMD_ROWS_REDUCTION == TRUE
            #pragma omp simd reduction(+ : acc)
#endif
            for (ulong j = s; j < e; j++) acc += mat->AS[j] * vect[mat->JA[j]];
            partial[IDX2D(r, tj, gc)] = acc;
        }
    }
    sumTilePartials(partial, mat->M, gc, outVect);
    TOC();
    free(partial);
    return EXIT_SUCCESS;
}

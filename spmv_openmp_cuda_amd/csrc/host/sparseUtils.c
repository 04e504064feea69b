/*
 * sparseUtils.c -- host-side allocation and reshaping of `spmat`.
 * Follows the reference's src/include/sparseMatrix.h:83-134 (alloc/free) and
 * src/commons/sparseUtils.c:145-222 (ellTranspose, spmatDiff, CSRToDense).
 */
#include <stdlib.h>
#include <string.h>

#include "sparseMatrix.h"
#include "utils.h"

void freeSpmatInternal(spmat* mat) {
    if (!mat) return;
    free(mat->AS); free(mat->JA); free(mat->IRP); free(mat->RL);
    mat->AS = NULL; mat->JA = NULL; mat->IRP = NULL; mat->RL = NULL;
}
void freeSpmat(spmat* mat) {
    freeSpmatInternal(mat);
    free(mat);
}

int allocSpMatrixInternal(ulong rows, ulong cols, spmat* mat) {
    mat->M = rows;
    mat->N = cols;
    mat->IRP = calloc(rows + 1, sizeof *mat->IRP);
    mat->RL = malloc((rows ? rows : 1) * sizeof *mat->RL);
    if (!mat->IRP || !mat->RL) {
        ERRPRINT("IRP/RL alloc err\n");
        freeSpmatInternal(mat);
        return EXIT_FAILURE;
    }
    return EXIT_SUCCESS;
}
spmat* allocSpMatrix(ulong rows, ulong cols) {
    spmat* mat = calloc(1, sizeof *mat);
    if (!mat) { ERRPRINT("mat  calloc failed\n"); return NULL; }
    if (allocSpMatrixInternal(rows, cols, mat)) { free(mat); return NULL; }
    return mat;
}

spmat* ellTranspose(spmat* m) {
    const ulong rows = m->M, slots = m->MAX_ROW_NZ, cells = rows * slots;
    spmat* out = calloc(1, sizeof *out);
    if (!out) { ERRPRINT("ellTranspose: out callc errd\n"); return NULL; }
    out->AS = malloc((cells ? cells : 1) * sizeof *out->AS);
    out->JA = malloc((cells ? cells : 1) * sizeof *out->JA);
    if (m->RL) out->RL = malloc((rows ? rows : 1) * sizeof *out->RL);
    if (!out->AS || !out->JA || (m->RL && !out->RL)) { ERRPRINT("ellTranspose: invalid malloc\n"); freeSpmat(out); return NULL; }
    if (m->RL) memcpy(out->RL, m->RL, rows * sizeof *out->RL);
    out->NZ = m->NZ;
    out->M = slots;             /* field meaning swaps, as in the reference */
    out->N = rows;
    out->MAX_ROW_NZ = rows;
    out->dev = SPMAT_TAG_ELL_TRANSPOSED;
    out->pitchJA = m->N;        /* the reference's field swap loses the column count: kept here (host structs do not use the
                                   pitch fields) so that the upload can check column ids against it */
    /* blocked to keep both sides of the transposition in cache */
    enum { TB = 32 };
    for (ulong r0 = 0; r0 < rows; r0 += TB)
        for (ulong c0 = 0; c0 < slots; c0 += TB) {
            const ulong r1 = MIN(r0 + TB, rows), c1 = MIN(c0 + TB, slots);
            for (ulong r = r0; r < r1; ++r)
                for (ulong c = c0; c < c1; ++c) {
                    out->JA[IDX2D(c, r, rows)] = m->JA[IDX2D(r, c, slots)];
                    out->AS[IDX2D(c, r, rows)] = m->AS[IDX2D(r, c, slots)];
                }
        }
    return out;
}

spmat* csrToEll(const spmat* csr) {
    ulong maxRow = 0;
    for (ulong r = 0; r < csr->M; ++r) maxRow = MAX(maxRow, csr->IRP[r + 1] - csr->IRP[r]);
#ifdef LIMIT_ELL_SIZE
    if (2 * csr->M * maxRow > (ulong)ELL_MAX_ENTRIES) {
        ERRPRINTS("csrToEll: %lu padded entries exceed the ELL threshold %lu\n", 2 * csr->M * maxRow, (ulong)ELL_MAX_ENTRIES);
        return NULL;
    }
#endif
    spmat* out = calloc(1, sizeof *out);
    if (!out) return NULL;
    const ulong cells = csr->M * maxRow;
    out->M = csr->M; out->N = csr->N; out->NZ = csr->NZ; out->MAX_ROW_NZ = maxRow;
    out->AS = calloc(cells ? cells : 1, sizeof *out->AS);
    out->JA = calloc(cells ? cells : 1, sizeof *out->JA);
    out->RL = malloc((csr->M ? csr->M : 1) * sizeof *out->RL);
    if (!out->AS || !out->JA || !out->RL) { freeSpmat(out); return NULL; }
    for (ulong r = 0; r < csr->M; ++r) {
        const ulong b = csr->IRP[r], len = csr->IRP[r + 1] - b;
        out->RL[r] = len;
        memcpy(out->JA + r * maxRow, csr->JA + b, len * sizeof *out->JA);
        memcpy(out->AS + r * maxRow, csr->AS + b, len * sizeof *out->AS);
    }
    return out;
}

int spmatDiff(spmat* A, spmat* B) {
    if (A->NZ != B->NZ) { ERRPRINT("NZ differ\n"); return EXIT_FAILURE; }
    if (doubleVectorsDiff(A->AS, B->AS, A->NZ, NULL)) { ERRPRINT("AS DIFFER\n"); return EXIT_FAILURE; }
    if (memcmp(A->JA, B->JA, A->NZ * sizeof *A->JA)) { ERRPRINT("JA differ\n"); return EXIT_FAILURE; }
    return EXIT_SUCCESS;
}

double* CSRToDense(spmat* sparseMat) {
    ulong denseSize;
    if (__builtin_umull_overflow(sparseMat->M, sparseMat->N, &denseSize)) { ERRPRINT("overflow in dense allocation\n"); return NULL; }
    double* dense = calloc(denseSize ? denseSize : 1, sizeof *dense);
    if (!dense) { ERRPRINT("dense matrix alloc failed\n"); return NULL; }
    for (ulong i = 0; i < sparseMat->M; ++i)
        for (ulong k = sparseMat->IRP[i]; k < sparseMat->IRP[i + 1]; ++k)
            dense[IDX2D(i, sparseMat->JA[k], sparseMat->N)] = sparseMat->AS[k];
    return dense;
}

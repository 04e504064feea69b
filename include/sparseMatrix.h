/*
 * sparseMatrix.h -- allocation / release / reshaping of `spmat` on the host.
 * Surface of the reference's src/include/sparseMatrix.h:83-151 (alloc, free,
 * ellTranspose, spmatDiff); the struct itself is in spmv_types.h.
 * Out of scope here: the 2-D column partitioning used only by the reference's
 * CPU tile variants (sparseUtils.c:37-142) -- see DESIGN.md.
 */
#ifndef SPMV_SPARSEMATRIX_H
#define SPMV_SPARSEMATRIX_H

#include "spmv_types.h"

#ifdef __cplusplus
extern "C" {
#endif

/* zeroed struct with IRP[rows+1] (zeroed) and RL[rows]; JA/AS left NULL */
spmat* allocSpMatrix(ulong rows, ulong cols);
int    allocSpMatrixInternal(ulong rows, ulong cols, spmat* mat);
/* free(NULL)-safe release of the arrays / of arrays + struct */
void   freeSpmatInternal(spmat* mat);
void   freeSpmat(spmat* mat);

/*
 * Column-major copy of a row-major ELL matrix, for coalesced thread-per-row
 * access (sparseUtils.c:145-185).  Keeps the reference's field convention:
 *   out->M = in->MAX_ROW_NZ (slots), out->N = in->M, out->MAX_ROW_NZ = in->M (rows)
 * and out->JA[slot*rows + row] = in->JA[row*slots + slot]; RL is duplicated.
 */
spmat* ellTranspose(spmat* m);

/* CSR -> row-major ELL with {JA=0, AS=0} padding; NULL if it trips the
 * ELL_MAX_ENTRIES guard the loader applies (parser.c:223-232). */
spmat* csrToEll(const spmat* csr);

/* EXIT_SUCCESS when NZ, JA and AS (within DOUBLE_DIFF_THREASH) agree
 * (sparseUtils.c:187-201; compares all NZ*sizeof(ulong) bytes of JA, where the
 * reference's memcmp stops after NZ bytes) */
int spmatDiff(spmat* A, spmat* B);

/* dense row-major copy of a CSR matrix, or NULL on overflow/alloc failure
 * (sparseUtils.c:203-222) */
double* CSRToDense(spmat* sparseMat);

#ifdef __cplusplus
}
#endif
#endif

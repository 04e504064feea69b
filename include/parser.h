/*
 * parser.h -- MatrixMarket coordinate files -> COO -> CSR / ELL.
 * Surface of the reference's src/include/parser.h:21-76 (entry, MatrixMarket,
 * MMRead, MMtoCOO, COOtoCSR, COOtoELL, MMtoCSR, MMtoELL) on top of a minimal
 * banner/size reader replacing the two NIST mmio calls it uses
 * (src/lib/mmio.c:96-179 mm_read_banner, :189-217 mm_read_mtx_crd_size).
 */
#ifndef SPMV_PARSER_H
#define SPMV_PARSER_H

#include <stdio.h>
#include "spmv_types.h"

#ifdef __cplusplus
extern "C" {
#endif

/* 4-char type code, same encoding as NIST mmio.h:16,31-46:
 * [0]='M' matrix, [1]='C' coordinate|'A' array, [2]='R'|'C'|'P'|'I', [3]='G'|'S'|'K'|'H' */
typedef char MM_typecode[4];
#define mm_is_matrix(t)     ((t)[0] == 'M')
#define mm_is_sparse(t)     ((t)[1] == 'C')
#define mm_is_dense(t)      ((t)[1] == 'A')
#define mm_is_complex(t)    ((t)[2] == 'C')
#define mm_is_real(t)       ((t)[2] == 'R')
#define mm_is_pattern(t)    ((t)[2] == 'P')
#define mm_is_integer(t)    ((t)[2] == 'I')
#define mm_is_general(t)    ((t)[3] == 'G')
#define mm_is_symmetric(t)  ((t)[3] == 'S')
#define mm_is_skew(t)       ((t)[3] == 'K')
#define mm_is_hermitian(t)  ((t)[3] == 'H')
#define MM_PREMATURE_EOF     12
#define MM_NO_HEADER         14
#define MM_UNSUPPORTED_TYPE  15
int mm_read_banner(FILE* f, MM_typecode* matcode);
int mm_read_mtx_crd_size(FILE* f, ulong* M, ulong* N, ulong* nz);

typedef struct {
    ulong  row;
    ulong  col;
    double val;
} entry;                    /* one COO entry, 0-based */

typedef struct {
    MM_typecode mcode;
    entry*      entries;
    ulong*      rowLens;
    ulong       M, N, NZ;   /* NZ = entries after symmetric expansion */
} MatrixMarket;

/* EXIT_SUCCESS iff the type code is a sparse real/integer/pattern matrix this
 * engine can run (rejects array and, unlike the reference, complex up front) */
int           MMCheck(MM_typecode mcode);
MatrixMarket* MMRead(char* matPath);
void          freeMatrixMarket(MatrixMarket* mm);
/* entries of `fp` (positioned after the size line) as 0-based COO; symmetric
 * files are mirrored, pattern files get value 1.0; *NZ is updated and
 * rowLens[] (zeroed by the caller, M entries) counts entries per row */
entry*        MMtoCOO(ulong* NZ, FILE* fp, MM_typecode mcode, ulong* rowLens);
/* The same from memory, by all host cores (csrc/host/mmfast.c): buf[0..len) is everything behind the size line,
 * buf[len] == 0.  *status: 0, MMFAST_ERROR (message printed) or MMFAST_FALLBACK (the entries are not laid out one per
 * line, or something the tokeniser does not want to judge: parse the stream with MMtoCOO instead). */
#define MMFAST_ERROR    1
#define MMFAST_FALLBACK 2
entry*        MMtoCOOFromBuffer(ulong* NZ, const char* buf, size_t len, MM_typecode mcode, ulong M, ulong N, ulong* rowLens, int* status);
/* COO -> CSR / ELL.  Entries of one row must arrive in ascending column order
 * (any interleaving of rows); otherwise EXIT_FAILURE. */
int           COOtoCSR(entry* entries, spmat* mat, ulong* rowLens);
int           COOtoELL(entry* entries, spmat* mat, ulong* rowLens);
spmat*        MMtoCSR(char* matPath);
spmat*        MMtoELL(char* matPath);

#ifdef __cplusplus
}
#endif
#endif

/*
 * parser.c -- MatrixMarket coordinate file -> COO -> CSR / ELL (host side).
 * Behaviour follows the reference's src/lib/parser.c:28-376 and the two NIST
 * mmio routines it calls (src/lib/mmio.c:96-179,189-217); written from scratch.
 * Semantics pinned by tests/golden (produced by the reference's own CLI):
 *   - 1-based file indices become 0-based            (parser.c:82-83)
 *   - `symmetric`: off-diagonal entries are mirrored, NZ = 2*NZ - diag (:49-51,85-97)
 *   - `pattern`  : value 1.0                         (:59-61)
 *   - `integer`  : parsed as double                  (:62-63)
 *   - rows keep the order in which their entries appear in the file; with
 *     CONSISTENCY_CHECKS (default on) an entry whose column does not increase
 *     within its row is an error                     (:195-202,265-272)
 *   - ELL refuses 2*M*maxRow > ELL_MAX_ENTRIES        (:223-232)
 * Deliberate differences: `complex` and `array` files are rejected up front
 * (the reference loops forever on `complex` without consistency checks), sizes
 * are parsed as 64-bit (the reference goes through `int`), every error path
 * releases what it allocated, out-of-range indices are an error.
 */
#include <ctype.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <sys/stat.h>

#include "parser.h"
#include "sparseMatrix.h"

#define MM_LINE_MAX 1025
#define MM_TOKEN_MAX 64

static void lowerInPlace(char* s) { for (; *s; ++s) *s = (char)tolower((unsigned char)*s); }

int mm_read_banner(FILE* f, MM_typecode* matcode) {
    char line[MM_LINE_MAX];
    char tok[5][MM_TOKEN_MAX];
    memset(*matcode, ' ', 4);
    if (!fgets(line, sizeof line, f)) return MM_PREMATURE_EOF;
    if (sscanf(line, "%63s %63s %63s %63s %63s", tok[0], tok[1], tok[2], tok[3], tok[4]) != 5)
        return MM_PREMATURE_EOF;
    for (int i = 1; i < 5; ++i) lowerInPlace(tok[i]);
    if (strncmp(tok[0], "%%MatrixMarket", 14) != 0) return MM_NO_HEADER;
    if (strcmp(tok[1], "matrix") != 0) return MM_UNSUPPORTED_TYPE;
    (*matcode)[0] = 'M';
    if (!strcmp(tok[2], "coordinate")) (*matcode)[1] = 'C';
    else if (!strcmp(tok[2], "array")) (*matcode)[1] = 'A';
    else return MM_UNSUPPORTED_TYPE;
    if (!strcmp(tok[3], "real")) (*matcode)[2] = 'R';
    else if (!strcmp(tok[3], "complex")) (*matcode)[2] = 'C';
    else if (!strcmp(tok[3], "pattern")) (*matcode)[2] = 'P';
    else if (!strcmp(tok[3], "integer")) (*matcode)[2] = 'I';
    else return MM_UNSUPPORTED_TYPE;
    if (!strcmp(tok[4], "general")) (*matcode)[3] = 'G';
    else if (!strcmp(tok[4], "symmetric")) (*matcode)[3] = 'S';
    else if (!strcmp(tok[4], "hermitian")) (*matcode)[3] = 'H';
    else if (!strcmp(tok[4], "skew-symmetric")) (*matcode)[3] = 'K';
    else return MM_UNSUPPORTED_TYPE;
    return 0;
}

int mm_read_mtx_crd_size(FILE* f, ulong* M, ulong* N, ulong* nz) {
    char line[MM_LINE_MAX];
    *M = *N = *nz = 0;
    for (;;) {                              /* skip %-comments and blank lines */
        if (!fgets(line, sizeof line, f)) return MM_PREMATURE_EOF;
        if (line[0] == '%') continue;
        if (sscanf(line, "%lu %lu %lu", M, N, nz) == 3) return 0;
        const char* p = line;
        while (*p && isspace((unsigned char)*p)) ++p;
        if (*p) return MM_UNSUPPORTED_TYPE;  /* non-blank garbage where the size line should be */
    }
}

int MMCheck(MM_typecode mcode) {
    if (!mm_is_matrix(mcode)) { ERRPRINT("invalid matrix: not a matrix\n"); return EXIT_FAILURE; }
    if (mm_is_dense(mcode)) { ERRPRINT("invalid matrix: not a supported sparse matrix\tDENSE MAT\n"); return EXIT_FAILURE; }
    if (mm_is_complex(mcode)) { ERRPRINT("invalid matrix: complex values are not supported\n"); return EXIT_FAILURE; }
    if (!(mm_is_real(mcode) || mm_is_integer(mcode) || mm_is_pattern(mcode))) {
        ERRPRINT("invalid matrix: unknown value type\n");
        return EXIT_FAILURE;
    }
    return EXIT_SUCCESS;
}

entry* MMtoCOO(ulong* NZ, FILE* fp, MM_typecode mcode, ulong* rowLens) {
    const int sym = mm_is_symmetric(mcode), pat = mm_is_pattern(mcode);
    const ulong declared = *NZ;
    if (declared > SIZE_MAX / sizeof(entry) / 2) {     /* e.g. a size line with "-1": %lu wraps it; the products below must not */
        ERRPRINTS("invalid matrix: %lu entries declared\n", declared);
        return NULL;
    }
    const ulong cap = sym ? 2 * declared : declared;
    entry* entries = malloc((cap ? cap : 1) * sizeof *entries);
    if (!entries) { ERRPRINT("MMtoCOO:  entries malloc errd\n"); return NULL; }
    ulong n = 0, diag = 0, fileEntries = 0;
    for (;;) {
        ulong row, col;
        double val = 1.0;
        int got = pat ? fscanf(fp, "%lu %lu", &row, &col) : fscanf(fp, "%lu %lu %lf", &row, &col, &val);
        if (got == EOF) {
            if (ferror(fp)) { perror("fscanf EOF"); goto fail; }
            break;
        }
        if (got != (pat ? 2 : 3)) { ERRPRINT("invalid matrix: not consistent entry scannable\n"); goto fail; }
        if (row == 0 || col == 0) { ERRPRINT("invalid matrix: 0 index in a 1-based file\n"); goto fail; }
        if (++fileEntries > declared || n + (sym && row != col ? 2 : 1) > cap) {
            ERRPRINT("invalid matrix: more entries than declared\n");
            goto fail;
        }
        rowLens[row - 1]++;
        entries[n++] = (entry){.row = row - 1, .col = col - 1, .val = val};
        if (sym && row != col) {
            rowLens[col - 1]++;
            entries[n++] = (entry){.row = col - 1, .col = row - 1, .val = val};
        } else diag++;
    }
    if (fileEntries != declared) {
        ERRPRINTS("invalid matrix: %lu entries declared, %lu found\n", declared, fileEntries);
        goto fail;
    }
    (void)diag;
    *NZ = n;
    return entries;
fail:
    free(entries);
    return NULL;
}

void freeMatrixMarket(MatrixMarket* mm) {
    if (!mm) return;
    free(mm->entries);
    free(mm->rowLens);
    free(mm);
}

MatrixMarket* MMRead(char* matPath) {
    FILE* fp = fopen(matPath, "r");
    if (!fp) { perror("fopen"); return NULL; }
    MatrixMarket* out = calloc(1, sizeof *out);
    if (!out) { ERRPRINT("MMRead out malloc errd\n"); fclose(fp); return NULL; }
    if (mm_read_banner(fp, &out->mcode) != 0) { fprintf(stderr, "mm_read_banner err at:%s\n", matPath); goto fail; }
    if (MMCheck(out->mcode)) goto fail;
    if (mm_read_mtx_crd_size(fp, &out->M, &out->N, &out->NZ)) { fprintf(stderr, "mm_read_mtx_crd_size err at %s:\n", matPath); goto fail; }
    if (!(out->rowLens = calloc(out->M ? out->M : 1, sizeof *out->rowLens))) { ERRPRINT("MMRead:\trowLens calloc errd\n"); goto fail; }
    {   /* fast path: the rest of a regular file goes to memory and is parsed by all cores (mmfast.c) */
        struct stat st;
        const long pos = ftell(fp);
        if (pos >= 0 && fstat(fileno(fp), &st) == 0 && S_ISREG(st.st_mode) && (long)st.st_size >= pos) {
            const size_t len = (size_t)st.st_size - (size_t)pos;
            char* buf = malloc(len + 1);
            if (buf && fread(buf, 1, len, fp) == len) {
                buf[len] = 0;
                int status = 0;
                out->entries = MMtoCOOFromBuffer(&out->NZ, buf, len, out->mcode, out->M, out->N, out->rowLens, &status);
                free(buf);
                if (out->entries) { fclose(fp); return out; }
                if (status != MMFAST_FALLBACK) { ERRPRINTS("MAT PARSE TO CSR ERR at:%s\n", matPath); goto fail; }
                memset(out->rowLens, 0, (out->M ? out->M : 1) * sizeof *out->rowLens);
            } else free(buf);
            if (fseek(fp, pos, SEEK_SET)) { perror("fseek"); goto fail; }
        }
    }
    {   /* serial path (pipes, odd layouts): bounds are checked in a first pass, the entries read in a second */
        const ulong M = out->M, N = out->N;
        long pos = ftell(fp);
        ulong r, c; double v; int ok = 1;
        const int pat = mm_is_pattern(out->mcode);
        for (;;) {
            int got = pat ? fscanf(fp, "%lu %lu", &r, &c) : fscanf(fp, "%lu %lu %lf", &r, &c, &v);
            if (got == EOF || got != (pat ? 2 : 3)) break;
            if (r > M || c > N || (mm_is_symmetric(out->mcode) && (c > M || r > N))) { ok = 0; break; }
        }
        if (!ok) { ERRPRINT("invalid matrix: entry outside the declared dimensions\n"); goto fail; }
        if (fseek(fp, pos, SEEK_SET)) { perror("fseek"); goto fail; }
    }
    if (!(out->entries = MMtoCOO(&out->NZ, fp, out->mcode, out->rowLens))) { ERRPRINTS("MAT PARSE TO CSR ERR at:%s\n", matPath); goto fail; }
    fclose(fp);
    return out;
fail:
    freeMatrixMarket(out);
    fclose(fp);
    return NULL;
}

/* shared by both conversions: place entry i at the next free slot of its row,
 * enforcing ascending columns per row */
static int placeEntries(const entry* entries, ulong nz, ulong M, const ulong* rowBase, ulong rowStride,
                        ulong* JA, double* AS) {
    int out = EXIT_FAILURE;
    ulong* next = calloc(M ? M : 1, sizeof *next);
    long* lastCol = malloc((M ? M : 1) * sizeof *lastCol);
    if (!next || !lastCol) { ERRPRINT("COO conversion: aux alloc errd\n"); goto done; }
    for (ulong r = 0; r < M; ++r) lastCol[r] = -1;
    for (ulong i = 0; i < nz; ++i) {
        const entry* e = entries + i;
        CONSISTENCY_CHECKS {
            if (lastCol[e->row] >= (long)e->col) {
                ERRPRINTS("not sorted entry:%ld,%ld,%lf", e->row, e->col, e->val);
                goto done;
            }
            lastCol[e->row] = (long)e->col;
        }
        const ulong at = (rowBase ? rowBase[e->row] : e->row * rowStride) + next[e->row]++;
        AS[at] = e->val;
        JA[at] = e->col;
    }
    out = EXIT_SUCCESS;
done:
    free(next);
    free(lastCol);
    return out;
}

int COOtoCSR(entry* entries, spmat* mat, ulong* rowLens) {
    mat->IRP[0] = 0;
    for (ulong r = 0; r < mat->M; ++r) mat->IRP[r + 1] = mat->IRP[r] + rowLens[r];
    if (mat->IRP[mat->M] != mat->NZ) { ERRPRINT("COOtoCSR: row lengths do not add up to NZ\n"); return EXIT_FAILURE; }
    return placeEntries(entries, mat->NZ, mat->M, mat->IRP, 0, mat->JA, mat->AS);
}

int COOtoELL(entry* entries, spmat* mat, ulong* rowLens) {
    ulong maxRow = 0;
    for (ulong r = 0; r < mat->M; ++r) maxRow = MAX(maxRow, rowLens[r]);
    ulong cells;
    if (__builtin_mul_overflow(mat->M, maxRow, &cells) || cells > SIZE_MAX / sizeof(double) / 2) {
        ERRPRINTS("MMtoELL:\t%lu rows x %lu slots do not fit the address space\n", mat->M, maxRow);
        return EXIT_FAILURE;
    }
#ifdef LIMIT_ELL_SIZE
    const ulong ellEntriesTot = 2 * cells;
    if (ellEntriesTot > (ulong)ELL_MAX_ENTRIES) {
        ERRPRINTS("Required entries %lu -> %lu uMB for the matrix exceed the designated threashold of: %lu  -> %lu MB for ellpack\n",
                  ellEntriesTot, (sizeof(double) * ellEntriesTot) >> 20, (ulong)ELL_MAX_ENTRIES,
                  (sizeof(double) * (ulong)ELL_MAX_ENTRIES) >> 20);
        return EXIT_FAILURE;
    }
#endif
    mat->AS = calloc(cells ? cells : 1, sizeof *mat->AS);     /* padding = {0.0, col 0} */
    mat->JA = calloc(cells ? cells : 1, sizeof *mat->JA);
    if (!mat->AS || !mat->JA) { ERRPRINT("MMtoELL:\tELL arrays calloc errd\n"); return EXIT_FAILURE; }
    mat->MAX_ROW_NZ = maxRow;
    return placeEntries(entries, mat->NZ, mat->M, NULL, maxRow, mat->JA, mat->AS);
}

static spmat* fromMM(char* matPath, int toCsr) {
    MatrixMarket* mm = MMRead(matPath);
    if (!mm) { ERRPRINT("MatrixMarket parse err\n"); return NULL; }
    spmat* mat = calloc(1, sizeof *mat);
    if (!mat) { ERRPRINT("MMto*: mat struct alloc errd"); freeMatrixMarket(mm); return NULL; }
    mat->M = mm->M; mat->N = mm->N; mat->NZ = mm->NZ;
    int rc;
    if (toCsr) {
        mat->IRP = calloc(mat->M + 1, sizeof *mat->IRP);
        mat->JA = malloc((mat->NZ ? mat->NZ : 1) * sizeof *mat->JA);
        mat->AS = malloc((mat->NZ ? mat->NZ : 1) * sizeof *mat->AS);
        rc = (mat->IRP && mat->JA && mat->AS) ? COOtoCSR(mm->entries, mat, mm->rowLens) : EXIT_FAILURE;
    } else {
        rc = COOtoELL(mm->entries, mat, mm->rowLens);
    }
    if (rc) { freeSpmat(mat); freeMatrixMarket(mm); return NULL; }
    mat->RL = mm->rowLens;      /* hand the row-length array over (parser.c:332-335) */
    mm->rowLens = NULL;
    freeMatrixMarket(mm);
    return mat;
}
spmat* MMtoCSR(char* matPath) { return fromMM(matPath, 1); }
spmat* MMtoELL(char* matPath) { return fromMM(matPath, 0); }

/*
 * structured.c -- MatrixMarket files shaped like the matrices the reference's
 * report publishes numbers for (doc/cudaNoRowLens_192_8.pdf, matrix
 * list doc/relazione.tex:463, sweep test/testAll.sh:7-8).  The SuiteSparse files
 * themselves cannot be fetched, so the bench measures stand-ins with the same
 * SHAPE -- rows, entries, longest row and, above all, column locality:
 *
 *   kind 0  3-D stencil on an nx x ny x nz grid, the 18 neighbours at |dx|+|dy|+|dz| in {1, 2} (6 faces + 12 edges),
 *           x fastest: 500 x 100 x 100 = 5 M rows, ~86 M entries, longest row 18  <- channel-500x100x100-b050
 *           (4.8 M rows, 85.4 M entries, longest row 18)
 *   kind 1  road network: degree 1..9 (mostly 2: chains), neighbours i-1 / i+1, short hops (geometric, mean 64) and a few
 *           far links: n = 12 M rows, ~25 M entries, longest row 9                <- asia_osm (11.95 M, 25.4 M, 9)
 *   kind 2  dense blocks along the diagonal: blocks of `bs` rows coupled to their two neighbours on each side and, one
 *           block in eight, to three more far blocks: n = 36 417, bs = 24: ~4.3 M entries, longest row <= 204
 *                                                                                 <- pdb1HYS (36 417, 4.34 M, 204)
 *
 * Rows are written in order with ascending columns (what the loader's consistency check demands,
 * src/lib/parser.c:195-202), values are dyadic rationals k/1024 (short decimals that parse back exactly).  The text
 * is produced by all cores -- every thread formats a contiguous block of rows into its own buffer with a hand-written
 * integer printer -- and written with one fwrite per block: 2.4 GB for the stencil in seconds, so that the bench can
 * afford to create the file and push it through MMtoCSR / MMtoELL (the loader runs at scale too).
 */
#include <omp.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "spmv_types.h"

#define STRUCT_MAX_ROW 256

static inline uint64_t mix64(uint64_t z) {
    z += 0x9E3779B97F4A7C15ull;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}

typedef struct { int kind; ulong p0, p1, p2, n; uint64_t seed; } gen_t;

static int cmpUlong(const void* a, const void* b) {
    const ulong x = *(const ulong*)a, y = *(const ulong*)b;
    return x < y ? -1 : x > y;
}

/* sorted distinct columns of `row` -> cols[]; returns their number */
static unsigned rowColumns(const gen_t* g, ulong row, ulong* cols) {
    unsigned n = 0;
    if (g->kind == 0) {
        const ulong nx = g->p0, ny = g->p1, nz = g->p2;
        const long x = (long)(row % nx), y = (long)(row / nx % ny), z = (long)(row / (nx * ny));
        for (long dz = -1; dz <= 1; ++dz)
            for (long dy = -1; dy <= 1; ++dy)
                for (long dx = -1; dx <= 1; ++dx) {
                    const long dist = labs(dx) + labs(dy) + labs(dz);
                    if (dist < 1 || dist > 2) continue;
                    const long X = x + dx, Y = y + dy, Z = z + dz;
                    if (X < 0 || Y < 0 || Z < 0 || X >= (long)nx || Y >= (long)ny || Z >= (long)nz) continue;
                    cols[n++] = ((ulong)Z * ny + (ulong)Y) * nx + (ulong)X;        /* (dz, dy, dx) ascending = column ascending */
                }
        return n;
    }
    if (g->kind == 1) {
        const uint64_t h = mix64(g->seed ^ mix64(row));
        const unsigned u = (unsigned)(h % 1000);
        unsigned deg = u < 50 ? 1 : u < 850 ? 2 : u < 970 ? 3 : u < 995 ? 4 : 5 + (unsigned)((h >> 20) % 5);
        ulong cand[16];
        unsigned nc = 0;
        if (row > 0) cand[nc++] = row - 1;
        if (row + 1 < g->n) cand[nc++] = row + 1;
        for (unsigned k = 0; nc < 14 && k < 12; ++k) {
            const uint64_t r = mix64(h + 0x51ED27ull * (k + 1));
            ulong c;
            if (r % 16 == 0) c = (ulong)((r >> 8) % g->n);                       /* a far link */
            else {
                ulong hop = 2;                                                   /* geometric: P(hop > t) ~ 2^-(t/44) */
                uint64_t bits = r >> 8;
                while ((bits & 1) && hop < 4096) { hop += 1 + (ulong)((bits >> 1) & 63); bits >>= 7; }
                c = (r >> 4) & 1 ? row + hop : row - hop;
                if (c >= g->n) c = row > hop ? row - hop : row + hop;             /* (row - hop wrapped or row + hop beyond the end) */
                if (c >= g->n) continue;
            }
            if (c == row) continue;
            cand[nc++] = c;
        }
        if (deg > nc) deg = nc;
        /* the first `deg` distinct candidates */
        for (unsigned k = 0; k < nc && n < deg; ++k) {
            int dup = 0;
            for (unsigned j = 0; j < n; ++j) dup |= cols[j] == cand[k];
            if (!dup) cols[n++] = cand[k];
        }
        qsort(cols, n, sizeof *cols, cmpUlong);
        return n;
    }
    /* kind 2 */
    const ulong bs = g->p1 ? g->p1 : 24, nb = (g->n + bs - 1) / bs, b = row / bs;
    ulong blocks[8];
    unsigned k = 0;
    for (long d = -2; d <= 2; ++d) {
        const long bb = (long)b + d;
        if (bb >= 0 && bb < (long)nb) blocks[k++] = (ulong)bb;
    }
    const uint64_t h = mix64(g->seed ^ mix64(b));
    if (h % 8 == 0)
        for (unsigned e = 0; e < 3; ++e) {
            const ulong fb = (ulong)(mix64(h + e) % nb);
            int dup = 0;
            for (unsigned j = 0; j < k; ++j) dup |= blocks[j] == fb;
            if (!dup) blocks[k++] = fb;
        }
    qsort(blocks, k, sizeof *blocks, cmpUlong);
    for (unsigned j = 0; j < k; ++j)
        for (ulong c = blocks[j] * bs; c < (blocks[j] + 1) * bs && c < g->n && n < STRUCT_MAX_ROW; ++c) cols[n++] = c;
    return n;
}

static inline char* putUlong(char* p, ulong v) {
    char tmp[24];
    int n = 0;
    do { tmp[n++] = (char)('0' + v % 10); v /= 10; } while (v);
    while (n) *p++ = tmp[--n];
    return p;
}

/* k / 1024, k in [-1023, 1023] \ {0}: sign, "0.", the ten decimals of |k| * 9765625 without trailing zeros */
static inline char* putValue(char* p, int k) {
    if (k < 0) { *p++ = '-'; k = -k; }
    *p++ = '0'; *p++ = '.';
    uint64_t frac = (uint64_t)k * 9765625ull;        /* < 10^10 */
    char d[10];
    for (int i = 9; i >= 0; --i) { d[i] = (char)('0' + frac % 10); frac /= 10; }
    int last = 9;
    while (last > 0 && d[last] == '0') --last;
    for (int i = 0; i <= last; ++i) *p++ = d[i];
    return p;
}

#define SPMV_SYNTH_PATTERN 16                       /* or-ed into `kind`: write a pattern file */

double spmvSynthStructuredValue(uint64_t seed, ulong row, ulong col) {
    int k = (int)(mix64(seed + 0xA5A5ull + mix64(row * 0x100000001B3ull ^ col)) % 2046) - 1023;
    if (k >= 0) ++k;                                  /* never 0 */
    return (double)k / 1024.0;
}
static inline int valueK(uint64_t seed, ulong row, ulong col) {
    int k = (int)(mix64(seed + 0xA5A5ull + mix64(row * 0x100000001B3ull ^ col)) % 2046) - 1023;
    return k >= 0 ? k + 1 : k;
}

/*
 * Writes the matrix to `path` as "%%MatrixMarket matrix coordinate real general" (kind | SPMV_SYNTH_PATTERN: "pattern
 * general", no values -- what the graphs of the DIMACS10 collection are distributed as).  p0..p2: kind 0 nx, ny, nz;
 * kind 1 rows; kind 2 rows, block size.  Reports rows, entries and the longest row.  EXIT_SUCCESS / EXIT_FAILURE.
 */
int spmvSynthWriteMtx(const char* path, int kind, ulong p0, ulong p1, ulong p2, uint64_t seed, ulong* M, ulong* NZ, ulong* maxRow) {
    const int pattern = (kind & SPMV_SYNTH_PATTERN) != 0;       /* "coordinate pattern general": entries without values (loaded as 1.0) */
    kind &= ~SPMV_SYNTH_PATTERN;
    gen_t g = {kind, p0, p1, p2, 0, seed};
    if (kind == 0) g.n = p0 * p1 * p2;
    else if (kind == 1 || kind == 2) g.n = p0;
    else return EXIT_FAILURE;
    if (g.n == 0) return EXIT_FAILURE;
    const int T = omp_get_max_threads();
    const int nBlk = T * 4;
    ulong* cnt = calloc((size_t)nBlk, sizeof *cnt);
    ulong* mx = calloc((size_t)nBlk, sizeof *mx);
    char** buf = calloc((size_t)nBlk, sizeof *buf);
    size_t* len = calloc((size_t)nBlk, sizeof *len);
    int rc = EXIT_FAILURE, bad = 0;
    FILE* fp = NULL;
    if (!cnt || !mx || !buf || !len) goto done;
    #pragma omp parallel for schedule(dynamic, 1)
    for (int b = 0; b < nBlk; ++b) {
        const ulong r0 = g.n / (ulong)nBlk * (ulong)b, r1 = b + 1 == nBlk ? g.n : g.n / (ulong)nBlk * (ulong)(b + 1);
        ulong cols[STRUCT_MAX_ROW];
        size_t cap = (size_t)(r1 - r0) * 64 + 4096, at = 0;
        char* out = malloc(cap);
        if (!out) { bad = 1; continue; }
        for (ulong r = r0; r < r1; ++r) {
            const unsigned n = rowColumns(&g, r, cols);
            cnt[b] += n;
            if (n > mx[b]) mx[b] = n;
            if (at + (size_t)n * 48 + 64 > cap) {
                cap = cap * 2 + (size_t)n * 48;
                char* bigger = realloc(out, cap);
                if (!bigger) { bad = 1; break; }
                out = bigger;
            }
            char* p = out + at;
            for (unsigned k = 0; k < n; ++k) {
                p = putUlong(p, r + 1); *p++ = ' ';
                p = putUlong(p, cols[k] + 1);
                if (!pattern) { *p++ = ' '; p = putValue(p, valueK(seed, r, cols[k])); }
                *p++ = '\n';
            }
            at = (size_t)(p - out);
        }
        buf[b] = out;
        len[b] = at;
    }
    if (bad) goto done;
    ulong nz = 0, longest = 0;
    for (int b = 0; b < nBlk; ++b) { nz += cnt[b]; if (mx[b] > longest) longest = mx[b]; }
    if (!(fp = fopen(path, "w"))) { perror("spmvSynthWriteMtx fopen"); goto done; }
    fprintf(fp, "%%%%MatrixMarket matrix coordinate %s general\n%% stand-in generated by spmvSynthWriteMtx kind %d (%lu %lu %lu) seed %lu\n%lu %lu %lu\n",
            pattern ? "pattern" : "real", kind, p0, p1, p2, (unsigned long)seed, g.n, g.n, nz);
    for (int b = 0; b < nBlk; ++b)
        if (len[b] && fwrite(buf[b], 1, len[b], fp) != len[b]) { perror("spmvSynthWriteMtx fwrite"); goto done; }
    if (fclose(fp)) { fp = NULL; perror("spmvSynthWriteMtx fclose"); goto done; }
    fp = NULL;
    if (M) *M = g.n;
    if (NZ) *NZ = nz;
    if (maxRow) *maxRow = longest;
    rc = EXIT_SUCCESS;
done:
    if (fp) fclose(fp);
    if (buf) for (int b = 0; b < nBlk; ++b) free(buf[b]);
    free(buf); free(len); free(cnt); free(mx);
    return rc;
}

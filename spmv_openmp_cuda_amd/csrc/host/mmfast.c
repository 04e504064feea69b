/*
 * mmfast.c -- the entry section of a MatrixMarket coordinate file parsed from
 * memory by all host cores.
 *
 * The reference reads entries with fscanf("%lu %lu %lf") one at a time
 * (src/lib/parser.c:59-64,73-97): ~4 M entries per second, i.e. 20 s for the
 * 85 M-entry matrices its report is about, before a single SpMV runs.  Here the
 * section is read into memory, cut at line ends into chunks, and every chunk is
 * tokenised independently: unsigned integers by hand, values by the exact
 * decimal fast path (at most 15 significant digits and |exponent| <= 22: one
 * correctly rounded multiplication or division, W. Clinger 1990) with strtod()
 * for everything else (long mantissas, inf/nan, hex floats), so every value is
 * the double fscanf's %lf would have produced.  The semantics of MMtoCOO are
 * kept: file order, 1-based -> 0-based, symmetric entries mirrored right behind
 * their original, pattern files get 1.0, bounds / zero-index / count errors
 * with the same messages.  Like fscanf the tokeniser is blind to line
 * structure INSIDE a chunk; a file whose entries straddle lines can make a chunk
 * start inside an entry -- detected (a chunk with a dangling token) and answered
 * with MMFAST_FALLBACK: the caller then parses the stream serially.
 */
#include <ctype.h>
#include <errno.h>
#include <omp.h>
#include <stdlib.h>
#include <string.h>

#include "parser.h"
#include "sparseMatrix.h"

enum { CH_OK = 0, CH_MALFORMED, CH_BOUNDS, CH_ZERO, CH_DANGLING, CH_NOMEM };

typedef struct {
    entry* e;
    ulong  n, fileEntries;
    int    status;
} chunk_t;

static inline int isSpace(char c) { return c == ' ' || c == '\n' || c == '\t' || c == '\r' || c == '\v' || c == '\f'; }

/* %lu the way the entries use it: optional '+', decimal digits */
static inline int scanUlong(const char** pp, const char* end, ulong* out) {
    const char* p = *pp;
    if (p < end && *p == '+') ++p;
    if (p >= end || *p < '0' || *p > '9') return 0;
    ulong v = 0;
    int digits = 0;
    while (p < end && *p >= '0' && *p <= '9') {
        v = v * 10 + (ulong)(*p - '0');
        ++p;
        if (++digits > 19) return 0;
    }
    if (p < end && !isSpace(*p)) return 0;
    *pp = p;
    *out = v;
    return 1;
}

static const double POW10[23] = {1e0, 1e1, 1e2, 1e3, 1e4, 1e5, 1e6, 1e7, 1e8, 1e9, 1e10, 1e11, 1e12, 1e13, 1e14, 1e15, 1e16,
                                 1e17, 1e18, 1e19, 1e20, 1e21, 1e22};

/* %lf: exact fast path, strtod otherwise.  The buffer is NUL-terminated behind `end`. */
static inline int scanDouble(const char** pp, const char* end, double* out) {
    const char* const start = *pp;
    const char* p = start;
    int neg = 0;
    if (p < end && (*p == '-' || *p == '+')) { neg = *p == '-'; ++p; }
    unsigned long long m = 0;
    int sig = 0, any = 0, exp10 = 0, fast = 1;
    while (p < end && *p >= '0' && *p <= '9') {
        any = 1;
        if (sig || *p != '0') { if (sig < 19) { m = m * 10 + (unsigned)(*p - '0'); ++sig; } else { fast = 0; } }
        ++p;
    }
    if (p < end && *p == '.') {
        ++p;
        while (p < end && *p >= '0' && *p <= '9') {
            any = 1;
            if (sig || *p != '0') { if (sig < 19) { m = m * 10 + (unsigned)(*p - '0'); ++sig; --exp10; } else { fast = 0; } }
            else --exp10;
            ++p;
        }
    }
    if (any && p < end && (*p == 'e' || *p == 'E')) {
        const char* q = p + 1;
        int eneg = 0, e = 0, ed = 0;
        if (q < end && (*q == '-' || *q == '+')) { eneg = *q == '-'; ++q; }
        while (q < end && *q >= '0' && *q <= '9') { if (e < 10000) e = e * 10 + (*q - '0'); ++q; ++ed; }
        if (ed) { exp10 += eneg ? -e : e; p = q; }
    }
    if (any && fast && sig <= 15 && exp10 >= -22 && exp10 <= 22 && (p >= end || isSpace(*p))) {
        double v = (double)m;
        v = exp10 < 0 ? v / POW10[-exp10] : v * POW10[exp10];
        *out = neg ? -v : v;
        *pp = p;
        return 1;
    }
    char* stop = NULL;
    errno = 0;
    const double v = strtod(start, &stop);
    if (stop == start || stop > end || (stop < end && !isSpace(*stop))) return 0;
    *out = v;
    *pp = stop;
    return 1;
}

static void parseChunk(const char* p, const char* end, int pat, int sym, ulong M, ulong N, chunk_t* c) {
    const size_t minLen = pat ? 4 : 6;               /* "1 1\n" / "1 1 1\n" */
    const size_t cap = ((size_t)(end - p) / minLen + 2) * (sym ? 2 : 1);
    c->e = malloc(cap * sizeof *c->e);
    c->n = c->fileEntries = 0;
    c->status = CH_OK;
    if (!c->e) { c->status = CH_NOMEM; return; }
    for (;;) {
        while (p < end && isSpace(*p)) ++p;
        if (p >= end) return;
        ulong row, col;
        double val = 1.0;
        if (!scanUlong(&p, end, &row)) { c->status = CH_MALFORMED; return; }
        while (p < end && isSpace(*p)) ++p;
        if (p >= end) { c->status = CH_DANGLING; return; }
        if (!scanUlong(&p, end, &col)) { c->status = CH_MALFORMED; return; }
        if (!pat) {
            while (p < end && isSpace(*p)) ++p;
            if (p >= end) { c->status = CH_DANGLING; return; }
            if (!scanDouble(&p, end, &val)) { c->status = CH_MALFORMED; return; }
        }
        if (row == 0 || col == 0) { c->status = CH_ZERO; return; }
        if (row > M || col > N || (sym && (col > M || row > N))) { c->status = CH_BOUNDS; return; }
        c->fileEntries++;
        c->e[c->n++] = (entry){.row = row - 1, .col = col - 1, .val = val};
        if (sym && row != col) c->e[c->n++] = (entry){.row = col - 1, .col = row - 1, .val = val};
    }
}

/*
 * buf[0..len) = everything behind the size line (buf[len] == 0).  On success
 * returns the COO entries (malloc), *NZ = their number after symmetric
 * expansion, rowLens[] (zeroed by the caller) counted; *status = 0.
 * On failure returns NULL with *status = MMFAST_ERROR (message printed) or
 * MMFAST_FALLBACK (nothing printed: parse the stream serially instead).
 */
entry* MMtoCOOFromBuffer(ulong* NZ, const char* buf, size_t len, MM_typecode mcode, ulong M, ulong N, ulong* rowLens, int* status) {
    const int sym = mm_is_symmetric(mcode), pat = mm_is_pattern(mcode);
    const ulong declared = *NZ;
    int nChunks = omp_get_max_threads() * 4;
    if ((size_t)nChunks > len / 65536 + 1) nChunks = (int)(len / 65536 + 1);
    chunk_t* ch = calloc((size_t)nChunks, sizeof *ch);
    const char** cut = malloc(((size_t)nChunks + 1) * sizeof *cut);
    entry* out = NULL;
    *status = MMFAST_ERROR;
    if (!ch || !cut) { ERRPRINT("MMtoCOO:  entries malloc errd\n"); goto done; }
    cut[0] = buf;
    for (int k = 1; k < nChunks; ++k) {              /* chunk starts: just behind a line end */
        const char* p = buf + len / (size_t)nChunks * (size_t)k;
        if (p < cut[k - 1]) p = cut[k - 1];
        while (p < buf + len && *p != '\n') ++p;
        cut[k] = p < buf + len ? p + 1 : buf + len;
    }
    cut[nChunks] = buf + len;
    #pragma omp parallel for schedule(dynamic, 1)
    for (int k = 0; k < nChunks; ++k) parseChunk(cut[k], cut[k + 1], pat, sym, M, N, &ch[k]);

    ulong total = 0, fileEntries = 0;
    for (int k = 0; k < nChunks; ++k) {
        switch (ch[k].status) {
            case CH_OK: break;
            case CH_DANGLING:                        /* an entry straddles a line end (or the file ends inside one): let fscanf decide */
            case CH_MALFORMED:                       /* could also be a chunk that started inside an entry: let fscanf decide */
                *status = MMFAST_FALLBACK; goto done;
            case CH_ZERO:   ERRPRINT("invalid matrix: 0 index in a 1-based file\n"); goto done;
            case CH_BOUNDS: ERRPRINT("invalid matrix: entry outside the declared dimensions\n"); goto done;
            default:        ERRPRINT("MMtoCOO:  entries malloc errd\n"); goto done;
        }
        total += ch[k].n;
        fileEntries += ch[k].fileEntries;
    }
    if (fileEntries > declared) { ERRPRINT("invalid matrix: more entries than declared\n"); goto done; }
    if (fileEntries != declared) { ERRPRINTS("invalid matrix: %lu entries declared, %lu found\n", declared, fileEntries); goto done; }
    out = malloc((total ? total : 1) * sizeof *out);
    if (!out) { ERRPRINT("MMtoCOO:  entries malloc errd\n"); goto done; }
    {
        ulong* off = malloc(((size_t)nChunks + 1) * sizeof *off);
        if (!off) { free(out); out = NULL; ERRPRINT("MMtoCOO:  entries malloc errd\n"); goto done; }
        off[0] = 0;
        for (int k = 0; k < nChunks; ++k) off[k + 1] = off[k] + ch[k].n;
        #pragma omp parallel for schedule(dynamic, 1)
        for (int k = 0; k < nChunks; ++k) {
            memcpy(out + off[k], ch[k].e, ch[k].n * sizeof *out);
            for (ulong i = 0; i < ch[k].n; ++i) {
                ulong* slot = &rowLens[ch[k].e[i].row];
                #pragma omp atomic
                (*slot)++;
            }
        }
        free(off);
    }
    *NZ = total;
    *status = 0;
done:
    if (ch) for (int k = 0; k < nChunks; ++k) free(ch[k].e);
    free(ch);
    free(cut);
    return out;
}
